// Development microbenchmark (GPU box), no memory traffic:
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/mfma_bench tools/mfma_bench.hip && /tmp/mfma_bench
//  (1) how many waves per SIMD does the fp32 matrix pipe need?  NACC independent accumulator
//      chains of v_mfma_f32_16x16x4_f32 (or 32x32x2) at 1, 2, 4 waves per SIMD;
//  (2) do matrix and vector instructions overlap on a SIMD?  "mix": an MFMA wave and a plain
//      v_fma_f32 wave on the same SIMD, each alone and together (fp32 and f16 MFMA); and one wave
//      with NV v_fma_f32 after every MFMA.
// Results of round 1 are in DESIGN.md section 9: (1) one wave is enough (145-148 TFLOP/s),
// (2) the times add.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;
}

// Co-issue probe: waves [0, 4) of a workgroup (one per SIMD) run MFMAs, waves [4, 8) run
// independent v_fma_f32 chains (nvalu per MFMA-equivalent slot); mode selects who runs.
__global__ void kmix(float* out, int iters, float a0, float b0, int mode) {
    const int wave = threadIdx.x >> 6;
    float a = a0 + threadIdx.x, b = b0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.678f) out[0] = s;
    } else {
        if (!(mode & 2)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a)); // plain (unpacked) VALU
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 12345.678f) out[0] = s;
    }
}
// same co-issue probe with the f16 matrix instruction (v_mfma_f32_16x16x32_f16, 4 passes)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void kmix16(float* out, int iters, float a0, float b0, int mode) {
    const int wave = threadIdx.x >> 6;
    float a = a0 + threadIdx.x, b = b0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 acc[8];
        f16x8 av, bv;
        for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; av[i] = (_Float16)a; bv[i] = (_Float16)b; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.678f) out[0] = s;
    } else {
        if (!(mode & 2)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 12345.678f) out[0] = s;
    }
}
// one wave per SIMD, NV v_fma_f32 after every MFMA in program order
template <int NV>
__global__ void kinter(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x, b = b0;
    f32x4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = a + i; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + k) & 7]) : "v"(b), "v"(a));
                __builtin_amdgcn_sched_barrier(0);
            }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
static void run(const char* name, K kern, int nacc, double flop_per_mfma, int wps) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 4000;
    dim3 grid(256 * 4), block(64 * wps); // 4 workgroups per CU of wps waves: one wave per SIMD each x wps
    grid = dim3(256), block = dim3(64 * 4 * wps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)256 * 4 * wps * iters * 4 * nacc;
    printf("%-10s nacc=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", name, nacc, wps, ms, mfma * flop_per_mfma / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int wps : {1, 2, 4}) {
        run("16x16x4", k16<4>, 4, 2048.0, wps);
        run("16x16x4", k16<8>, 8, 2048.0, wps);
        run("32x32x2", k32<2>, 2, 4096.0, wps);
        run("32x32x2", k32<4>, 4, 4096.0, wps);
    }
    for (int mode : {1, 2, 3}) {
        float* out; hipMalloc(&out, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kmix, dim3(256), dim3(512), 0, 0, out, 10, 1.f, 1.f, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix, dim3(256), dim3(512), 0, 0, out, 4000, 1.f, 1.f, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mix mode=%d (1 = MFMA waves only, 2 = VALU waves only, 3 = both): %.3f ms\n", mode, ms);
    }
    for (int mode : {1, 2, 3}) {
        float* out; hipMalloc(&out, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kmix16, dim3(256), dim3(512), 0, 0, out, 10, 1.f, 1.f, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix16, dim3(256), dim3(512), 0, 0, out, 4000, 1.f, 1.f, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("f16 mix mode=%d (64 f16 MFMAs vs 128 v_fma per iteration): %.3f ms\n", mode, ms);
    }
    auto runi = [](const char* name, auto kern) {
        float* out; hipMalloc(&out, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 10, 1.f, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 4000, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms (pure MFMA stream: 1.83 ms)\n", name, ms);
    };
    runi("one wave/SIMD, 1 v_fma after each MFMA", kinter<1>);
    runi("one wave/SIMD, 2 v_fma after each MFMA", kinter<2>);
    runi("one wave/SIMD, 4 v_fma after each MFMA", kinter<4>);
    runi("one wave/SIMD, 6 v_fma after each MFMA", kinter<6>);
    return 0;
}
