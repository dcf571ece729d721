// Hardware probe (GPU box): what one wave per SIMD pays for vector and LDS work placed BETWEEN
// in-place v_mfma_f32_16x16x32_f16 (the f16x3 conv kernel's tile loop: 30 MFMAs, ~30 VALU, 10
// ds_read_b128, 2 ds_write_b128 per 16-frame tile).  NV VALU (v_fma_f32 on independent registers)
// and ND ds_read_b128 per 30 MFMAs, spread evenly by sched_group_barrier.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/mfma_mix_bench tools/mfma_mix_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int ND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_mix(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = in[i & 1023];
    __syncthreads();
    f16x8 a[5], b[10];
    for (int s = 0; s < 5; ++s) for (int j = 0; j < 8; ++j) a[s][j] = (_Float16)in[(lane * 8 + j + 64 * s) & 1023];
    for (int s = 0; s < 10; ++s) b[s] = *reinterpret_cast<const f16x8*>(lds + ((lane * 16 + s * 1024) & 32767));
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[lane + j];
    const float c = in[5];
    int off = (lane & 15) * 64 + (lane >> 4) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        f16x8 nb[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) nb[s] = b[s];
#pragma unroll
        for (int s = 0; s < ND; ++s) nb[s] = *reinterpret_cast<const f16x8*>(lds + ((off + s * 1024) & 32767));
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[2 * s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[2 * s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[2 * s + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[2 * s + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 - s], b[2 * s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 - s], b[2 * s], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], c, 0.25f);
#pragma unroll
        for (int s = 0; s < 10; ++s) b[s] = nb[s];
        off = (off + 1024) & 16383;
        asm volatile("" : "+v"(off));
        // one MFMA, then its share of the vector / LDS work
#pragma unroll
        for (int k = 0; k < 30; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (ND && k % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV / 30, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc1[3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int ND, int WAVES> void run(const float* in, float* out, unsigned long long* cyc) {
    const int iters = 3000, blocks = 256;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mix<NV, ND, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, cyc, iters);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NV, ND, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    printf("waves/SIMD %d  30 MFMA + %2d VALU + %2d ds_read_b128 : %6.0f cycles / iteration (wave 0 of block 128), %6.1f ns / iteration wall\n",
           WAVES / 4, NV, ND, (double)h[128] / iters, ms * 1e6 / iters);
}

int main() {
    float *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    float h[1024]; srand(1); for (int i = 0; i < 1024; ++i) h[i] = (rand() % 2001 - 1000) / 4000.0f;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    run<0, 0, 4>(in, out, cyc); run<30, 0, 4>(in, out, cyc); run<60, 0, 4>(in, out, cyc); run<90, 0, 4>(in, out, cyc);
    run<0, 10, 4>(in, out, cyc); run<30, 10, 4>(in, out, cyc); run<60, 10, 4>(in, out, cyc);
    run<0, 0, 8>(in, out, cyc); run<30, 0, 8>(in, out, cyc); run<60, 0, 8>(in, out, cyc); run<30, 10, 8>(in, out, cyc); run<60, 10, 8>(in, out, cyc);
    return 0;
}
