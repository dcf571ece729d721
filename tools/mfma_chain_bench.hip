// Hardware probe (GPU box): cycles per v_mfma_f32_16x16x32_{f16,bf16} (and 32x32x16) as a function
// of the number of INDEPENDENT accumulator chains a wave keeps in flight, one wave per SIMD.
// The conv kernels accumulate 15 dependent MFMAs per M-tile with only 2-3 chains per wave.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/mfma_chain_bench tools/mfma_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND> // KIND 0: 16x16x32 f16, 1: 16x16x32 bf16, 2: 32x32x16 f16
__global__ __launch_bounds__(512, 1) void k_chain(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)in[lane * 8 + j]; b[j] = (_Float16)in[512 + lane * 8 + j];
        ab[j] = (__bf16)in[lane * 8 + j]; bb[j] = (__bf16)in[512 + lane * 8 + j];
    }
    f32x4 acc[K];
    f32x16 acc32[K];
    for (int k = 0; k < K; ++k) {
        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 16; ++j) acc32[k][j] = 0.f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
                else if (KIND == 1) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[k], 0, 0, 0);
                else acc32[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[k], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += acc[k][0] + acc[k][3] + acc32[k][0] + acc32[k][15];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int KIND> void run(const float* in, float* out, unsigned long long* cyc, int blocks, int threads = 256) {
    const int iters = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_chain<K, KIND>), dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<K, KIND>), dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[4]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double n = (double)iters * 4 * K;
    const char* names[3] = {"16x16x32 f16 ", "16x16x32 bf16", "32x32x16 f16 "};
    printf("%s  chains %2d  blocks %3d  waves/SIMD %d : %6.1f cycles / MFMA per wave = %5.1f per SIMD, %7.3f ms\n", names[KIND], K, blocks, threads / 256, h[0] / n, h[0] / n / (threads / 256), ms);
}

int main() {
    float *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    float h[1024]; srand(1); for (int i = 0; i < 1024; ++i) h[i] = (rand() % 2001 - 1000) / 4000.0f;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    for (int blocks : {1, 256}) {
        run<1, 0>(in, out, cyc, blocks); run<2, 0>(in, out, cyc, blocks); run<3, 0>(in, out, cyc, blocks);
        run<4, 0>(in, out, cyc, blocks); run<6, 0>(in, out, cyc, blocks); run<8, 0>(in, out, cyc, blocks);
        run<1, 1>(in, out, cyc, blocks); run<2, 1>(in, out, cyc, blocks); run<4, 1>(in, out, cyc, blocks);
        run<1, 2>(in, out, cyc, blocks); run<2, 2>(in, out, cyc, blocks); run<4, 2>(in, out, cyc, blocks);
        // two waves per SIMD (512-thread workgroups)
        run<1, 0>(in, out, cyc, blocks, 512); run<2, 0>(in, out, cyc, blocks, 512); run<3, 0>(in, out, cyc, blocks, 512);
        run<2, 1>(in, out, cyc, blocks, 512); run<2, 2>(in, out, cyc, blocks, 512);
    }
    return 0;
}
