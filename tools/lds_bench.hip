// Development probe (GPU box): LDS read bandwidth per CU for the chain kernel's access pattern --
// every wave of a 512-thread workgroup reads the same 64 KB of LDS with lane-contiguous
// ds_read_b128 (fragment index * 64 + lane), nothing else.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/lds_bench tools/lds_bench.hip && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH> // bytes per lane per read: 16 (b128) or 8 (b64)
__global__ __launch_bounds__(512) void lds_read(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 16
        for (int f = 0; f < 64; ++f) { // 64 fragments x 64 lanes x 16 B = 64 KB per wave per iteration
            if (WIDTH == 16) {
                acc += *reinterpret_cast<const f32x4*>(smem + (f * 64 + lane) * 16);
            } else {
                const float2 a = *reinterpret_cast<const float2*>(smem + (f * 64 + lane) * 16);
                const float2 b = *reinterpret_cast<const float2*>(smem + (f * 64 + lane) * 16 + 8);
                acc[0] += a.x; acc[1] += a.y; acc[2] += b.x; acc[3] += b.y;
            }
        }
        asm volatile("" ::: "memory");
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    for (int width : {16, 8}) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int n) {
            if (width == 16) hipLaunchKernelGGL(lds_read<16>, dim3(256), dim3(512), 65536, 0, out, n);
            else hipLaunchKernelGGL(lds_read<8>, dim3(256), dim3(512), 65536, 0, out, n);
        };
        launch(10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes_per_cu = 8.0 * 65536 * iters; // 8 waves x 64 KB per iteration
        printf("ds_read_b%d: %.3f ms, %.1f GB/s per CU, %.1f TB/s chip, %.1f B/clk/CU at the nominal %.2f GHz\n", width * 8, ms,
               bytes_per_cu / ms / 1e6, bytes_per_cu * 256 / ms / 1e9, bytes_per_cu / (ms * 1e-3) / (ghz * 1e9), ghz);
    }
    return 0;
}
