// Development micro-benchmark: mixed 96:168 traffic with ROLES -- per 512-thread workgroup (one
// per CU) W waves only write (long coalesced bursts), the others only read.  Tells whether a
// writer-wave design could lift the mixed-traffic rate of the persistent kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int W, int BURST>  // W writer waves per block, BURST = float4 per writer burst (64 = 1 KiB)
__global__ __launch_bounds__(512) void k_roles(const float4* __restrict__ x, float4* __restrict__ y, float4* sink,
                                                size_t nx, size_t ny) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < W) {
        const size_t writers = (size_t)gridDim.x * W, me = (size_t)blockIdx.x * W + wave;
        for (size_t base = me * BURST; base < ny; base += writers * BURST)
#pragma unroll 4
            for (size_t i = lane; i < BURST; i += 64)
                if (base + i < ny) y[base + i] = make_float4(1.f, 2.f, 3.f, (float)i);
    } else {
        const size_t readers = (size_t)gridDim.x * (8 - W) * 64, me = ((size_t)blockIdx.x * (8 - W) + (wave - W)) * 64 + lane;
        float4 a = make_float4(0, 0, 0, 0);
        for (size_t i = me; i < nx; i += readers) { float4 v = x[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        if (a.x == 123.456f) sink[0] = a;
    }
}
int main() {
    const size_t frames = 65536ull * 200, nx = frames * 6, ny = frames * 21 / 2;
    float4 *x, *y; CK(hipMalloc(&x, nx * 16)); CK(hipMalloc(&y, ny * 16 + 64)); CK(hipMemset(x, 0, nx * 16)); CK(hipMemset(y, 0, ny * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-40s %.4f ms %6.0f GB/s\n", name, ms, (nx + ny) * 16.0 / ms / 1e6);
    };
#define RUN(W, B) run("writers/CU=" #W " burst=" #B " float4", [&] { hipLaunchKernelGGL((k_roles<W, B>), dim3(256), dim3(512), 0, 0, x, y, y + ny, nx, ny); })
    RUN(1, 64); RUN(1, 256); RUN(1, 2048); RUN(2, 64); RUN(2, 256); RUN(2, 2048); RUN(4, 64); RUN(4, 256); RUN(4, 2048);
    return 0;
}
