#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_write(float4* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
template <int U> __global__ void k_write_u(float4* __restrict__ y, size_t n) {   // U stores in flight per thread
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride)
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = make_float4(1.f, 2.f, 3.f, (float)u);
    for (; i < n; i += stride) y[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t ny = 65536ull * 200 * 21 / 2;
    float4* y; CK(hipMalloc(&y, ny * 16)); CK(hipMemset(y, 0, ny * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int b, int t, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-14s grid %5d x %4d : %.4f ms %6.0f GB/s\n", name, b, t, ms, ny * 16.0 / ms / 1e6);
    };
    for (int t : {64, 256, 1024})
        for (int b : {64, 128, 256, 512, 1024, 2048})
            run("plain", b, t, [&] { hipLaunchKernelGGL(k_write, dim3(b), dim3(t), 0, 0, y, ny); });
    for (int b : {256, 512, 1024})
        run("unroll8", b, 256, [&] { hipLaunchKernelGGL(k_write_u<8>, dim3(b), dim3(256), 0, 0, y, ny); });
    run("memset", 0, 0, [&] { (void)hipMemsetAsync(y, 1, ny * 16, 0); });
    return 0;
}
