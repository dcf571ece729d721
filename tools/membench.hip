// Development micro-benchmark (GPU box): what HBM delivers for the hot path's traffic
// MIX with ideal access (16 B/lane, fully coalesced, no compute): per "frame" 96 B read
// and 168 B written, 65 536 x 200 frames = 1.26 GB in, 2.2 GB out per pass.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/membench tools/membench.hip && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define NTL(p) __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p))
#define NTS(v, p) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_read(const float4* __restrict__ x, float4* sink, size_t n) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x == 123.456f) sink[0] = a;
}
__global__ void k_write(float4* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_write_nt(float4* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        NTS((v4f{1.f, 2.f, 3.f, (float)i}), &y[i]);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int AUX> __global__ void k_write_buf(float4* __restrict__ y, size_t n) {
    // one descriptor per 2^31-byte window is enough for 2.2 GB? no: use per-block windows
    const size_t per = (n + gridDim.x - 1) / gridDim.x;            // float4 per block
    const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y + lo, (short)0, (int)((hi - lo) * 16), 0x00020000);
    for (size_t i = threadIdx.x; i < hi - lo; i += blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)i}, rs, (int)(i * 16), 0, AUX);
}
__global__ void k_mix2_nt(const float4* __restrict__ x, float4* __restrict__ y, size_t ngroups) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        v4f a = NTL(&x[g]), b = NTL(&x[g + ngroups]), c = NTL(&x[g + 2 * ngroups]), d = NTL(&x[g + 3 * ngroups]);
        NTS(a, &y[g]); NTS(b, &y[g + ngroups]); NTS(c, &y[g + 2 * ngroups]); NTS(d, &y[g + 3 * ngroups]);
        v4f s = a + b + c + d;
        NTS(s, &y[g + 4 * ngroups]); NTS(s, &y[g + 5 * ngroups]); NTS(s, &y[g + 6 * ngroups]);
    }
}
// each WAVE writes PWB contiguous bytes per step (1 KiB per instruction), waves round-robin over the buffer
template <int PWB> __global__ void k_write_wave(float4* __restrict__ y, size_t n) {
    const size_t per = PWB / 16;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t base = wave * per; base < n; base += nwaves * per)
#pragma unroll 4
        for (size_t i = lane; i < per; i += 64)
            if (base + i < n) y[base + i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// each LANE writes 64 contiguous bytes (4 x 16 B), lanes contiguous
__global__ void k_write_lane64(float4* __restrict__ y, size_t n) {
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        y[i] = make_float4(1.f, 2.f, 3.f, 4.f); y[i + 1] = make_float4(1.f, 2.f, 3.f, 4.f);
        y[i + 2] = make_float4(1.f, 2.f, 3.f, 4.f); y[i + 3] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
// mixed: each group of 4 input float4 produces 7 output float4 (96:168 = 4:7)
__global__ void k_mix(const float4* __restrict__ x, float4* __restrict__ y, size_t ngroups) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups * 4; g += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[g];
        y[g] = v;                                   // 4 of 7
    }
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups * 3; g += (size_t)gridDim.x * blockDim.x)
        y[ngroups * 4 + g] = make_float4(1.f, 2.f, 3.f, (float)g);   // 3 of 7
}
// mixed, interleaved in time: every thread alternates reads and writes
__global__ void k_mix2(const float4* __restrict__ x, float4* __restrict__ y, size_t ngroups) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        float4 a = x[g], b = x[g + ngroups], c = x[g + 2 * ngroups], d = x[g + 3 * ngroups];
        y[g] = a; y[g + ngroups] = b; y[g + 2 * ngroups] = c; y[g + 3 * ngroups] = d;
        float4 s = make_float4(a.x + b.x, a.y + c.y, b.z + d.z, c.w + d.w);
        y[g + 4 * ngroups] = s; y[g + 5 * ngroups] = s; y[g + 6 * ngroups] = s;
    }
}
int main() {
    const size_t frames = 65536ull * 200;
    const size_t nx = frames * 6, ny = frames * 21 / 2;  // float4 counts: 96 B and 168 B per frame
    float4 *x, *y;
    CK(hipMalloc(&x, nx * 16)); CK(hipMalloc(&y, ny * 16));
    CK(hipMemset(x, 0, nx * 16)); CK(hipMemset(y, 0, ny * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-34s %.4f ms  %7.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int blocks : {512, 2048}) {
        printf("grid %d x 256\n", blocks);
        run("read 1.26 GB", [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, x, y, nx); }, nx * 16.0);
        run("write 2.20 GB", [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write nt 2.20 GB", [&] { hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=0", [&] { hipLaunchKernelGGL(k_write_buf<0>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=1 (sc0)", [&] { hipLaunchKernelGGL(k_write_buf<1>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=2 (nt)", [&] { hipLaunchKernelGGL(k_write_buf<2>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=3 (sc0 nt)", [&] { hipLaunchKernelGGL(k_write_buf<3>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=16 (sc1)", [&] { hipLaunchKernelGGL(k_write_buf<16>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=17 (sc0 sc1)", [&] { hipLaunchKernelGGL(k_write_buf<17>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write buffer aux=18 (sc1 nt)", [&] { hipLaunchKernelGGL(k_write_buf<18>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write wave 1 KiB bursts", [&] { hipLaunchKernelGGL(k_write_wave<1024>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write wave 4 KiB bursts", [&] { hipLaunchKernelGGL(k_write_wave<4096>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write wave 32 KiB bursts", [&] { hipLaunchKernelGGL(k_write_wave<32768>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write wave 256 KiB bursts", [&] { hipLaunchKernelGGL(k_write_wave<262144>, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write lane 64 B", [&] { hipLaunchKernelGGL(k_write_lane64, dim3(blocks), dim3(256), 0, 0, y, ny); }, ny * 16.0);
        run("write 1024-thread blocks", [&] { hipLaunchKernelGGL(k_write, dim3(blocks / 4), dim3(1024), 0, 0, y, ny); }, ny * 16.0);
        run("hipMemsetAsync 2.20 GB", [&] { (void)hipMemsetAsync(y, 1, ny * 16, 0); }, ny * 16.0);
        run("mixed 96:168 nt loads+stores", [&] { hipLaunchKernelGGL(k_mix2_nt, dim3(blocks), dim3(256), 0, 0, x, y, nx / 4); }, (nx + ny) * 16.0);
        run("mixed 96:168 (copy then fill)", [&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, x, y, nx / 4); }, (nx + ny) * 16.0);
        run("mixed 96:168 (interleaved)", [&] { hipLaunchKernelGGL(k_mix2, dim3(blocks), dim3(256), 0, 0, x, y, nx / 4); }, (nx + ny) * 16.0);
    }
    return 0;
}
