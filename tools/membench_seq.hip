// Development micro-benchmark (GPU box): what HBM delivers for the hot path's traffic at the
// kernel's REAL granularity, with no compute: every wave streams whole sequences -- 19 200 B read
// (200 frames x 96 B), 33 600 B written (200 x 168 B) -- from a persistent grid of one workgroup per
// CU, W waves per workgroup.  Compared with the "ideal" grid-stride form in which the whole grid
// walks one contiguous window (the pattern of the runtime's fill kernel, __amd_rocclr_fillBufferAligned:
// 256 workgroups x 256 threads, 16 B per lane, grid stride).
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/membench_seq tools/membench_seq.hip
//   tools/_build/membench_seq [sequences per launch = 65536] [quick: W = 8 only]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int T = 200, XB = T * 96, YB = T * 168;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}

// MODE bit0: read the input sequence, bit1: write the output sequence.
// PAT 0: output stores lane-linear (33 x 1 KiB per sequence); PAT 1: the kernel's head pattern
// (per 16-frame tile three 16-B stores per lane: 16 rows x 64-B segments at a 168-B stride).
// ADJ 1: the W waves of a workgroup take W adjacent sequences; ADJ 0: sequences strided by the grid.
// SAUX / LAUX: cache-policy bits of the stores / loads (0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1).
template <int MODE, int PAT, int ADJ, int SAUX = 0, int LAUX = 0>
__global__ void k_seq(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int stride = gridDim.x * W;
    int seq = ADJ ? blockIdx.x * W + wave : blockIdx.x + gridDim.x * wave;
    unsigned acc = 0;
    for (; seq < nseq; seq += stride) {
        u32x4 v[19];
        if (MODE & 1) {
            const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq * XB, XB);
#pragma unroll
            for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LAUX);
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
        }
        if (MODE & 2) {
            const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq * YB, YB);
            const u32x4 d = {acc, 2u, 3u, (unsigned)seq};
            if (PAT == 0) {
#pragma unroll
                for (int j = 0; j < 33; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SAUX);
            } else {
                const int tcol = lane & 15, q = lane >> 4;
                const int off = tcol * 168 + 16 * q;
#pragma unroll
                for (int m = 0; m < 13; ++m) {
                    __builtin_amdgcn_raw_buffer_store_b128(d, ws, off, m * 2688, SAUX);
                    __builtin_amdgcn_raw_buffer_store_b128(d, ws, off, m * 2688 + 64, SAUX);
                    if (q < 2) __builtin_amdgcn_raw_buffer_store_b128(d, ws, off, m * 2688 + 128, SAUX);
                    else if (q == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{acc, 1u}, ws, off, m * 2688 + 128, SAUX);
                }
            }
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// the ideal form: the whole grid walks contiguous windows, 4 loads : 7 stores per lane and step
__global__ void k_ideal(const float4* __restrict__ x, float4* __restrict__ y, size_t ngroups) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        float4 a = x[g], b = x[g + ngroups], c = x[g + 2 * ngroups], d = x[g + 3 * ngroups];
        y[g] = a; y[g + ngroups] = b; y[g + 2 * ngroups] = c; y[g + 3 * ngroups] = d;
        float4 s = make_float4(a.x + b.x, a.y + c.y, b.z + d.z, c.w + d.w);
        y[g + 4 * ngroups] = s; y[g + 5 * ngroups] = s; y[g + 6 * ngroups] = s;
    }
}

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 65536; // sequences per launch
    const bool quick = argc > 2;                       // only the W = 8 rows
    char *x, *y; unsigned* sink;
    CK(hipMalloc(&x, (size_t)nseq * XB)); CK(hipMalloc(&y, (size_t)nseq * YB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(x, 1, (size_t)nseq * XB)); CK(hipMemset(y, 0, (size_t)nseq * YB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int W, double bytes, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20; best = ms < best ? ms : best; tot += ms;
        }
        printf("%-44s W=%2d : avg %.4f ms  best %.4f ms  %6.0f GB/s\n", name, W, tot / 5, best, bytes / (tot / 5) / 1e6);
    };
    const double rb = (double)nseq * XB, wb = (double)nseq * YB;
    for (int W : {1, 2, 4, 8, 16}) {
        if (quick && W != 8) continue;
        const dim3 g(256), b(64 * W);
        run("read-only  seq", W, rb, [&] { hipLaunchKernelGGL((k_seq<1, 0, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("write-only seq linear", W, wb, [&] { hipLaunchKernelGGL((k_seq<2, 0, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("write-only seq head-pattern", W, wb, [&] { hipLaunchKernelGGL((k_seq<2, 1, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed seq linear", W, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed seq head-pattern", W, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed seq head-pattern, adjacent waves", W, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 1>), g, b, 0, 0, x, y, nseq, sink); });
    }
    {   // cache policies on the kernel's own shape (8 waves per CU, head-pattern stores)
        const dim3 g(256), b(64 * 8);
        run("mixed head-pattern, stores sc0", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 1, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed head-pattern, stores sc1", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 16, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed head-pattern, stores sc0 sc1", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 17, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed head-pattern, loads nt", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 0, 2>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed head-pattern, loads sc1", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 0, 16>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed head-pattern, loads nt + stores sc1", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 1, 0, 16, 2>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, loads nt", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 0, 2>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, stores nt", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 2, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, stores sc0", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 1, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, stores sc1", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 16, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, stores sc0 nt", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 3, 0>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear, stores nt + loads nt", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0, 2, 2>), g, b, 0, 0, x, y, nseq, sink); });
        run("mixed linear (default policy, again)", 8, rb + wb, [&] { hipLaunchKernelGGL((k_seq<3, 0, 0>), g, b, 0, 0, x, y, nseq, sink); });
    }
    for (int W : {1, 4, 8}) {
        const dim3 g(256), b(64 * W);
        run("mixed IDEAL grid-stride (fill-kernel form)", W, rb + wb,
            [&] { hipLaunchKernelGGL(k_ideal, g, b, 0, 0, (const float4*)x, (float4*)y, (size_t)nseq * XB / 64); });
    }
    run("hipMemsetAsync of the output", 0, wb, [&] { (void)hipMemsetAsync(y, 1, (size_t)nseq * YB, 0); });
    return 0;
}
