// Development micro-benchmark (GPU box), round 3: WHEN and BY WHOM the hot path's HBM traffic is issued.
// tools/membench_seq.hip fixed the traffic's granularity (a wave streams whole sequences: 19 200 B in,
// 33 600 B out, lane-linear) and found 2.63 ms per 262 144 sequences at 8 waves per CU against 2.2 ms
// for the same bytes as a pure read followed by a pure write.  This probe varies the schedule:
//   base      every wave: 19 loads, wait, 33 stores                     (= "mixed seq linear")
//   trickle   the stores leave in 13 groups of <= 3 with idle time between (what the head's tile
//             loop does), the loads of the NEXT sequence already in flight (the kernel's prefetch)
//   half      12 waves per CU on half sequences (the three-waves-per-SIMD geometry)
//   roles     R reader waves + (W - R) writer waves per CU (a store-wave design)
//   phased    all waves of the chip alternate between a read window and a write window of a common
//             clock (s_memrealtime), i.e. the DRAM bus turns around once per period instead of always
//   copy      plain float4 copy of the same byte count with a big grid (calibrates the box)
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/membench_sched tools/membench_sched.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int T = 200, XB = T * 96, YB = T * 168;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); } // 100 MHz, chip-wide

// ---- base / trickle ---------------------------------------------------------------------------
// GROUPS = 1: all 33 stores back to back.  GROUPS = 13: 13 groups (3,3,...,3 -> 39 slots, 33 real)
// with s_sleep(SLEEP) between them.  PREFETCH: the next sequence's loads are issued before this
// sequence's stores (the kernel's order).
template <int GROUPS, int SLEEP, bool PREFETCH, int LA = 0, int SA = 0> // LA / SA: cache policy (aux) of loads / stores
__global__ void k_trickle(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int stride = gridDim.x * W;
    int seq = blockIdx.x + gridDim.x * wave;
    if (seq >= nseq) return;
    unsigned acc = 0;
    u32x4 v[19];
    {
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq * XB, XB);
#pragma unroll
        for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
    }
    for (; seq < nseq; seq += stride) {
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
        if (PREFETCH) {
            const int nx = seq + stride;
            const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)(nx < nseq ? nx : seq) * XB, nx < nseq ? XB : 0);
#pragma unroll
            for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
        }
        const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq * YB, YB);
        const u32x4 d = {acc, 2u, 3u, (unsigned)seq};
        constexpr int PER = (33 + GROUPS - 1) / GROUPS;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int j = g * PER; j < (g + 1) * PER && j < 33; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (!PREFETCH) {
            const int nx = seq + stride;
            const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)(nx < nseq ? nx : seq) * XB, nx < nseq ? XB : 0);
#pragma unroll
            for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// ---- half sequences (100 frames + 8 halo: 108 rows in = 11 loads, 100 rows out = 17 stores) -----
template <int GROUPS, int SLEEP, int LA = 0, int SA = 0>
__global__ void k_half(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int stride = gridDim.x * W, nch = 2 * nseq;
    int ch = blockIdx.x + gridDim.x * wave;
    if (ch >= nch) return;
    unsigned acc = 0;
    u32x4 v[11];
    auto issue = [&](int c, bool on) {
        const int s = c >> 1, h = c & 1;
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)s * XB + (h ? 92 * 96 : 0), on ? 108 * 96 : 0);
#pragma unroll
        for (int j = 0; j < 11; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
    };
    issue(ch, true);
    for (; ch < nch; ch += stride) {
#pragma unroll
        for (int j = 0; j < 11; ++j) acc += v[j][0] ^ v[j][3];
        const int nx = ch + stride;
        issue(nx < nch ? nx : ch, nx < nch);
        const int s = ch >> 1, h = ch & 1;
        const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)s * YB + (h ? 100 * 168 : 0), 100 * 168);
        const u32x4 d = {acc, 2u, 3u, (unsigned)ch};
        constexpr int PER = (17 + GROUPS - 1) / GROUPS;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int j = g * PER; j < (g + 1) * PER && j < 17; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// ---- roles: waves [0, R) read every sequence of the workgroup's share, waves [R, W) write them ----
template <int R, int LA = 0, int SA = 0>
__global__ void k_roles(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned acc = 0;
    if (wave < R) {
        for (int seq = blockIdx.x + gridDim.x * wave; seq < nseq; seq += gridDim.x * R) {
            const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq * XB, XB);
            u32x4 v[19];
#pragma unroll
            for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
        }
    } else {
        const int nw = W - R;
        for (int seq = blockIdx.x + gridDim.x * (wave - R); seq < nseq; seq += gridDim.x * nw) {
            const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq * YB, YB);
            const u32x4 d = {acc, 2u, 3u, (unsigned)seq};
#pragma unroll
            for (int j = 0; j < 33; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// ---- phased: loads only while (now - t0) mod P < RD, stores only in the rest of the period ------
__global__ void k_stamp(unsigned long long* t0) { if (threadIdx.x == 0) t0[0] = rt(); }
__global__ void k_phased(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink,
                         const unsigned long long* t0p, int P, int RD) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int stride = gridDim.x * W;
    const unsigned long long t0 = t0p[0];
    unsigned acc = 0;
    auto wait_window = [&](bool reading) {
        for (int spin = 0; spin < 100000; ++spin) { // bounded: a stuck clock cannot hang the grid
            const int ph = (int)((rt() - t0) % (unsigned)P);
            if ((ph < RD) == reading) break;
            __builtin_amdgcn_s_sleep(2);
        }
    };
    for (int seq = blockIdx.x + gridDim.x * wave; seq < nseq; seq += stride) {
        u32x4 v[19];
        wait_window(true);
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq * XB, XB);
#pragma unroll
        for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, 0);
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
        wait_window(false);
        const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq * YB, YB);
        const u32x4 d = {acc, 2u, 3u, (unsigned)seq};
#pragma unroll
        for (int j = 0; j < 33; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, 0);
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

__global__ void k_copy(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i];
}
__global__ void k_copy_gs(const float4* __restrict__ x, float4* __restrict__ y, size_t n) { // grid stride, persistent
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
// ---- sync: persistent, but the workgroup's waves move in ROUNDS (a barrier per round): round k = the eight
// ADJACENT sequences 8 (b + G k) .. + 7 (ADJ) or b + G (8k + wave) (far apart).  What the non-persistent form
// does per workgroup, without its dispatch.  PREFETCH: next round's loads before this round's stores.
template <bool PREFETCH, bool ADJ, int LA, int SA>
__global__ void k_sync(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int rounds = nseq / (8 * (int)gridDim.x);
    auto seq_of = [&](int k) { return ADJ ? 8 * ((int)blockIdx.x + (int)gridDim.x * k) + wave : (int)blockIdx.x + (int)gridDim.x * (8 * k + wave); };
    unsigned acc = 0;
    u32x4 v[19];
    auto issue = [&](int k) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq_of(k < rounds ? k : 0) * XB, k < rounds ? XB : 0);
#pragma unroll
        for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
    };
    issue(0);
    for (int k = 0; k < rounds; ++k) {
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
        __syncthreads();
        if (PREFETCH) issue(k + 1);
        const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq_of(k) * YB, YB);
        const u32x4 d = {acc, 2u, 3u, (unsigned)k};
#pragma unroll
        for (int j = 0; j < 33; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
        if (!PREFETCH) issue(k + 1);
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// ---- dyn: persistent, every wave CLAIMS runs of RUN consecutive sequences from one global counter (claimed one run
// ahead, so the atomic's latency is never waited for): the in-order dynamic dispatch of the non-persistent form,
// inside a persistent kernel.
template <int RUN, int LA, int SA, int GROUPS = 1, int SLEEP = 0> // GROUPS / SLEEP: the stores trickle like the head layer's
__global__ void k_dyn(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink, unsigned* counter) {
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    u32x4 v[19];
    auto claim = [&]() {
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(counter, (unsigned)RUN);
        return (int)__builtin_amdgcn_readfirstlane(c);
    };
    auto issue = [&](int sq) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)(sq < nseq ? sq : 0) * XB, sq < nseq ? XB : 0);
#pragma unroll
        for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
    };
    int cur = claim(), nxt = claim();
    issue(cur);
    while (cur < nseq) {
        for (int r = 0; r < RUN; ++r) {
            const int sq = cur + r;
            if (sq >= nseq) break;
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
            issue(r + 1 < RUN ? sq + 1 : nxt); // prefetch: next sequence of the run, or the first of the next run
            const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)sq * YB, YB);
            const u32x4 d = {acc, 2u, 3u, (unsigned)sq};
            constexpr int PER = (33 + GROUPS - 1) / GROUPS;
#pragma unroll
            for (int gq = 0; gq < GROUPS; ++gq) {
#pragma unroll
                for (int j = gq * PER; j < (gq + 1) * PER && j < 33; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
        cur = nxt;
        nxt = claim();
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// ---- dyn + scratch: the head layer's trickle goes to a per-wave scratch (33.6 KB each, reused every sequence, so it
// should stay in L2 / Infinity Cache), and the sequence's rows then leave for y as ONE burst read back from there.
template <int SLEEP, int SCR_ST, int SCR_LD>
__global__ void k_dyn_scratch(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink, unsigned* counter,
                              char* __restrict__ scratch) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t ss = rsrc(scratch + (size_t)gw * (34 * 1024), 34 * 1024);
    unsigned acc = 0;
    u32x4 v[19];
    auto claim = [&]() {
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(counter, 2u);
        return (int)__builtin_amdgcn_readfirstlane(c);
    };
    auto issue = [&](int sq) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)(sq < nseq ? sq : 0) * XB, sq < nseq ? XB : 0);
#pragma unroll
        for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, 2);
    };
    int cur = claim(), nxt = claim();
    issue(cur);
    while (cur < nseq) {
        for (int r = 0; r < 2; ++r) {
            const int sq = cur + r;
            if (sq >= nseq) break;
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
            issue(r == 0 ? sq + 1 : nxt);
            const u32x4 d = {acc, 2u, 3u, (unsigned)sq};
#pragma unroll
            for (int gq = 0; gq < 11; ++gq) { // the trickle, into the scratch
#pragma unroll
                for (int j = gq * 3; j < gq * 3 + 3; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ss, lane * 16, j * 1024, SCR_ST);
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
            const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)sq * YB, YB);
#pragma unroll
            for (int h = 0; h < 3; ++h) { // the burst: 11 rows at a time through registers
                u32x4 t[11];
#pragma unroll
                for (int j = 0; j < 11; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b128(ss, lane * 16, (h * 11 + j) * 1024, SCR_LD);
#pragma unroll
                for (int j = 0; j < 11; ++j) __builtin_amdgcn_raw_buffer_store_b128(t[j], ws, lane * 16, (h * 11 + j) * 1024, 18);
            }
        }
        cur = nxt;
        nxt = claim();
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

// non-persistent seq kernel: one wave per sequence, the dispatcher hands workgroups out in order
template <int LA = 0, int SA = 0>
__global__ void k_seq_np(const char* __restrict__ x, char* __restrict__ y, int nseq, unsigned* sink) {
    const int W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int seq = blockIdx.x * W + wave;
    if (seq >= nseq) return;
    const __amdgpu_buffer_rsrc_t rs = rsrc(x + (size_t)seq * XB, XB);
    u32x4 v[19];
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 19; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, LA);
#pragma unroll
    for (int j = 0; j < 19; ++j) acc += v[j][0] ^ v[j][3];
    const __amdgpu_buffer_rsrc_t ws = rsrc(y + (size_t)seq * YB, YB);
    const u32x4 d = {acc, 2u, 3u, (unsigned)seq};
#pragma unroll
    for (int j = 0; j < 33; ++j) __builtin_amdgcn_raw_buffer_store_b128(d, ws, lane * 16, j * 1024, SA);
    if (acc == 0x12345679u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 262144;
    char *x, *y; unsigned* sink; unsigned long long* t0;
    CK(hipMalloc(&x, (size_t)nseq * XB)); CK(hipMalloc(&y, (size_t)nseq * YB)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&t0, 64));
    CK(hipMemset(x, 1, (size_t)nseq * XB)); CK(hipMemset(y, 0, (size_t)nseq * YB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)nseq * (XB + YB);
    auto run = [&](const char* name, double b, auto launch) {
        for (int i = 0; i < 4; ++i) launch();
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; best = ms < best ? ms : best; tot += ms;
        }
        printf("%-56s avg %.4f ms  best %.4f ms  %6.0f GB/s\n", name, tot / 4, best, b / (tot / 4) / 1e6);
        fflush(stdout);
    };
#define L(K, W) [&] { hipLaunchKernelGGL(K, dim3(256), dim3(64 * (W)), 0, 0, x, y, nseq, sink); }
    if (argc > 2 && !strcmp(argv[2], "policy")) { // cache policies (aux: 1 = sc0, 2 = nt, 16 = sc1) on the kernel's pattern
        for (int rep = 0; rep < 2; ++rep) {
            printf("# policy pass %d, %d sequences\n", rep, nseq);
            run("base W=8 prefetch, loads 0  stores 0", bytes, L((k_trickle<1, 0, true, 0, 0>), 8));
            run("base W=8 prefetch, loads 0  stores nt", bytes, L((k_trickle<1, 0, true, 0, 2>), 8));
            run("base W=8 prefetch, loads nt stores nt", bytes, L((k_trickle<1, 0, true, 2, 2>), 8));
            run("base W=8 prefetch, loads nt stores nt sc1", bytes, L((k_trickle<1, 0, true, 2, 18>), 8));
            run("base W=8 prefetch, loads 0  stores nt sc1", bytes, L((k_trickle<1, 0, true, 0, 18>), 8));
            run("base W=8 prefetch, loads 0  stores sc1", bytes, L((k_trickle<1, 0, true, 0, 16>), 8));
            run("base W=8 prefetch, loads 0  stores sc0 sc1", bytes, L((k_trickle<1, 0, true, 0, 17>), 8));
            run("base W=8 prefetch, loads nt sc1 stores nt sc1", bytes, L((k_trickle<1, 0, true, 18, 18>), 8));
            run("trickle W=8 11 x 3 sleep 10, loads nt stores nt sc1", bytes, L((k_trickle<11, 10, true, 2, 18>), 8));
            run("trickle W=8 11 x 3 sleep 10, loads 0 stores 0", bytes, L((k_trickle<11, 10, true, 0, 0>), 8));
            run("half W=12, loads 0 stores 0", bytes, L((k_half<1, 0, 0, 0>), 12));
            run("half W=12, loads nt stores nt sc1", bytes, L((k_half<1, 0, 2, 18>), 12));
            run("half W=12 6 x 3 sleep 10, loads nt stores nt sc1", bytes, L((k_half<6, 10, 2, 18>), 12));
            run("base W=4 prefetch, loads nt stores nt sc1", bytes, L((k_trickle<1, 0, true, 2, 18>), 4));
            run("base W=16 prefetch, loads nt stores nt sc1", bytes, L((k_trickle<1, 0, true, 2, 18>), 16));
            run("non-persistent 8 waves/wg, loads 0 stores 0", bytes, [&] { hipLaunchKernelGGL((k_seq_np<0, 0>), dim3(nseq / 8), dim3(512), 0, 0, x, y, nseq, sink); });
            run("non-persistent 8 waves/wg, loads nt stores nt sc1", bytes, [&] { hipLaunchKernelGGL((k_seq_np<2, 18>), dim3(nseq / 8), dim3(512), 0, 0, x, y, nseq, sink); });
            // the same with the real kernel's occupancy: 160 KB of dynamic LDS per 8-wave workgroup = one workgroup per CU
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq_np<2, 18>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq_np<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            run("non-persistent 8 waves/wg, ONE workgroup per CU (160 KB LDS), loads 0 stores 0", bytes, [&] { hipLaunchKernelGGL((k_seq_np<0, 0>), dim3(nseq / 8), dim3(512), 160 * 1024, 0, x, y, nseq, sink); });
            run("non-persistent 8 waves/wg, ONE workgroup per CU (160 KB LDS), nt / nt sc1", bytes, [&] { hipLaunchKernelGGL((k_seq_np<2, 18>), dim3(nseq / 8), dim3(512), 160 * 1024, 0, x, y, nseq, sink); });
            run("non-persistent 8 waves/wg, TWO workgroups per CU (80 KB LDS), nt / nt sc1", bytes, [&] { hipLaunchKernelGGL((k_seq_np<2, 18>), dim3(nseq / 8), dim3(512), 80 * 1024, 0, x, y, nseq, sink); });
#define LD(K) [&] { CK(hipMemsetAsync(t0, 0, 8, 0)); hipLaunchKernelGGL(K, dim3(256), dim3(512), 0, 0, x, y, nseq, sink, (unsigned*)t0); }
            run("dyn claims, runs of 1, nt / nt sc1", bytes, LD((k_dyn<1, 2, 18>)));
            run("dyn claims, runs of 2, nt / nt sc1", bytes, LD((k_dyn<2, 2, 18>)));
            run("dyn claims, runs of 4, nt / nt sc1", bytes, LD((k_dyn<4, 2, 18>)));
            run("dyn claims, runs of 8, nt / nt sc1", bytes, LD((k_dyn<8, 2, 18>)));
            run("dyn claims, runs of 16, nt / nt sc1", bytes, LD((k_dyn<16, 2, 18>)));
            run("dyn claims, runs of 4, default policy", bytes, LD((k_dyn<4, 0, 0>)));
            run("dyn claims, runs of 2, stores 11 x 3 sleep 8", bytes, LD((k_dyn<2, 2, 18, 11, 8>)));
            run("dyn claims, runs of 2, stores 11 x 3 sleep 16", bytes, LD((k_dyn<2, 2, 18, 11, 16>)));
            run("dyn claims, runs of 2, stores 11 x 3 sleep 32", bytes, LD((k_dyn<2, 2, 18, 11, 32>)));
            run("dyn claims, runs of 2, stores 3 x 11 sleep 32", bytes, LD((k_dyn<2, 2, 18, 3, 32>)));
            {
                char* scratch; CK(hipMalloc(&scratch, (size_t)2048 * 34 * 1024)); CK(hipMemset(scratch, 0, (size_t)2048 * 34 * 1024));
#define LDS_(K) [&] { CK(hipMemsetAsync(t0, 0, 8, 0)); hipLaunchKernelGGL(K, dim3(256), dim3(512), 0, 0, x, y, nseq, sink, (unsigned*)t0, scratch); }
                run("dyn 2 + scratch: trickle (sleep 16) to scratch default, burst to y", bytes, LDS_((k_dyn_scratch<16, 0, 0>)));
                run("dyn 2 + scratch: trickle (sleep 16) to scratch sc0, burst loads sc0", bytes, LDS_((k_dyn_scratch<16, 1, 1>)));
                run("dyn 2 + scratch: no sleep, scratch default", bytes, LDS_((k_dyn_scratch<0, 0, 0>)));
                CK(hipFree(scratch));
            }
            run("dyn claims, runs of 2, stores 6 x 6 sleep 16", bytes, LD((k_dyn<2, 2, 18, 6, 16>)));
            run("dyn claims, runs of 2, stores 6 x 6 sleep 32", bytes, LD((k_dyn<2, 2, 18, 6, 32>)));
            run("dyn claims, runs of 2, stores 8 x 5 sleep 16", bytes, LD((k_dyn<2, 2, 18, 8, 16>)));
            run("dyn claims, runs of 2, stores 2 x 17 sleep 64", bytes, LD((k_dyn<2, 2, 18, 2, 64>)));
            run("sync rounds, adjacent, no prefetch, nt / nt sc1", bytes, L((k_sync<false, true, 2, 18>), 8));
            run("sync rounds, adjacent, prefetch, nt / nt sc1", bytes, L((k_sync<true, true, 2, 18>), 8));
            run("sync rounds, far apart, no prefetch, nt / nt sc1", bytes, L((k_sync<false, false, 2, 18>), 8));
            run("sync rounds, far apart, prefetch, nt / nt sc1", bytes, L((k_sync<true, false, 2, 18>), 8));
            run("sync rounds, adjacent, no prefetch, default policy", bytes, L((k_sync<false, true, 0, 0>), 8));
            run("roles W=8 4 + 4, loads 0 stores 0", bytes, L((k_roles<4, 0, 0>), 8));
            run("roles W=8 4 + 4, loads nt stores nt sc1", bytes, L((k_roles<4, 2, 18>), 8));
            run("roles W=8 7 + 1, loads nt stores nt sc1", bytes, L((k_roles<7, 2, 18>), 8));
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) { // twice: box drift shows as a difference between the two passes
        printf("# pass %d, %d sequences\n", rep, nseq);
        run("base W=8 (19 loads, 33 stores, no prefetch)", bytes, L((k_trickle<1, 0, false>), 8));
        run("base W=8, next loads before the stores", bytes, L((k_trickle<1, 0, true>), 8));
        run("base W=4, next loads before the stores", bytes, L((k_trickle<1, 0, true>), 4));
        run("base W=16, next loads before the stores", bytes, L((k_trickle<1, 0, true>), 16));
        run("trickle W=8: 11 groups of 3, sleep 4", bytes, L((k_trickle<11, 4, true>), 8));
        run("trickle W=8: 11 groups of 3, sleep 10", bytes, L((k_trickle<11, 10, true>), 8));
        run("trickle W=8: 11 groups of 3, sleep 20", bytes, L((k_trickle<11, 20, true>), 8));
        run("trickle W=8: 3 groups of 11, sleep 30", bytes, L((k_trickle<3, 30, true>), 8));
        run("half W=12: 11 loads, 17 stores", bytes, L((k_half<1, 0>), 12));
        run("half W=12: 6 groups of 3, sleep 10", bytes, L((k_half<6, 10>), 12));
        run("half W=12: 6 groups of 3, sleep 20", bytes, L((k_half<6, 20>), 12));
        run("half W=8", bytes, L((k_half<1, 0>), 8));
        run("roles W=8: 6 readers + 2 writers", bytes, L((k_roles<6>), 8));
        run("roles W=8: 4 readers + 4 writers", bytes, L((k_roles<4>), 8));
        run("roles W=8: 7 readers + 1 writer", bytes, L((k_roles<7>), 8));
        run("roles W=4: 3 readers + 1 writer", bytes, L((k_roles<3>), 4));
        run("roles W=12: 8 readers + 4 writers", bytes, L((k_roles<8>), 12));
        for (int P : {400, 800, 1600, 3200}) // period in 10-ns ticks
            for (int rdpct : {30, 36, 45}) {
                char nm[96]; snprintf(nm, sizeof nm, "phased W=8: period %.0f us, read window %d %%", P / 100.0, rdpct);
                const int RD = P * rdpct / 100;
                run(nm, bytes, [&] {
                    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, 0, t0);
                    hipLaunchKernelGGL(k_phased, dim3(256), dim3(512), 0, 0, x, y, nseq, sink, t0, P, RD);
                });
            }
        run("non-persistent, 8 waves per workgroup", bytes, [&] { hipLaunchKernelGGL((k_seq_np<0, 0>), dim3(nseq / 8), dim3(512), 0, 0, x, y, nseq, sink); });
        run("non-persistent, 4 waves per workgroup", bytes, [&] { hipLaunchKernelGGL((k_seq_np<0, 0>), dim3(nseq / 4), dim3(256), 0, 0, x, y, nseq, sink); });
        {
            const size_t n = (size_t)nseq * XB / 16; // copy the input buffer's size into y: 2 x 5.03 GB
            run("float4 copy, 256-thread blocks, one element per thread", 2.0 * n * 16,
                [&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float4*)x, (float4*)y, n); });
            run("float4 copy, grid stride 2048 x 256", 2.0 * n * 16,
                [&] { hipLaunchKernelGGL(k_copy_gs, dim3(2048), dim3(256), 0, 0, (const float4*)x, (float4*)y, n); });
            run("float4 copy, grid stride 256 x 512", 2.0 * n * 16,
                [&] { hipLaunchKernelGGL(k_copy_gs, dim3(256), dim3(512), 0, 0, (const float4*)x, (float4*)y, n); });
        }
    }
    return 0;
}
