// Development-only instrumentation of the gfx950 kernels.  Nothing in this header is part of the
// shipped library: with the default B2H_ABLATE = 0 every hook below is an empty statement or a
// constant-false condition and no stamp buffer or debug export exists in libb2h.so.
//
//   B2H_ABLATE=<bits> python -m hand_pose_sl_amd.build --force
//
// builds a timing / tracing variant (results may be wrong by construction; never shipped):
//   TransformerEnc chain kernel (tools/ablate_tenc.sh): 256 no LayerNorm math, 1024 no blob staging,
//     2048 no per-stage barrier, 4096 no stores, 8192 no MFMA,
//     16384 s_memtime stamps of one workgroup's phases (tools/chain_stamps.py);
//   f16x3 conv kernel: 32768 s_memtime stamps of one wave's phases (tools/conv3_stamps.py),
//     64 one weight fragment per layer instead of 20-30 (prices the per-chunk weight reloads from L2),
//     128 no input loads;
//   persistent 16-bit conv kernel: 65536 s_memtime stamps of one workgroup's phases (tools/conv16_stamps.py).
#pragma once
#ifndef B2H_ABLATE
#define B2H_ABLATE 0
#endif

namespace b2h {

#if B2H_ABLATE & 32768
__device__ unsigned long long g_conv3_dbg[4 * 16];
#define B2H_STAMP3(cx, k)                                                                                   \
    do {                                                                                                    \
        if (blockIdx.x == gridDim.x / 2 && (cx).lane == 0) g_conv3_dbg[(threadIdx.x >> 6) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define B2H_STAMP3(cx, k) do { } while (0)
#endif

// persistent 16-bit conv kernel: 65536 s_memtime stamps of ONE workgroup's eight waves at the phase
// boundaries of their first 32 chunks (tools/conv16_stamps.py): [wave][chunk][8]
#if B2H_ABLATE & 65536
__device__ unsigned long long g_conv16_dbg[8 * 32 * 8];
#define B2H_STAMP16(wave, lane, it, k)                                                                      \
    do {                                                                                                    \
        if (blockIdx.x == gridDim.x / 2 && (lane) == 0 && (it) >= 40 && (it) < 72)                          \
            g_conv16_dbg[((wave) * 32 + (int)((it) - 40)) * 8 + (k)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
// ... and s_memrealtime (100 MHz, chip-wide) at every wave's entry and exit plus its chunk count: [wg][wave][3]
__device__ unsigned long long g_conv16_span[256 * 8 * 3];
#define B2H_SPAN16(wave, lane, slot, val)                                                                   \
    do {                                                                                                    \
        if ((lane) == 0 && blockIdx.x < 256) g_conv16_span[(blockIdx.x * 8 + (wave)) * 3 + (slot)] = (val); \
    } while (0)
#else
#define B2H_STAMP16(wave, lane, it, k) do { } while (0)
#define B2H_SPAN16(wave, lane, slot, val) do { } while (0)
#endif

#if B2H_ABLATE & 16384
__device__ unsigned long long g_chain_dbg[8 * 64];
#define B2H_STAMP()                                                                                        \
    do {                                                                                                   \
        if (blockIdx.x == gridDim.x / 2 && lane == 0 && a.nstages == 3 && nstamp < 64)                     \
            g_chain_dbg[wave * 64 + nstamp] = __builtin_amdgcn_s_memtime();                                \
        ++nstamp;                                                                                          \
    } while (0)
#else
#define B2H_STAMP() do { } while (0)
#endif

} // namespace b2h
