// fp32-grade matrix-core kernel for WIDE models (conv_channels 33..64): the 3 x f16 split of
// kernel_mfma3.h on the 64-channel geometry of kernel_mfma16w.h.
//
// Path: ConvModel.forward, HandPoseModels.py:40-64; `--conv-channels` is a free integer (run.py:37).
//   arithmetic : every activation and weight split x = hi + lo in f16 (22 significant bits), every
//                product three v_mfma_f32_16x16x32_f16 (Wlo.xhi + Whi.xlo + Whi.xhi), fp32 accumulate;
//                valid while |activation|, |weight| < 65504
//   geometry   : hi and lo LDS images with rows [time][64 ch] of f16 = 128 B each (chunk c of row P at
//                c ^ (P & 7), as kernel_mfma16w.h), one tap = two k-steps, four M-tiles (head: three),
//                hidden out-channel slot (mt, 4q + r) <-> channel 16q + 4mt + r
//   registers  : a hidden layer's hi + lo weight fragments are 4 x 5 x 2 x 2 x 4 = 320 registers, so the
//                kernel runs ONE wave per SIMD (launch bound 256 threads, 512-entry register file;
//                hipcc parks ~100 of them in AGPRs and copies them back per tile, <= 1 copy per MFMA,
//                which is free beside an MFMA: tools/mfma_mix_bench.hip).  LDS 36 KB per wave, one
//                4-wave workgroup per CU.
// 390 MFMAs per 16 frames (narrow f16x3: 135): matrix-pipe-bound; ~8x the exact-fp32 VALU kernel that
// was the only fp32-grade path for these widths.
#pragma once
#include "kernel_mfma16w.h" // lds_offw, wide_chan_of, wide_mt, wide_ks
#include "kernel_mfma3.h"   // split8

namespace b2h {

constexpr int kImg3W = kRows * kWideRowB; // bytes of one image (hi or lo) of a wave: 18 KB

template <int L, bool FUSED>
__device__ __forceinline__ void layer3w(const ChunkCtx& cx, const MfmaParams& mp) {
    constexpr int MT = wide_mt(L), KS = wide_ks(L);
    constexpr int h = 6 - 2 * L;
    const int lo = max(cx.s - h, 0), hi = min(cx.e + h, cx.T);
    const int ntiles = (hi - lo + 15) >> 4;
    char* img_h = cx.lds;
    char* img_l = cx.lds + kImg3W;

    f16x8 Ah[MT][kTaps][KS], Al[MT][kTaps][KS]; // in-positions 32ks + 8q + j of out-channel slot (lane & 15)
    f32x4 bias[MT];
    {
        // [mt][tap][ks][hi|lo][lane] x 16 B, buffer loads (descriptor + lane offset: see kernel_mfma16w.h)
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(mp.w[L], MT * kTaps * KS * 2048);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int f = ((mt * kTaps + s) * KS + ks) * 2048;
                    Ah[mt][s][ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, cx.lane * 16, f, 0));
                    Al[mt][s][ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, cx.lane * 16, f + 1024, 0));
                }
        const f32x4* bp = reinterpret_cast<const f32x4*>(mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + cx.q];
    }
    const int pin = 8 - 2 * L - cx.s;
    const int pout = pin - 2;
    // swizzled byte offsets of this lane's tap rows and of its write-back row in tile 0 (chunk 4 + q of a
    // row = chunk q's offset ^ 64, chunk 2q + 1 = chunk 2q's ^ 16); a tile step is 16 rows = 2048 B
    int roff[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) roff[s] = lds_offw(lo + cx.tcol + s - kPad + pin, cx.q);
    const int woff = lds_offw(lo + cx.tcol + pout, 2 * cx.q);
    HeadStore<FUSED> hs;
    if constexpr (L == 3) hs.init(cx, lo);

#pragma unroll 1
    for (int m = 0; m < ntiles; ++m) {
        const int tau = lo + 16 * m;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int o = (roff[s] + m * (16 * kWideRowB)) ^ (64 * ks);
                const f16x8 bh = *reinterpret_cast<const f16x8*>(img_h + o);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(img_l + o);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mt][s][ks], bh, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s][ks], bl, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s][ks], bh, acc[mt], 0, 0, 0);
            }
        if constexpr (L < 3) {
            const bool crossing = tau + 16 > cx.T; // only this tile can hold frames >= T (next layer's padding)
            const bool inside = tau + cx.tcol < cx.T;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) { // channels 16q + 8hh .. +7 = M-tiles 2hh, 2hh + 1 = chunk 2q + hh
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_bits(acc[2 * hh + (j >> 2)][j & 3]);
                if (crossing) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
                }
                f16x8 oh, ol;
                split8(v, oh, ol);
                const int o = (woff + m * (16 * kWideRowB)) ^ (16 * hh);
                *reinterpret_cast<f16x8*>(img_h + o) = oh;
                *reinterpret_cast<f16x8*>(img_l + o) = ol;
            }
        } else {
            hs.store(cx, acc, 0);
            hs.off += 16 * kOutCh * 4;
            hs.t0 += 16;
        }
    }
    if constexpr (L < 3) {
        if (hi == cx.T) { // rows T, T+1 of the next layer's input: zero unless a tile covered them
            const int covered = lo + 16 * ntiles;
            const int t = cx.T + ((cx.lane >> 3) & 1);
            if (cx.lane < 32 && t >= covered) // 2 rows x 8 chunks x 2 images
                *reinterpret_cast<f32x4*>((cx.lane < 16 ? img_h : img_l) + lds_offw(t + pout, cx.lane & 7)) =
                    f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wave_lds_sync();
    }
}

// ---- input staging: (T,24) fp32 rows -> hi / lo images of the layer-1 input (chunks 0..3) ---------
__device__ __forceinline__ void stage_input3w(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb) {
    char* img_h = cx.lds;
    char* img_l = cx.lds + kImg3W;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const StagedRows st = stage_rows<kWideRowB>(
        cx, xs, pos_emb, // channels 4c4 .. 4c4+3: half (c4 & 1) of 16-B chunk c4 >> 1
        [&](int P, int c4) { return lds_offw(P, c4 >> 1) + (c4 & 1) * 8; },
        [&](int off, float4 w) {
            f16x4 wh, wl;
            split4(w, wh, wl);
            *reinterpret_cast<f16x4*>(img_h + off) = wh;
            *reinterpret_cast<f16x4*>(img_l + off) = wl;
        },
        [&](int P, float pe) { // in-positions 24..31 = chunk 3; layer 1 reads chunks 0..3 only
            const _Float16 ph = (_Float16)pe;
            f16x8 zh, zl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { zh[j] = (_Float16)0.f; zl[j] = (_Float16)0.f; }
            zh[0] = ph;
            zl[0] = (_Float16)(pe - (float)ph);
            *reinterpret_cast<f16x8*>(img_h + lds_offw(P, 3)) = zh;
            *reinterpret_cast<f16x8*>(img_l + lds_offw(P, 3)) = zl;
        });
    // zero rows (all 8 chunks of both images): t in [-8,0) at the sequence start (every layer's low
    // padding) and t = T, T+1 at the sequence end
    if (cx.s == 0) {
        *reinterpret_cast<f32x4*>(img_h + lds_offw(cx.lane >> 3, cx.lane & 7)) = z4;
        *reinterpret_cast<f32x4*>(img_l + lds_offw(cx.lane >> 3, cx.lane & 7)) = z4;
    }
    if (st.at_end && cx.lane < 32) // 2 rows x 8 chunks x 2 images
        *reinterpret_cast<f32x4*>((cx.lane < 16 ? img_h : img_l) + lds_offw(st.P0 + st.nrows + ((cx.lane >> 3) & 1), cx.lane & 7)) = z4;
    wave_lds_sync();
}

// One wave per (sequence, chunk); no workgroup barrier anywhere.  One 4-wave workgroup per CU.
template <bool FUSED>
__global__ __launch_bounds__(64 * kWavesPerBlock, 1) void b2h_fwd_mfma_f16x3w(
    const float* __restrict__ x, float* __restrict__ y, int T, int chunks_per_seq, int chunk_len,
    int64_t nchunks, MfmaParams mp, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem_mfma3w[];
    ChunkCtx cx;
    if (!chunk_ctx(cx, smem_mfma3w, 2 * kImg3W, y, T, chunks_per_seq, chunk_len, nchunks, fa)) return;
    stage_input3w(cx, x + cx.seq * (int64_t)T * kInCh, mp.pos_emb);
    layer3w<0, FUSED>(cx, mp); layer3w<1, FUSED>(cx, mp); layer3w<2, FUSED>(cx, mp); layer3w<3, FUSED>(cx, mp);
}

} // namespace b2h
