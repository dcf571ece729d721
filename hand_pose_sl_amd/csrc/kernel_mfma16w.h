// bf16/f16 matrix-core kernel for WIDE models: conv_channels in 33..64.
//
// Path: ConvModel.forward, HandPoseModels.py:40-64; `--conv-channels` is a free integer
// (run.py:37, HandPoseModels.py:18,24-32).  Same mapping as the other wave-per-chunk kernels
// (kernel_mfma.h: one wave owns one chunk of one sequence and carries it through all four layers,
// weights in registers, in-place LDS image, no workgroup barrier) with twice the channel padding:
//
//   LDS image  : rows [time][64 ch] of 16-bit = 128 B, 16-B chunk c of physical row P stored at
//                chunk c ^ (P & 7): conflict-free for the ds_read_b128 fragment reads at every row
//                alignment and for the ds_write_b128 write-back (searched over all XOR swizzles
//                with the lane groups of MI355X_MICROARCH.md; a 16-row tile step leaves it unchanged)
//   layer l    : D[slot][time] += W_l[slot][(tap, pos)] . Act[(tap, pos)][time]; one tap = TWO
//                16x16x32 k-steps (in-positions 0..31, 32..63; layer 1 has 24|25 inputs = one),
//                4 M-tiles of out-channel slots (head: 3 = 42 -> 48).  All of a layer's weight
//                fragments stay in registers: 4 x 5 x 2 x 4 = 160 VGPRs, which still leaves two
//                waves per SIMD (hipcc: ~200 registers), so no second pass and no second image.
//   slot map   : hidden out-channel (M-tile mt, row 4q+r) <-> channel 16q + 4mt + r, so a lane's 16
//                results are the 16 consecutive channels of chunks 2q, 2q+1 of the next layer's
//                row: two ds_write_b128 per lane and tile.
//
// Weights come from L2 once per layer and chunk (40 KB for a hidden layer at 64 channels); the
// kernel is matrix-pipe-bound (130 MFMAs per 16 frames against 45 at <= 32 channels).
#pragma once
#include "kernel_mfma.h"
#include "kernel_mfma16.h" // pack2, relu_bits

namespace b2h {

constexpr int kWideRowB = 128;                 // bytes per LDS row: 64 channels x 16 bit
constexpr int kWideMT = 4;                     // M-tiles of a hidden layer (64 out-channel slots)
constexpr int kImgW = kRows * kWideRowB;       // LDS bytes per wave (18 KB: 2 x 4 waves per CU)

__host__ __device__ inline int wide_chan_of(int mt, int row) { return 16 * (row >> 2) + 4 * mt + (row & 3); }

__device__ __forceinline__ int lds_offw(int P, int c) { return P * kWideRowB + ((c ^ (P & 7)) << 4); }

// fragment counts / offsets of the packed weights: per layer [mt][tap][ks][lane] x 16 B
__host__ __device__ constexpr int wide_mt(int L) { return L == 3 ? 3 : kWideMT; }
__host__ __device__ constexpr int wide_ks(int L) { return L == 0 ? 1 : 2; }

template <int PREC, int L, bool FUSED>
__device__ __forceinline__ void layer16w(const ChunkCtx& cx, const MfmaParams& mp) {
    using P = Prec<PREC>;
    using vec8 = typename P::vec8;
    constexpr int MT = wide_mt(L), KS = wide_ks(L);
    constexpr int h = 6 - 2 * L;
    const int lo = max(cx.s - h, 0), hi = min(cx.e + h, cx.T);
    const int ntiles = (hi - lo + 15) >> 4;

    vec8 A[MT][kTaps][KS]; // in-positions 32ks + 8q + j of out-channel slot (lane & 15)
    f32x4 bias[MT];
    {
        // buffer loads: one descriptor in SGPRs + the lane offset, fragment offsets as immediates /
        // soffset -- 64-bit per-fragment addresses (40 KB span) cost 20 VGPRs and spilled
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(mp.w[L], MT * kTaps * KS * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    A[mt][s][ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(
                                                                wrs, cx.lane * 16, ((mt * kTaps + s) * KS + ks) * 1024, 0));
        const f32x4* bp = reinterpret_cast<const f32x4*>(mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + cx.q];
    }
    const int pin = 8 - 2 * L - cx.s;
    const int pout = pin - 2;
    // Fragment and write-back addresses as 32-bit LDS pointers of the tile the loop stands at (chunk 4 + q of
    // a row sits at chunk q's offset ^ 64, which is +-64 depending on the row, so the second k-step cannot be
    // an immediate: one XOR where it is read; likewise chunk 2q + 1 = chunk 2q's ^ 16).  A tile step is 16 rows = 2048 B and
    // leaves the swizzle term (P & 7) unchanged, so the pointers advance once per TWO tiles, opaque to the
    // compiler, and every access in between is register + immediate (round 2's loop re-derived all twenty
    // fragment addresses per tile: two vector instructions each).
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) const vec8 lds_cvec8;
    typedef __attribute__((address_space(3))) u32x4 lds_u4;
    lds_char* rp[kTaps]; // k-step 0; k-step 1 is this ^ 64, formed where it is used (a pointer per k-step spilled)
#pragma unroll
    for (int s = 0; s < kTaps; ++s) rp[s] = (lds_char*)(cx.lds + lds_offw(lo + cx.tcol + s - kPad + pin, cx.q));
    lds_char* wp = (lds_char*)(cx.lds + lds_offw(lo + cx.tcol + pout, 2 * cx.q));
    auto flip = [](lds_char* p, unsigned bits) { return (lds_char*)(uintptr_t)((unsigned)(uintptr_t)p ^ bits); };
    int tq = lo + cx.tcol; // this lane's frame in the tile the loop stands at
    HeadStore<FUSED> hs;
    if constexpr (L == 3) hs.init(cx, lo);

    // Fragments travel through a RING of five registers sets, four k-steps ahead of their use (4 x 4 MFMAs =
    // 256 cycles, the LDS latency under load): the loop body is two tiles = 20 (layer 1: 10) k-steps, a
    // multiple of five, so every ring slot is a register NAME.  A whole tile's fragments a tile ahead (40
    // registers beside the 160 of the weights) spilled; five sets are 20.  Past a layer's last tile the ring
    // reads rows nobody uses (LDS reads beyond the allocation return 0).
    constexpr int NS = kTaps * KS, kDist = 4, kRing = 5;
    static_assert((2 * NS) % kRing == 0, "ring slots must be static across the two-tile body");
    vec8 Br[kRing];
    f32x4 acc[MT];
    auto frag = [&](int step, int tile) { // k-step `step` of the tile `tile` tiles past the pointers
        const int s = step / KS, ks = step % KS;
        return *(lds_cvec8*)((ks ? flip(rp[s], 64u) : rp[s]) + tile * (16 * kWideRowB));
    };
    // epilogue of the tile `k` tiles past the one the pointers stand at; `mask`: only a layer's LAST tile
    // can hold frames >= T (the zero padding of the next layer)
    auto epi = [&](int k, bool mask) {
        if constexpr (L < 3) {
            float v[16]; // channels 16q + 4mt + r = slot 4mt + r of this lane's two chunks
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
            if (mask) {
                const bool inside = tq + 16 * k < cx.T;
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = inside ? v[j] : 0.f;
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const u32x4 o = {pack2<PREC>(v[8 * hh], v[8 * hh + 1]), pack2<PREC>(v[8 * hh + 2], v[8 * hh + 3]),
                                 pack2<PREC>(v[8 * hh + 4], v[8 * hh + 5]), pack2<PREC>(v[8 * hh + 6], v[8 * hh + 7])};
                *(lds_u4*)((hh ? flip(wp, 16u) : wp) + k * (16 * kWideRowB)) = o;
            }
        } else {
            hs.store(cx, acc, k);
        }
    };
    auto advance2 = [&]() {
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            rp[s] += 2 * 16 * kWideRowB;
            asm volatile("" : "+v"(rp[s]));
        }
        wp += 2 * 16 * kWideRowB;
        asm volatile("" : "+v"(wp));
        tq += 32;
        if constexpr (L == 3) hs.advance2();
    };
    // the two-tile body: tiles k = 0, 1 past the pointers (`two` = false: only tile 0), masks for a last tile
    auto body = [&](bool two, bool mask0, bool mask1) {
#pragma unroll
        for (int j = 0; j < 2 * NS; ++j) {
            const int k = j / NS, jj = j % NS;
            if (k == 1 && !two) break;
            if (jj == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
            }
            Br[(j + kDist) % kRing] = frag((j + kDist) % NS, (j + kDist) / NS); // slot of k-step j - 1: already issued
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = P::mfma(A[mt][jj / KS][jj % KS], Br[j % kRing], acc[mt]);
            if (jj == NS - 1) epi(k, k == 0 ? mask0 : mask1);
        }
    };
#pragma unroll
    for (int j = 0; j < kDist; ++j) Br[j] = frag(j, 0);
    int m = 0;
#pragma unroll 1
    for (; m + 2 < ntiles; m += 2) { // two tiles, neither the layer's last: no mask, no branch
        body(true, false, false);
        advance2();
    }
    if (m + 1 < ntiles) body(true, false, true);
    else body(false, true, false);
    if constexpr (L < 3) {
        if (hi == cx.T) { // rows T, T+1 of the next layer's input: zero unless a tile covered them
            const int covered = lo + 16 * ntiles;
            const int t = cx.T + (cx.lane >> 3);
            if (cx.lane < 16 && t >= covered)
                *reinterpret_cast<uint4*>(cx.lds + lds_offw(t + pout, cx.lane & 7)) = uint4{0u, 0u, 0u, 0u};
        }
        wave_lds_sync();
    }
}

// ---- input staging: (T,24) fp32 rows -> 16-bit image of the layer-1 input (chunks 0..3) ----------
template <int PREC>
__device__ __forceinline__ void stage_input16w(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb) {
    const uint4 z4 = {0u, 0u, 0u, 0u};
    const StagedRows st = stage_rows<kWideRowB>(
        cx, xs, pos_emb, // channels 4c4 .. 4c4+3: half (c4 & 1) of 16-B chunk c4 >> 1
        [&](int P, int c4) { return lds_offw(P, c4 >> 1) + (c4 & 1) * 8; },
        [&](int off, float4 w) {
            *reinterpret_cast<uint2*>(cx.lds + off) = uint2{pack2<PREC>(w.x, w.y), pack2<PREC>(w.z, w.w)};
        },
        [&](int P, float pe) { // in-positions 24..31 = chunk 3; layer 1 reads chunks 0..3 only
            uint4 z = z4;
            z.x = pack2<PREC>(pe, 0.f);
            *reinterpret_cast<uint4*>(cx.lds + lds_offw(P, 3)) = z;
        });
    // zero rows (all 8 chunks: later layers read them whole): t in [-8,0) at the sequence start (every
    // layer's low padding) and t = T, T+1 at the sequence end
    if (cx.s == 0) *reinterpret_cast<uint4*>(cx.lds + lds_offw(cx.lane >> 3, cx.lane & 7)) = z4;
    if (st.at_end && cx.lane < 16)
        *reinterpret_cast<uint4*>(cx.lds + lds_offw(st.P0 + st.nrows + (cx.lane >> 3), cx.lane & 7)) = z4;
    wave_lds_sync();
}

// One wave per (sequence, chunk); no workgroup barrier anywhere.  18 KB of LDS per wave: two 4-wave
// workgroups per CU = 2 waves per SIMD, each within 256 VGPRs.
template <int PREC, bool FUSED>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void b2h_fwd_mfma16w(
    const float* __restrict__ x, float* __restrict__ y, int T, int chunks_per_seq, int chunk_len,
    int64_t nchunks, MfmaParams mp, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem_mfma16w[];
    ChunkCtx cx;
    if (!chunk_ctx(cx, smem_mfma16w, kImgW, y, T, chunks_per_seq, chunk_len, nchunks, fa)) return;
    stage_input16w<PREC>(cx, x + cx.seq * (int64_t)T * kInCh, mp.pos_emb);
    layer16w<PREC, 0, FUSED>(cx, mp); layer16w<PREC, 1, FUSED>(cx, mp);
    layer16w<PREC, 2, FUSED>(cx, mp); layer16w<PREC, 3, FUSED>(cx, mp);
}

} // namespace b2h
