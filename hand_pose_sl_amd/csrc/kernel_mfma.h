// Exact-fp32 matrix-core kernel (v_mfma_f32_16x16x4_f32) and the pieces it shares with the
// persistent 16-bit kernel (kernel_mfma16.h): precision traits, the swizzled LDS image.
//
// Path: ConvModel.forward, HandPoseModels.py:40-64.  One WAVE owns one chunk of one
// sequence (a whole sequence when T <= kChunk) and carries it through all four layers with
// no workgroup barrier:
//
//   x (B,T,24) fp32 --coalesced 16-B loads--> LDS rows [time][32 ch] fp32 (128-B rows)
//   layer l:  D[chan][time] += W_l[chan][(tap,ch)] * Act[(tap,ch)][time]
//             A = weights (fragment-ordered, registers), B = activations; the accumulator
//             starts from the bias fragment; ReLU and the per-layer zero padding mask
//             (t >= T -> 0) are applied in registers and written back 16 B per lane.
//   layer 4:  42 channels as 3 M-tiles, stored straight to y (B,T,42) fp32.
//
// In-place LDS: layer l+1's input row for time t lives 2 rows BELOW layer l's
// (P(t,l) = t - s + 8 - 2l), so tile m's write-back can never touch a row that tiles > m
// still have to read; one buffer per wave instead of two.  Rows for t < 0 are zeroed once
// and never written; rows T, T+1 are re-zeroed after each layer at the sequence end.
//
// 16-bit LDS image (used by kernel_mfma16.h): 16-B chunk c of physical row P is stored at
// chunk c ^ ((P>>1)&3) (64-B rows): conflict-free for the ds_read_b128 fragment reads at
// every row alignment and for the ds_write_b128 write-back (tools/lds_bank_check.py: x1 / x1
// with the swizzle, x2 / x4 without).
#pragma once
#include "b2h_common.h"
#include "dev/b2h_dev.h" // B2H_ABLATE hooks: constant-false in the shipped build

namespace b2h {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 112;            // output frames per wave-chunk: 18 KB of LDS per wave, so two
                                       // 4-wave workgroups share a CU (2 waves/SIMD); T=200 -> 112 + 88
constexpr int kRows = kChunk + 32;     // 8 halo + chunk + 8 halo + 16 tile overrun
constexpr int kWavesPerBlock = 4;

enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };

template <int PREC> struct Prec;
template <> struct Prec<PREC_BF16> {
    using elem = __bf16; using vec8 = bf16x8; using vec4 = bf16x4;
    static constexpr int kRowBytes = 64;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Prec<PREC_F16> {
    using elem = _Float16; using vec8 = f16x8; using vec4 = f16x4;
    static constexpr int kRowBytes = 64;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Prec<PREC_F32> {
    static constexpr int kRowBytes = 128;
};

// byte offset of 16-B chunk c of physical row P
template <int ROWB> __device__ __forceinline__ int lds_off(int P, int c) {
    if (ROWB == 64) return P * 64 + ((c ^ ((P >> 1) & 3)) << 4);
    return P * 128 + (c << 4); // fp32 rows: MFMA-bound 16x over, no swizzle needed
}

struct ChunkCtx {
    char* lds;      // this wave's LDS buffer
    int lane, tcol, q;
    int T;          // sequence length
    int s, e;       // output frames [s, e) of this chunk
    int64_t seq;    // sequence index
    float* y;       // output base of this sequence (T,42)
    FusedArgs fa;
    int64_t nvalid; // frames kept by the tail mask
};

// Buffer descriptor over [base, base + bytes): loads past the end return 0 and stores
// past the end are dropped by the hardware range check, so one VGPR byte offset
// (plus SGPR/immediate offsets) addresses everything and no lane predicate is needed.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// Cache policy of the streams (the aux operand of the buffer builtins: 1 = sc0, 2 = nt, 16 = sc1).  The rows a
// kernel reads and the rows it writes are touched once per launch and are far larger than L2 + Infinity Cache,
// so they are marked non-temporal, and stores additionally get device scope (sc1), which sends them through to
// memory instead of leaving dirty lines for L2 to evict later in its own order.  Round 3, same-process A/B with
// ONE output buffer (profiles/r3_bf16/ab_cache_policy.txt), bf16 kernel at 262 144 x 200: default 2 725 us,
// nt stores 2 687, nt loads alone 2 726, nt both 2 672, nt loads + nt sc1 stores 2 663 (-2.3 %); at
// 2 000 x 200 26.6 -> 23.4 us (no write-back of dirty lines left for the end of the kernel).  Weights,
// biases and positional tables keep the default policy: they are what L2 is for.
constexpr int kLdStream = 2;
constexpr int kStStream = 18;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, bytes, 0x00020000);
}

// Head epilogue of the wave-per-chunk kernels.  Lane (tcol, q) owns channels 16mt + 4q .. +3 of
// frame t: 16 B at byte 168 (t - s) + 64 mt + 16 q of the chunk's output rows [s, e), stored with
// buffer_store_dwordx4 (rows are 8-byte aligned only, which buffer stores allow); frames >= e fall
// outside the descriptor and are dropped by the range check, channels 40, 41 are an 8-byte store.
// Default cache policy on purpose: these are 64-byte pieces of 128-byte lines that L2 has to merge -- with the
// streaming policy of the lane-linear stores (kStStream) the f16x3 kernel ran 1 548 -> 2 923 us.
// Every offset that varies is in the VGPR offset (immediate soffset), so hipcc pads the store-data
// write-after-read hazard itself (tests/test_isa_audit.py).
// FUSED: x factor (traintest.py:387-388) and the tail mask (utils.py:309-312); the plain
// instantiation carries neither the multiplies nor the selects.
template <bool FUSED> struct HeadStore {
    __amdgpu_buffer_rsrc_t rs;
    int off, t0; // byte offset / frame of this lane in the tile the caller stands at
    float mul;
    __device__ __forceinline__ void init(const ChunkCtx& cx, int lo) { // lo == cx.s for the head
        rs = make_rsrc(cx.y + (int64_t)lo * kOutCh, (cx.e - lo) * (kOutCh * 4));
        off = cx.tcol * (kOutCh * 4) + 16 * cx.q;
        t0 = lo + cx.tcol;
        mul = (cx.fa.flags & kPostDenorm) ? cx.fa.factor : 1.0f; // x 1.0f is exact
    }
    __device__ __forceinline__ void advance2() { off += 2 * (16 * kOutCh * 4); t0 += 32; }
    __device__ __forceinline__ void store(const ChunkCtx& cx, const f32x4 (&acc)[3], int k) { // tile k past the caller's
        const bool dead = FUSED && (int64_t)(t0 + 16 * k) >= cx.nvalid;
        const int vo = off + k * (16 * kOutCh * 4);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            f32x4 v = acc[mt];
            if constexpr (FUSED) {
                v = v * mul;
                if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (mt < 2 || cx.q < 2)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 64 * mt, 0, 0);
            else if (cx.q == 2) // channels 40, 41 (elements passed BY VALUE: see kernel_mfma16.h)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, rs,
                                                      vo + 64 * mt, 0, 0);
        }
    }
};

// ---- pieces shared by the wave-per-chunk kernels (this file, kernel_mfma3.h, kernel_mfma16w.h,
// kernel_mfma3w.h) ---------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() { // one wave's LDS writes visible to its own later reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// This wave's chunk: one wave per (sequence, chunk), `lds_per_wave` bytes of the workgroup's dynamic LDS
// each.  False when the wave has no chunk (tail of the grid).
__device__ __forceinline__ bool chunk_ctx(ChunkCtx& cx, char* smem, int lds_per_wave, float* __restrict__ y, int T,
                                          int chunks_per_seq, int chunk_len, int64_t nchunks, const FusedArgs& fa) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (chunk >= nchunks) return false;
    cx.lds = smem + (size_t)wave * lds_per_wave;
    cx.lane = threadIdx.x & 63;
    cx.tcol = cx.lane & 15;
    cx.q = cx.lane >> 4;
    cx.T = T;
    cx.seq = chunk / chunks_per_seq;
    const int c = (int)(chunk - cx.seq * chunks_per_seq);
    cx.s = c * chunk_len; // <= kChunk frames (the LDS image's capacity); shorter when the batch is small
    cx.e = min(cx.s + chunk_len, T);
    cx.y = y + cx.seq * (int64_t)T * kOutCh;
    cx.fa = fa;
    cx.nvalid = T;
    if ((fa.flags & kPostMask) && fa.n_frames) cx.nvalid = fa.n_frames[cx.seq];
    return true;
}

// Input staging: the chunk's rows [s - 8, e + 8) clipped to the sequence, (T,24) fp32, as 12 lane-linear
// buffer loads per lane, ALL in flight before the first use (128 rows x 6 float4 = 12 x 64; rows past
// the chunk's last one come back as zeros from the range check and are written as such: they are
// either the zero padding behind the sequence end or rows no valid frame depends on), the reference's
// item transforms applied when fused, each float4 (channels 4c4 .. 4c4+3 of physical row P, P(t,0) =
// t - s + 8) handed to put_at(byte offset, w).  Three load iterations cover exactly 32 rows, so a lane
// needs only three (row, column) pairs and three image offsets off_of(P, c4); iteration u = 3G + jj
// writes at off[jj] + 32 G ROWB (every layout's swizzle term is unchanged by a step of 32 rows).
// Then pad(P, pe) once per row for in-positions 24..31 (pos_emb: position 24 = t/100,
// HandPoseModels.py:71-75; the layer-1 weights are packed with the position channel moved to slot 24).
// Returns (first physical row, number of rows, whether the chunk ends at the sequence end).
constexpr int kStageRegs = 12;
static_assert((kChunk + 2 * kHalo) * (kInCh / 4) == kStageRegs * 64, "12 float4 per lane cover the largest chunk");
struct StagedRows { int P0, nrows; bool at_end; };
template <int ROWB, typename OffOf, typename PutAt, typename PutPad>
__device__ __forceinline__ StagedRows stage_rows(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb,
                                                 OffOf off_of, PutAt put_at, PutPad pad) {
    const int in_lo = max(cx.s - kHalo, 0), in_hi = min(cx.e + kHalo, cx.T);
    const int pin = 8 - cx.s;
    const int nf4 = (in_hi - in_lo) * (kInCh / 4);
    const int P0 = in_lo + pin; // 0, or 8 at the sequence start
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(xs + (int64_t)in_lo * kInCh, (B2H_ABLATE & 128) ? 0 : nf4 * 16);
    float4 v[kStageRegs];
#pragma unroll
    for (int u = 0; u < kStageRegs; ++u)
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, cx.lane * 16, u * 1024, kLdStream));
    int rr[3], off[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int i = cx.lane + 64 * jj;
        rr[jj] = i / 6;
        off[jj] = off_of(P0 + rr[jj], i - rr[jj] * 6);
    }
    const int flags = __builtin_amdgcn_readfirstlane(cx.fa.flags);
#pragma unroll
    for (int u = 0; u < kStageRegs; ++u) {
        const int G = u / 3, jj = u % 3;
        float4 w = v[u];
        if (flags & kPreChest) { // body -= body[:,1]  (steps/utils.py:203-210); joint 1 = channels 2, 3 of the row
            const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, (32 * G + rr[jj]) * (kInCh * 4) + 8, 0, 0);
            const float cx_ = __uint_as_float(c[0]), cy_ = __uint_as_float(c[1]);
            w.x -= cx_; w.y -= cy_; w.z -= cx_; w.w -= cy_;
        }
        if (flags & kPreNorm) { // body / factor     (steps/utils.py:180-190)
            w.x = w.x / cx.fa.factor; w.y = w.y / cx.fa.factor;
            w.z = w.z / cx.fa.factor; w.w = w.w / cx.fa.factor;
        }
        put_at(off[jj] + G * 32 * ROWB, w);
    }
    const int nrows = in_hi - in_lo;
    for (int r = cx.lane; r < nrows; r += 64) {
        const int t = in_lo + r;
        pad(t + pin, pos_emb ? (float)t / 100.0f : 0.f);
    }
    return StagedRows{P0, nrows, in_hi == cx.T};
}

// ---- exact-fp32 layers (v_mfma_f32_16x16x4_f32) -------------------------------
// WIDE = false: conv_channels <= 32 (32-channel rows of 128 B, two k-groups of 16 per tap, two M-tiles,
// slot map 8q + 4mt + r, 160 fragment registers, two waves per SIMD).
// WIDE = true : conv_channels 33..64 (64-channel rows of 256 B, four k-groups per tap -- layer 1 has
// 24|25 inputs = two --, four M-tiles, slot map 16q + 4mt + r; a hidden layer's fragments are 320
// registers, so ONE wave per SIMD over the 512-entry register file, one 4-wave workgroup per CU).
// No swizzle on these rows: one ds_read_b128 feeds 16 (32) MFMAs of 32 cycles, LDS is never the limit.
template <bool WIDE> struct Geo32 {
    static constexpr int kRowB = WIDE ? 256 : 128;
    static constexpr int kChunks = kRowB / 16;
    static constexpr int kImg = kRows * kRowB; // LDS bytes per wave: 18 / 36 KB
    static __host__ __device__ constexpr int mt(int L) { return L == 3 ? 3 : (WIDE ? 4 : 2); }
    static __host__ __device__ constexpr int groups(int L) { return (WIDE && L > 0) ? 4 : 2; }
    static __device__ __forceinline__ int off(int P, int c) { return P * kRowB + (c << 4); }
};

template <int L, bool FUSED, bool WIDE>
__device__ __forceinline__ void layer32(const ChunkCtx& cx, const MfmaParams& mp) {
    using G = Geo32<WIDE>;
    constexpr int MT = G::mt(L), NG = G::groups(L), MTH = G::mt(0);
    constexpr int h = 6 - 2 * L;
    const int lo = max(cx.s - h, 0), hi = min(cx.e + h, cx.T);
    const int ntiles = (hi - lo + 15) >> 4;

    f32x4 A[MT][kTaps][NG]; // [.][tap][g][j]: in-channel 16g + 4q + j
    f32x4 bias[MT];
    {
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(mp.w[L], MT * kTaps * NG * 1024); // [mt][tap][g][lane] x 16 B
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    A[mt][s][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                wrs, cx.lane * 16, ((mt * kTaps + s) * NG + g) * 1024, 0));
        const f32x4* bp = reinterpret_cast<const f32x4*>(mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + cx.q];
    }
    const int pin = 8 - 2 * L - cx.s;
    const int pout = pin - 2;
    HeadStore<FUSED> hs;
    if constexpr (L == 3) hs.init(cx, lo);

#pragma unroll 1
    for (int m = 0; m < ntiles; ++m) {
        const int tau = lo + 16 * m;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            const int Pr = tau + cx.tcol + s - kPad + pin;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(cx.lds + G::off(Pr, 4 * g + cx.q));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][s][g][j], bv[j], acc[mt], 0, 0, 0);
            }
        }
        const int t = tau + cx.tcol;
        if constexpr (L < 3) {
            f32x4 o[MTH];
#pragma unroll
            for (int mt = 0; mt < MTH; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) // max(v, 0) as one v_max_i32 on the bits
                    o[mt][r] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, (float)acc[mt][r]), 0));
            if (tau + 16 > cx.T) { // only the tile that crosses the sequence end: frames >= T are padding
                const bool inside = t < cx.T;
#pragma unroll
                for (int mt = 0; mt < MTH; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[mt][r] = inside ? o[mt][r] : 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < MTH; ++mt) // channels (8|16)q + 4mt .. +3  ->  16-B chunk (2|4)q + mt
                *reinterpret_cast<f32x4*>(cx.lds + G::off(t + pout, MTH * cx.q + mt)) = o[mt];
        } else {
            hs.store(cx, acc, 0);
            hs.off += 16 * kOutCh * 4;
            hs.t0 += 16;
        }
    }
    if constexpr (L < 3) {
        if (hi == cx.T) { // rows T, T+1 of the next layer's input: zero unless a tile covered them
            const int covered = lo + 16 * ntiles;
            const int t = cx.T + cx.lane / G::kChunks;
            if (cx.lane < 2 * G::kChunks && t >= covered)
                *reinterpret_cast<f32x4*>(cx.lds + G::off(t + pout, cx.lane % G::kChunks)) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wave_lds_sync();
    }
}

// ---- input staging: (T,24) fp32 rows -> LDS image of layer-1 input ------------
template <bool WIDE>
__device__ __forceinline__ void stage_input32(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb) {
    using G = Geo32<WIDE>;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const StagedRows st = stage_rows<G::kRowB>(
        cx, xs, pos_emb, [&](int P, int c4) { return G::off(P, c4); },
        [&](int off, float4 w) { *reinterpret_cast<float4*>(cx.lds + off) = w; },
        [&](int P, float pe) { // channel padding 24..31 = chunks 6, 7 (layer 1 reads chunks 0..7 only)
            *reinterpret_cast<f32x4*>(cx.lds + G::off(P, 6)) = f32x4{pe, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(cx.lds + G::off(P, 7)) = z4;
        });
    // zero rows (whole rows: later layers read them whole): t in [-8,0) at the sequence start (all
    // layers' low padding) and t = T, T+1 at its end
    if (cx.s == 0)
        for (int i = cx.lane; i < 8 * G::kChunks; i += 64)
            *reinterpret_cast<f32x4*>(cx.lds + G::off(i / G::kChunks, i % G::kChunks)) = z4;
    if (st.at_end && cx.lane < 2 * G::kChunks)
        *reinterpret_cast<f32x4*>(cx.lds + G::off(st.P0 + st.nrows + cx.lane / G::kChunks, cx.lane % G::kChunks)) = z4;
    wave_lds_sync();
}

// One wave per (sequence, chunk); no workgroup barrier anywhere.  Narrow: two 4-wave workgroups per CU by
// LDS = 2 waves per SIMD, each within 256 VGPRs (keeps the accumulators out of AGPRs).  Wide: one.
template <bool FUSED, bool WIDE>
__global__ __launch_bounds__(64 * kWavesPerBlock, WIDE ? 1 : 2) void b2h_fwd_mfma_f32(
    const float* __restrict__ x, float* __restrict__ y, int T, int chunks_per_seq, int chunk_len,
    int64_t nchunks, MfmaParams mp, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem_mfma[];
    ChunkCtx cx;
    if (!chunk_ctx(cx, smem_mfma, Geo32<WIDE>::kImg, y, T, chunks_per_seq, chunk_len, nchunks, fa)) return;
    stage_input32<WIDE>(cx, x + cx.seq * (int64_t)T * kInCh, mp.pos_emb);
    layer32<0, FUSED, WIDE>(cx, mp); layer32<1, FUSED, WIDE>(cx, mp);
    layer32<2, FUSED, WIDE>(cx, mp); layer32<3, FUSED, WIDE>(cx, mp);
}

} // namespace b2h
