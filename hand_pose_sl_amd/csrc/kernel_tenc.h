// TransformerEnc (body2hand/src/models/HandPoseModels.py:118-178) on gfx950.
// Two kernels per precision:
//
//   b2h_attn_mfma_f32 / b2h_attn_mfma_h3
//                       self-attention, workgroup = (sequence, head), wave = 16 query frames:
//                       scores and P.V on the matrix cores, softmax in registers over all T keys
//                       (the reference passes no mask, HandPoseModels.py:170).
//   b2h_tenc_chain<H3>  everything between two attention calls, which is all per-frame: a wave
//                       carries 16 frames through a chain of Linear layers in registers (the
//                       accumulator layout IS the next GEMM's operand layout), with bias, ReLU,
//                       residual + LayerNorm fused between them; weights double-buffered in LDS.
//
// B2H_TENC_F32 computes every product in fp32 (v_mfma_f32_16x16x4_f32); B2H_TENC_F16X3 (the _h3 /
// <true> instantiations) splits every operand into f16 hi + lo and uses three
// v_mfma_f32_16x16x32_f16 per product with fp32 accumulation: fp32-grade results at 3/16 of the
// matrix cycles.  Softmax, LayerNorm, bias and residual arithmetic is fp32 in both.
//
// Per forward: 1 front chain (src + pe -> pose2hidden -> Q,K,V) and per layer 1 attention + 1
// chain launch; only the residual stream, the attention output and Q,K,V cross HBM (2.5 KB per
// frame of caller-provided workspace).
//
// All kernels address global memory with buffer instructions over per-workgroup descriptors:
// the hardware range check replaces lane predicates, which keeps every s_waitcnt a counted one.
// On gfx950 MFMA and VALU work do not overlap on a SIMD (tools/mfma_bench.hip), so there is
// no attempt to hide epilogues under MFMAs: the bound is their sum.
#pragma once
#include <utility>
#include "b2h_common.h"
#include "kernel_mfma.h"   // f32x4
#include "kernel_mfma16.h" // pack helpers
#include "dev/b2h_dev.h"   // B2H_ABLATE / B2H_STAMP hooks: constant-false / empty in the shipped build

namespace b2h {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kTencD = 128;      // nhid (d_model and feed-forward width in the reference's CLIs)
constexpr int kTencHeads = 4;
constexpr int kTencHd = 32;      // head dim
constexpr int kLinWaves = 8;     // waves per workgroup of the linear kernel (16 frames each)
constexpr int kLinChunkMT = 8;   // M-tiles (x16 features) of weights staged in LDS at a time

// Sum / max over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} that hold one frame's values, left in all four.
// Two __shfl_xor = ds_bpermute_b32 + s_waitcnt lgkmcnt(0) each.  Round 3 tried v_permlane16_swap /
// v_permlane32_swap (gfx950) on two copies of the value instead (swap(a, b) exchanges a's odd rows or upper half
// with b's even rows or lower half, so a + b is the pairwise sum in every lane): correct as inline asm with the
// two wait states the swap needs after a vector write of its operands (tools/permlane_probe.hip; the BUILTINS of
// this hipcc return the first result twice: 0 of 64 lanes right), but 0.8 % SLOWER on the f16x3 forward in a
// same-process A/B (profiles/r3_tenc/ab_gemm_units_and_permlane.txt): the shuffle's latency was already hidden
// by the SIMD's other wave, the swaps' nops and issue slots are not.
__device__ __forceinline__ float quad_sum(float s) {
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}
__device__ __forceinline__ float quad_max(float s) {
    s = fmaxf(s, __shfl_xor(s, 16, 64));
    s = fmaxf(s, __shfl_xor(s, 32, 64));
    return s;
}

// Self-attention on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32): softmax(q k^T) v with
// q pre-scaled by head_dim^-0.5 (torch.nn.MultiheadAttention).
//   qkv : (B*T, 384) = [q | k | v] x 128, head h = columns h*32 .. h*32+31 of each third
//   out : (B*T, 128), head h -> columns h*32 ..  Workgroup = (sequence,
// head); wave w owns query tile w (16 frames) against all key tiles:
//   S^T[key][query] = K[key][d] . (Q[query][d] * 32^-0.5)      8 MFMAs per 16x16 tile
//   softmax over keys: the 4*ntiles scores of a query sit in ONE lane quartet
//                      (registers + lanes l, l^16, l^32), keys >= T masked to -inf
//   O^T[d][query] = V^T[d][key] . P^T[key][query]: the score tile's accumulator IS the B operand
//                      (D row 4q+r  <->  B k-index q for fixed r): no shuffle, no LDS round trip
// K and V of the head live in LDS as [key][kAttnRow] fp32 with kAttnRow = 36: the 4-float pad
// spreads the 16 key rows of a fragment read (ds_read_b128, K) and the 4 row groups of a V column
// read (ds_read_b32) over all banks (a 32-float row puts them on the same ones).  Rows T..16*NT
// are zero: every global access is a buffer instruction over the sequence's rows, whose range
// check returns 0 past T and drops stores, so nothing is predicated and all loads of the
// workgroup are in flight together.
constexpr int kAttnMaxTiles = 8; // T <= 128
constexpr int kAttnRow = 36;

template <int NT> // number of 16-frame tiles = ceil(T / 16): compile-time so that the loops are branch-free
__global__ __launch_bounds__(64 * NT) void b2h_attn_mfma_f32(const float* __restrict__ qkv,
                                                          float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem_attn2[];
    float* Ks = reinterpret_cast<float*>(smem_attn2);
    float* Vs = Ks + NT * 16 * kAttnRow;
    const int b = blockIdx.x / kTencHeads, h = blockIdx.x % kTencHeads;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + (int64_t)b * T * (3 * kTencD), T * 3 * kTencD * 4);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const int tq = wave * 16 + col;
    // K, V rows of the head: NT*16 rows x 8 float4 = two per thread each; and this lane's query
    f32x4 k4[2], v4[2], qb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = threadIdx.x + it * 64 * NT, t = i >> 3, c = i & 7;
        const int off = (t * 3 * kTencD + h * kTencHd + 4 * c) * 4;
        k4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, kTencD * 4, 0));
        v4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 2 * kTencD * 4, 0));
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) // B operand of S^T: d = 16g + 4q + j
        qb[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs, (tq * 3 * kTencD + h * kTencHd + 16 * g + 4 * q) * 4, 0, 0));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = threadIdx.x + it * 64 * NT, t = i >> 3, c = i & 7;
        *reinterpret_cast<f32x4*>(Ks + t * kAttnRow + 4 * c) = k4[it];
        *reinterpret_cast<f32x4*>(Vs + t * kAttnRow + 4 * c) = v4[it];
    }
    qb[0] *= 0.17677669529663687f; // pre-scaled query (torch scales q, not the scores)
    qb[1] *= 0.17677669529663687f;
    __syncthreads();
    f32x4 sc[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // A operand: key row kt*16 + col, d = 16g + 4q + j
            const f32x4 ka = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + col) * kAttnRow + 16 * g + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j], qb[g][j], sc[kt], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { // D row 4q + r = key index within the tile
            if (kt * 16 + 4 * q + r >= T) sc[kt][r] = -INFINITY;
            mx = fmaxf(mx, sc[kt][r]);
        }
    }
    mx = quad_max(mx);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[kt][r] = expf(sc[kt][r] - mx); // masked keys: exp(-inf) = 0
            l += sc[kt][r];
        }
    l = quad_sum(l);
    // O^T[d][query]: for step (kt, r) lane q supplies P^T[kt*16 + 4q + r][query] = sc[kt][r];
    // the A operand is V[kt*16 + 4q + r][16mt + col]
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vrow = Vs + (kt * 16 + 4 * q + r) * kAttnRow + col;
            o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[0], sc[kt][r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16], sc[kt][r], o[1], 0, 0, 0);
        }
    // D rows 16mt + 4q + r = d; queries >= T fall outside the descriptor
    const float inv = 1.0f / l;
    const __amdgpu_buffer_rsrc_t ors = make_rsrc(out + (int64_t)b * T * kTencD, T * kTencD * 4);
    const int ooff = (tq * kTencD + h * kTencHd + 4 * q) * 4;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[0] * inv), ors, ooff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[1] * inv), ors, ooff, 64, 0);
}

// Self-attention with 3 x f16 split operands (B2H_TENC_F16X3; the split itself: see the chain's
// f16x3 section below).  Same workgroup / wave mapping and softmax as b2h_attn_mfma_f32; both
// products run on v_mfma_f32_16x16x32_f16 as hi.hi + hi.lo + lo.hi with fp32 accumulation:
//   S^T tile = K[16 keys][32 dims] . Q^T           one k-step (k = dims): A from LDS rows of K
//              (f16 hi / lo, 64 B per key), B = this lane's query dims 8q .. 8q+7
//   O^T      = V^T[32 dims][keys] . P^T            k = keys, 32 per step: the lane's score
//              registers of key tiles 2s (j < 4) and 2s+1 (j >= 4) ARE the B operand when k-slot
//              (s, q, j) means key 32s + 16(j>>2) + 4q + (j&3); V^T is stored in LDS with its keys
//              in that slot order, so an A fragment is one 16-byte read.
constexpr int kAttnVtRow = 136; // halves per V^T row: 128 key slots + 8 (rows 4 banks apart)

template <int NT>
__global__ __launch_bounds__(64 * NT) void b2h_attn_mfma_h3(const float* __restrict__ qkv, float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem_attn3[];
    constexpr int KS = (NT + 1) / 2;                       // k-steps of 32 keys
    _Float16* Kh = reinterpret_cast<_Float16*>(smem_attn3); // [NT*16 keys][32 dims]
    _Float16* Kl = Kh + NT * 16 * kTencHd;
    _Float16* Vh = Kl + NT * 16 * kTencHd;                  // [32 dims][kAttnVtRow key slots]
    _Float16* Vl = Vh + kTencHd * kAttnVtRow;
    const int b = blockIdx.x / kTencHeads, h = blockIdx.x % kTencHeads;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + (int64_t)b * T * (3 * kTencD), T * 3 * kTencD * 4);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const int tq = wave * 16 + col;
    // K, V of the head: each thread takes dims 4c .. 4c+3 of the key PAIR (2u, 2u+1) -- adjacent
    // slots of V^T, so a dim's two keys go out as one 32-bit LDS write
    const int vc = threadIdx.x & 7, vu = threadIdx.x >> 3;
    f32x4 k4[2], v4[2], q4[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int off = ((2 * vu + it) * 3 * kTencD + h * kTencHd + 4 * vc) * 4;
        k4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, kTencD * 4, 0));
        v4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 2 * kTencD * 4, 0));
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) // this lane's query, dims 8q + 4g .. +3
        q4[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs, (tq * 3 * kTencD + h * kTencHd + 8 * q + 4 * g) * 4, 0, 0));
    if (KS * 32 > NT * 16) { // odd NT: the last k-step's upper 16 key slots have no writer
        for (int i = threadIdx.x; i < kTencHd * 16; i += 64 * NT) {
            const int d = i >> 4, p = (KS - 1) * 32 + 8 * ((i >> 2) & 3) + 4 + (i & 3);
            Vh[d * kAttnVtRow + p] = (_Float16)0.f;
            Vl[d * kAttnVtRow + p] = (_Float16)0.f;
        }
    }
    {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const int t0 = 2 * vu;
        // key t -> slot: k-step t>>5, then 8*((t>>2)&3) + 4*((t>>4)&1) + (t&3); t0 is even: t0+1 -> p+1
        const int p = (t0 & ~31) + 8 * ((t0 >> 2) & 3) + 4 * ((t0 >> 4) & 1) + (t0 & 3);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f16x4 kh, kl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 a = (_Float16)k4[it][e];
                kh[e] = a;
                kl[e] = (_Float16)(k4[it][e] - (float)a);
            }
            *reinterpret_cast<f16x4*>(Kh + (t0 + it) * kTencHd + 4 * vc) = kh;
            *reinterpret_cast<f16x4*>(Kl + (t0 + it) * kTencHd + 4 * vc) = kl;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 a0 = (_Float16)v4[0][e], a1 = (_Float16)v4[1][e];
            *reinterpret_cast<f16x2*>(Vh + (4 * vc + e) * kAttnVtRow + p) = f16x2{a0, a1};
            *reinterpret_cast<f16x2*>(Vl + (4 * vc + e) * kAttnVtRow + p) =
                f16x2{(_Float16)(v4[0][e] - (float)a0), (_Float16)(v4[1][e] - (float)a1)};
        }
    }
    f16x8 qh, ql; // B operand of S^T, pre-scaled (torch scales q, not the scores)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = q4[j >> 2][j & 3] * 0.17677669529663687f;
        const _Float16 a = (_Float16)x;
        qh[j] = a;
        ql[j] = (_Float16)(x - (float)a);
    }
    __syncthreads();
    f32x4 sc[2 * KS];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(Kh + (kt * 16 + col) * kTencHd + 8 * q);
        const f16x8 al = *reinterpret_cast<const f16x8*>(Kl + (kt * 16 + col) * kTencHd + 8 * q);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh, a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { // D row 4q + r = key index within the tile
            if (kt * 16 + 4 * q + r >= T) a[r] = -INFINITY;
            mx = fmaxf(mx, a[r]);
        }
        sc[kt] = a;
    }
    if (2 * KS > NT) sc[2 * KS - 1] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mx = quad_max(mx);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2 * KS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[kt][r] = __expf(sc[kt][r] - mx); // v_exp_f32 (1 ulp); masked keys: exp(-inf) = 0
            l += sc[kt][r];
        }
    l = quad_sum(l);
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = sc[2 * s + (j >> 2)][j & 3];
            const _Float16 a = (_Float16)x;
            ph[j] = a;
            pl[j] = (_Float16)(x - (float)a);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { // A: V^T row d = 16mt + col, key slots 32s + 8q .. +7
            const f16x8 vh = *reinterpret_cast<const f16x8*>(Vh + (16 * mt + col) * kAttnVtRow + 32 * s + 8 * q);
            const f16x8 vl = *reinterpret_cast<const f16x8*>(Vl + (16 * mt + col) * kAttnVtRow + 32 * s + 8 * q);
            o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[mt], 0, 0, 0);
            o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[mt], 0, 0, 0);
            o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[mt], 0, 0, 0);
        }
    }
    const float inv = 1.0f / l;
    const __amdgpu_buffer_rsrc_t ors = make_rsrc(out + (int64_t)b * T * kTencD, T * kTencD * 4);
    const int ooff = (tq * kTencD + h * kTencHd + 4 * q) * 4;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[0] * inv), ors, ooff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[1] * inv), ors, ooff, 64, 0);
}

// ---- per-frame chain ---------------------------------------------------------------------
// Between two attention calls every operation of the model is per-frame.  For
// v_mfma_f32_16x16x4_f32 the accumulator tile of one GEMM (lane = frame, register r of M-tile
// mt = feature 16mt + 4q + r) IS the B operand of the next GEMM (k-group g = mt, lane k-index
// q, step j = r), so a wave carries its 16 frames through a whole chain of Linear layers in
// registers:   attention output -> out_proj -> +residual -> LayerNorm1 -> linear1 -> ReLU ->
//              linear2 -> +residual -> LayerNorm2 -> next layer's Q, K, V (or the output head).
// Each stage's weights (<= 128 outputs, 64 KB) + bias/gamma/beta are one blob, double-buffered
// in LDS: stage s+1's blob is requested from L2 before stage s's MFMAs and written to the other
// buffer after them; one barrier per stage.
enum { ST_SET = 0, ST_RELU = 1, ST_RESLN_GLOBAL = 2, ST_RESLN_REG = 3, ST_STORE = 4 };
constexpr int kChainMaxStages = 8;
constexpr int kStageParams = 3 * kTencD;                                   // bias, gamma, beta
constexpr int kStageBlobMax = kLinChunkMT * 8 * 64 * 4 + kStageParams;    // floats: 16384 + 384

struct ChainStage {
    const float* blob;   // [wf4 x 16 B of weight fragments][bias 128][gamma 128][beta 128]
    float* out;          // ST_STORE: destination; other types: optional copy of the stage result (or nullptr)
    int type, mtiles, kgroups, ldo, nout;
    int wf4;             // fp32: mtiles*kgroups*64 (k-groups of 16); f16x3: 2*mtiles*kgroups*64 (k-groups of 32, hi + lo)
};
struct ChainArgs {
    const float* x;      // (N, ldx) rows entering the chain
    int ldx, kgroups0, kvalid;
    const float* pe;     // optional positional encoding added to x (frame n -> pe[n % T])
    int T;
    const float* res;    // residual rows (N, 128) for ST_RESLN_GLOBAL
    int64_t n;
    int nstages;
    // item transforms fused around the model (b2h_tenc_forward_fused; same flags as FusedArgs):
    //   front launch: kPreChest x -= x[:, chest] and kPreNorm x /= factor, BEFORE the positional
    //                 encoding is added (the reference transforms the item, then the model adds pe:
    //                 steps/utils.py:180-210, HandPoseModels.py:167);
    //   last launch : kPostDenorm y *= factor (traintest.py:270-271) and kPostMask
    //                 y[seq, n_frames[seq]:] = 0 (utils.py:309-312) in the 42-wide output store.
    int flags;
    float factor;
    const int64_t* n_frames; // (B) for kPostMask
    int Tseq;                // frames per sequence (the launch's own `T` is 1 behind the front)
    ChainStage st[kChainMaxStages];
};

template <int KG, int MT>
__device__ __forceinline__ void chain_gemm(const f32x4* __restrict__ wl, int lane, const f32x4 (&cur)[8], f32x4 (&acc)[8]) {
    // per k-group: the MT fragments, then MT independent accumulator chains advance together
    // (v_mfma_f32_16x16x4_f32 has 40 cycles of dependent latency against a 32-cycle issue)
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        f32x4 aw[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aw[mt] = wl[(mt * KG + g) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (B2H_ABLATE & 8192) acc[mt][j] += aw[mt][j] * cur[g][j];
                else acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt][j], cur[g][j], acc[mt], 0, 0, 0);
            }
    }
}

// ---- 3 x f16 split operands (B2H_TENC_F16X3) ---------------------------------------------------
// x = hi + lo with hi = f16(x), lo = f16(x - hi) carries 22 significant bits, and
//     W.x  ~=  Whi.xhi + Whi.xlo + Wlo.xhi          (the dropped Wlo.xlo term is ~2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 (fp32 accumulate) at 16x the fp32 matrix rate each: fp32-grade
// results (max error of the golden model vs fp64: 1.2e-6, the same as the fp32 kernel) for 3/16 of
// the matrix cycles, valid while |activation| < 65504 (f16 range).
// The accumulator -> operand identity of the fp32 chain carries over: k-slot (g, q, j) of the
// 32-wide k-group g holds feature 32g + 16(j>>2) + 4q + (j&3), i.e. lane (frame, q) packs its own
// accumulator tiles 2g (j < 4) and 2g+1 (j >= 4); the host packs the weights in the same slot order.
__device__ __forceinline__ void chain_split(const f32x4 (&v)[8], f16x8 (&bh)[4], f16x8 (&bl)[4]) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; j += 2) { // two values per packed convert; residual = one v_fma_mix_f32 each
            const float x0 = v[2 * g + (j >> 2)][j & 3], x1 = v[2 * g + (j >> 2)][(j & 3) + 1];
            const f16x2 h = f16x2{(_Float16)x0, (_Float16)x1};
            const uint32_t hb = __builtin_bit_cast(uint32_t, h);
            float r0, r1; // x - hi, exact in fp32 (asm: hipcc otherwise converts hi back and subtracts)
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x1));
            const f16x2 l = f16x2{(_Float16)r0, (_Float16)r1};
            bh[g][j] = h[0]; bh[g][j + 1] = h[1];
            bl[g][j] = l[0]; bl[g][j + 1] = l[1];
        }
}

template <int KG, int MT>
__device__ __forceinline__ void chain_gemm_h3(const f32x4* __restrict__ wl, int lane, const f16x8 (&bh)[4],
                                              const f16x8 (&bl)[4], f32x4 (&acc)[8]) {
    const f16x8* whi = reinterpret_cast<const f16x8*>(wl);
    const f16x8* wlo = whi + MT * KG * 64;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = whi[(mt * KG + g) * 64 + lane];
            al[mt] = wlo[(mt * KG + g) * 64 + lane];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[g], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[g], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[g], acc[mt], 0, 0, 0);
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int KG, int MT, int UM = (MT + 1) / 2, int D = 1>
__device__ __forceinline__ void chain_gemm_h3_db(const f32x4* __restrict__ wl, int lane, const f16x8 (&bh)[4],
                                                 const f16x8 (&bl)[4], f32x4 (&acc)[8]) {
    // Round 3: the fragments travel in UNITS of UM M-tiles of one k-group (UM hi + UM lo fragments), through a
    // ring of D + 1 unit buffers: unit u + D is requested from LDS before unit u's MFMAs are issued, so its
    // ~200 cycles of LDS latency lie under MFMAs instead of in front of them.  Before, a whole k-group (16
    // fragments) was requested and then waited for, four times per GEMM: stamps showed 3.3-3.5 k cycles per
    // 96 MFMAs even with half the workgroup's waves idle (tools/chain_stamps.py), against ~1.6 k of issue time.
    // Per accumulator the products are added in the same order as before (k-group by k-group: lo.hi, hi.lo,
    // hi.hi).
    constexpr int NU = (MT + UM - 1) / UM; // units per k-group
    constexpr int U = NU * KG;
    const f16x8* whi = reinterpret_cast<const f16x8*>(wl) + lane;
    const f16x8* wlo = whi + MT * KG * 64;
    f16x8 ah[D + 1][UM], al[D + 1][UM];
    auto load = [&](auto uc) {
        constexpr int u = decltype(uc)::value, g = u / NU, m0 = (u % NU) * UM, n = (MT - m0 < UM) ? MT - m0 : UM;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            ah[u % (D + 1)][i] = whi[((m0 + i) * KG + g) * 64];
            al[u % (D + 1)][i] = wlo[((m0 + i) * KG + g) * 64];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * n, 0); // pin the order: these reads ...
    };
    auto mma = [&](auto uc) {
        constexpr int u = decltype(uc)::value, g = u / NU, m0 = (u % NU) * UM, n = (MT - m0 < UM) ? MT - m0 : UM;
        const f16x8 (&h)[UM] = ah[u % (D + 1)];
        const f16x8 (&l)[UM] = al[u % (D + 1)];
#pragma unroll
        for (int i = 0; i < n; ++i) acc[m0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l[i], bh[g], acc[m0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < n; ++i) acc[m0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[i], bl[g], acc[m0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < n; ++i) acc[m0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[i], bh[g], acc[m0 + i], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * n, 0); // ... then these MFMAs
    };
    auto pre = [&](auto uc) { if constexpr (decltype(uc)::value < U) load(uc); };
    static_for<D>(pre);
    auto step = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u + D < U) load(std::integral_constant<int, u + D>{});
        mma(uc);
    };
    static_for<U>(step);
}

// Memory operations are buffer instructions over per-workgroup descriptors: the hardware range
// check stands in for every lane predicate (rows past the batch, feature groups a stage does not
// have, a missing output), so the loop has no branch around a load or store and the compiler can
// count outstanding operations instead of draining them (with predicated global_load/store it
// waited vmcnt(0) before the first MFMA of every stage and before every store).
constexpr uint32_t kOob = 0x7ffffff0u; // byte offset no descriptor here reaches

__device__ __forceinline__ f32x4 chain_ld(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}


// ---- Q, K, V projection + self-attention in one kernel (B2H_TENC_F16X3) ---------------------------
// Round 3.  In the two-kernel form above the chain wrote Q, K, V (1 536 B per frame and layer) for the
// attention kernel to read back: 60 % of the path's HBM traffic, on kernels that sit on their HBM floor.
// Here the projection lives where its result is used: a workgroup is bound to one HEAD for the whole
// launch -- its 96 x 128 slice of in_proj_weight (rows of Q_h, K_h, V_h; f16 hi + lo fragments, 48 KB)
// is copied into LDS once -- and walks over sequences: wave w owns frames 16w .. 16w+15 of the sequence,
//     x rows (residual stream, 512 B per frame)  --split-->  [Q_h | K_h | V_h] = W_h . x + b_h     (72 MFMAs)
//     Q_h stays in registers: with the chain's k-slot order (slot (q, j) = dim 16(j>>2) + 4q + (j&3)) a lane's
//         eight accumulators ARE its query fragment; K_h goes to LDS as the lane's one 16-byte chunk of its
//         key row in the same slot order, V_h as V^T[dim][key slot] like b2h_attn_mfma_h3 stores it
//     barrier, then scores / softmax / P.V exactly as in b2h_attn_mfma_h3.
// K and V are double-buffered (sequence i + 1 is projected while slower waves still attend to sequence i:
// one barrier per sequence), the next sequence's x rows are requested before this one's MFMAs.  Only the
// residual stream (read) and the attention output (written) cross HBM: 1 KB per frame and layer instead of
// 2 KB here + 1.5 KB of Q, K, V stores in the chain.  The four head-workgroups of a sequence list sit on
// one XCD (blockIdx = 8 * (4 * slot + head) + xcd), so three of their four reads of a row are L2 hits.
// hi = f16(v) packed two per instruction, residual v - hi as one mixed-precision FMA per value, lo = f16(residual):
// 16 vector instructions for 8 values (the scalar casts cost twice that; see split8 in kernel_mfma3.h)
__device__ __forceinline__ void tenc_split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h = f16x2{(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(v[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(v[2 * i + 1]));
        const f16x2 l = f16x2{(_Float16)r0, (_Float16)r1};
        hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
        lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
    }
}

struct AttnQkvArgs {
    const float* x;        // (B*T, 128) residual stream entering the layer
    float* out;            // (B*T, 128) attention output, head h -> columns 32h ..
    const float* blob[kTencHeads]; // per head: [hi: mt(6)][g(4)][lane] f16x8, [lo] the same, then 384 fp32 (bias in the first 96)
    int T;
    int64_t B;
};
constexpr int kQkvMT = 6;                                       // M-tiles of a head's projection: Q 0-1, K 2-3, V 4-5
constexpr int kQkvBlobBytes = 2 * kQkvMT * 4 * 64 * 16 + kStageParams * 4; // 49 152 + 1 536

template <int NT>
__global__ __launch_bounds__(64 * NT) void b2h_attn_qkv_h3(AttnQkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_aq[];
    constexpr int KS = (NT + 1) / 2;                          // k-steps of 32 keys
    constexpr int kKRow = 48;                                 // halves per K row: 32 dims + 16 of padding -- with 64-B rows
                                                              // four keys of a ds_read_b128 lane group share a bank quad
                                                              // (30 % of this kernel's LDS cycles were conflicts); 96 B: none
    constexpr int kKBytes = NT * 16 * kKRow * 2;              // one K image (hi or lo)
    constexpr int kVBytes = kTencHd * kAttnVtRow * 2;         // one V^T image
    constexpr int kKV = 2 * kKBytes + 2 * kVBytes;            // K hi, K lo, V^T hi, V^T lo
    const f32x4* wl = reinterpret_cast<const f32x4*>(smem_aq);
    const float* prm = reinterpret_cast<const float*>(smem_aq + 2 * kQkvMT * 4 * 64 * 16);
    char* kvbase = smem_aq + kQkvBlobBytes;
    // which head, which sequences: blockIdx = 8 * (4 * slot + head) + xcd
    const int xcd = blockIdx.x & 7, inx = blockIdx.x >> 3, h = inx & 3, slot = inx >> 2;
    const int nslots = (int)((gridDim.x >> 3) >> 2);          // sequence slots per XCD
    const int64_t stride = 8 * (int64_t)nslots;
    int64_t b = (int64_t)slot * 8 + xcd;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const int tq = wave * 16 + col;
    // the head's weights: once per workgroup
    for (int i = threadIdx.x; i < kQkvBlobBytes / 16; i += 64 * NT)
        reinterpret_cast<uint4*>(smem_aq)[i] = reinterpret_cast<const uint4*>(a.blob[h])[i];
    if (b >= a.B) return; // (after the copy loop only for symmetry; no barrier has been entered yet)
    auto load_x = [&](f32x4 (&xr)[8], int64_t bb) { // this lane's frame, features 16g + 4q .. +3; rows >= T read 0
        const bool on = bb < a.B;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (on ? bb : 0) * a.T * kTencD, on ? a.T * kTencD * 4 : 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) xr[g] = chain_ld(rs, (uint32_t)(tq * kTencD + 16 * g + 4 * q) * 4u);
    };
    f32x4 xr[8];
    load_x(xr, b);
    if (KS * 32 > NT * 16) { // odd NT: the last k-step's upper 16 key slots of V^T have no writer (both buffers)
        for (int i = threadIdx.x; i < 2 * kTencHd * 16; i += 64 * NT) {
            const int bufi = i / (kTencHd * 16), r = i % (kTencHd * 16);
            const int d = r >> 4, p = (KS - 1) * 32 + 8 * ((r >> 2) & 3) + 4 + (r & 3);
            _Float16* Vh = reinterpret_cast<_Float16*>(kvbase + bufi * kKV + 2 * kKBytes);
            Vh[d * kAttnVtRow + p] = (_Float16)0.f;
            (Vh + kTencHd * kAttnVtRow)[d * kAttnVtRow + p] = (_Float16)0.f;
        }
    }
    __syncthreads(); // weights (and the V^T padding) in LDS
    // key t -> V^T slot: k-step t>>5, then 8*((t>>2)&3) + 4*((t>>4)&1) + (t&3)
    const int vslot = (tq & ~31) + 8 * ((tq >> 2) & 3) + 4 * ((tq >> 4) & 1) + (tq & 3);
    int buf = 0;
#pragma unroll 1
    for (; b < a.B; b += stride, buf ^= 1) {
        _Float16* Kh = reinterpret_cast<_Float16*>(kvbase + buf * kKV);
        _Float16* Kl = Kh + NT * 16 * kKRow;
        _Float16* Vh = Kl + NT * 16 * kKRow;
        _Float16* Vl = Vh + kTencHd * kAttnVtRow;
        f16x8 bh[4], bl[4];
        chain_split(xr, bh, bl);
        load_x(xr, b + stride); // the next sequence's rows travel under this one's work
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < kQkvMT; ++mt) acc[mt] = *reinterpret_cast<const f32x4*>(prm + 16 * mt + 4 * q);
        chain_gemm_h3<4, kQkvMT>(wl, lane, bh, bl, acc);
        // acc[mt][r] = output 16 mt + 4q + r of this lane's frame: Q_h = outputs 0..31, K_h = 32..63, V_h = 64..95
        f16x8 qh, ql;
        {
            float vq[8], vk[8], vv[8]; // slot (q, j) = dim 16 (j >> 2) + 4q + (j & 3) = accumulator (j >> 2, j & 3)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                vq[j] = acc[j >> 2][j & 3] * 0.17677669529663687f; // torch scales q, not the scores
                vk[j] = acc[2 + (j >> 2)][j & 3];
                vv[j] = acc[4 + (j >> 2)][j & 3];
            }
            f16x8 kh, kl, vh, vl;
            tenc_split8(vq, qh, ql);
            tenc_split8(vk, kh, kl);
            tenc_split8(vv, vh, vl);
            *reinterpret_cast<f16x8*>(Kh + tq * kKRow + 8 * q) = kh;
            *reinterpret_cast<f16x8*>(Kl + tq * kKRow + 8 * q) = kl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { // V[key tq][dim 16 (j >> 2) + 4q + (j & 3)] -> V^T[dim][slot of the key]
                const int d = 16 * (j >> 2) + 4 * q + (j & 3);
                Vh[d * kAttnVtRow + vslot] = vh[j];
                Vl[d * kAttnVtRow + vslot] = vl[j];
            }
        }
        __syncthreads(); // K, V of this sequence complete (the other buffer is free again two sequences on)
        f32x4 sc[2 * KS];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(Kh + (kt * 16 + col) * kKRow + 8 * q);
            const f16x8 al = *reinterpret_cast<const f16x8*>(Kl + (kt * 16 + col) * kKRow + 8 * q);
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh, s4, 0, 0, 0);
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql, s4, 0, 0, 0);
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh, s4, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { // D row 4q + r = key index within the tile; only the last tile can cross T
                if (kt == NT - 1 && kt * 16 + 4 * q + r >= a.T) s4[r] = -INFINITY;
                mx = fmaxf(mx, s4[r]);
            }
            sc[kt] = s4;
        }
        if (2 * KS > NT) sc[2 * KS - 1] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        mx = quad_max(mx);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2 * KS; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] = __expf(sc[kt][r] - mx); // v_exp_f32 (1 ulp); masked keys: exp(-inf) = 0
                l += sc[kt][r];
            }
        l = quad_sum(l);
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f16x8 ph, pl;
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = sc[2 * s + (j >> 2)][j & 3];
            tenc_split8(pv, ph, pl);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { // A: V^T row d = 16mt + col, key slots 32s + 8q .. +7
                const f16x8 vh = *reinterpret_cast<const f16x8*>(Vh + (16 * mt + col) * kAttnVtRow + 32 * s + 8 * q);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(Vl + (16 * mt + col) * kAttnVtRow + 32 * s + 8 * q);
                o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[mt], 0, 0, 0);
                o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[mt], 0, 0, 0);
                o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[mt], 0, 0, 0);
            }
        }
        const float inv = 1.0f / l;
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(a.out + b * a.T * kTencD, a.T * kTencD * 4);
        const int ooff = (tq * kTencD + h * kTencHd + 4 * q) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[0] * inv), ors, ooff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[1] * inv), ors, ooff, 64, 0);
    }
}

template <bool H3> // false: fp32 operands (v_mfma_f32_16x16x4_f32); true: 3 x f16 split
__global__ __launch_bounds__(64 * kLinWaves, 2) void b2h_tenc_chain(ChainArgs a) {
    int nstamp = 0;
    (void)nstamp;
    extern __shared__ __attribute__((aligned(16))) char smem_chain[];
    f32x4* buf0 = reinterpret_cast<f32x4*>(smem_chain);
    f32x4* buf1 = buf0 + kStageBlobMax / 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, tcol = lane & 15, q0 = lane >> 4;
    constexpr int kFrames = 16 * kLinWaves, kThreads = 64 * kLinWaves;
    // Persistent since round 3: a workgroup walks over 128-frame blocks (blockIdx.x, + gridDim.x, ...).  With the
    // Q, K, V stages gone a block is three stages, and the per-workgroup prologue (first blob from L2, row loads
    // from HBM, the dispatch itself) had grown to a third of a workgroup's life (tools/chain_stamps.py: 15 k of
    // 46 k cycles); now the next block's first blob travels under this block's last stage.
    const int nblocks = (int)((a.n + kFrames - 1) / kFrames);         // (< 2^31: the host refuses more)
    const int fr0 = wave * 16 + tcol;                                 // this lane's frame in its block
    constexpr int kPerThread = (kStageBlobMax / 4 + kThreads - 1) / kThreads; // float4 per thread: 9
    static_assert(kPerThread == 9 && kThreads * 16 == 8192, "blob staging below assumes 9 x 8 KiB");

    auto blob_bytes = [&](int s) { return (a.st[s].wf4 + kStageParams / 4) * 16; };
    auto fetch_blob = [&](f32x4 (&w)[kPerThread], const float* blob, int bytes) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(blob, (B2H_ABLATE & 1024) ? 0 : bytes);
#pragma unroll
        for (int u = 0; u < kPerThread; ++u)
            w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16, u * 8192, 0));
    };
    auto put_blob = [&](f32x4* dst, const f32x4 (&w)[kPerThread]) {
#pragma unroll
        for (int u = 0; u < kPerThread - 1; ++u) dst[threadIdx.x + u * kThreads] = w[u];
        if (threadIdx.x + (kPerThread - 1) * kThreads < kStageBlobMax / 4)
            dst[threadIdx.x + (kPerThread - 1) * kThreads] = w[kPerThread - 1];
    };

    B2H_STAMP(); // 0: entry
    f32x4 wreg[kPerThread];
    fetch_blob(wreg, a.st[0].blob, blob_bytes(0));
    int gs = 0; // stages done by this workgroup: the parity of the LDS buffer the current stage reads
#pragma unroll 1
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int64_t n0 = (int64_t)blk * kFrames;                        // first frame of the block
    const int rows = (int)(a.n - n0 < kFrames ? a.n - n0 : kFrames);  // frames in it
    const bool next_block = blk + (int)gridDim.x < nblocks;
    int fr = fr0, q = q0; // opaque per block: hoisted out of the block loop, the 24 row offsets derived from `fr`
    asm volatile("" : "+v"(fr), "+v"(q)); // and the eight `have` masks (16 SGPRs) and feature offsets derived from
    // `q` stay live through every stage, and the kernel spills vector and scalar registers
    // rows entering the chain (features 16g + 4q .. +3 per k-group), the positional encoding
    // added to them (front launch), and the residual rows of a leading ST_RESLN_GLOBAL stage
    f32x4 cur[8], resid[8];
    {
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x + n0 * a.ldx, rows * a.ldx * 4);
        const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.pe, a.pe ? a.T * a.kvalid * 4 : 0);
        const __amdgpu_buffer_rsrc_t rrs = make_rsrc(a.res ? a.res + n0 * kTencD : nullptr, a.res ? rows * kTencD * 4 : 0);
        const uint32_t pos = (uint32_t)((n0 + fr) % (a.T > 0 ? a.T : 1));
        const int pre = __builtin_amdgcn_readfirstlane(a.flags) & (kPreChest | kPreNorm);
        // one bound for "k-group exists and feature is valid" (4q < 16, so g < kgroups0 <=> 16g + 4q < 16 kgroups0);
        // as two tests per k-group the compiler kept eight lane masks in 16 SGPRs across the block loop and spilled
        const int klim = 16 * a.kgroups0 < a.kvalid ? 16 * a.kgroups0 : a.kvalid;
        f32x4 chest = {0.f, 0.f, 0.f, 0.f};
        if (pre & kPreChest) { // joint 1 = channels 2, 3 of the frame's own row: (x, y, x, y) per float4
            const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)((uint32_t)(fr * a.ldx + 2) * 4u), 0, 0);
            const float cx_ = __uint_as_float(c[0]), cy_ = __uint_as_float(c[1]);
            chest = f32x4{cx_, cy_, cx_, cy_};
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int k0 = 16 * g + 4 * q;
            const bool have = k0 < klim; // g < kgroups0 && k0 < kvalid
            cur[g] = chain_ld(xrs, have ? (uint32_t)(fr * a.ldx + k0) * 4u : kOob);
            if (pre) { // wave-uniform; lanes without data hold zeros and stay zero
                if (have) cur[g] -= chest;
                if (pre & kPreNorm) cur[g] = cur[g] / a.factor; // true division, like the reference
            }
            cur[g] += chain_ld(prs, have ? (pos * (uint32_t)a.kvalid + k0) * 4u : kOob);
            resid[g] = chain_ld(rrs, (uint32_t)(fr * kTencD + k0) * 4u);
        }
    }
    // output-side transforms of the 42-wide head (last launch only)
    const int post = __builtin_amdgcn_readfirstlane(a.flags) & (kPostDenorm | kPostMask);
    const float omul = (post & kPostDenorm) ? a.factor : 1.0f;
    bool odead = false;
    if ((post & kPostMask) && a.n_frames && fr < rows) {
        const int64_t nn = n0 + fr;
        odead = (nn % a.Tseq) >= a.n_frames[nn / a.Tseq];
    }
    f16x8 bh[4], bl[4]; // H3: the GEMM operand, split from `cur`
    if constexpr (H3) chain_split(cur, bh, bl);
    if (gs == 0) { // the workgroup's first block: its first blob is still in registers
        put_blob(buf0, wreg);
        B2H_STAMP(); // 1: prologue done
        __syncthreads();
        B2H_STAMP(); // 2: past the first barrier
    }

#pragma unroll 1
    for (int s = 0; s < a.nstages; ++s, ++gs) {
        const ChainStage st = a.st[s];
        f32x4* wl = (gs & 1) ? buf1 : buf0;
        f32x4* wn = (gs & 1) ? buf0 : buf1;
        const float* prm = reinterpret_cast<const float*>(wl + st.wf4);
        // the next stage's blob -- after the last stage: the next block's first -- is requested now and lands
        // under this stage's MFMAs
        const bool more = s + 1 < a.nstages;
        fetch_blob(wreg, a.st[more ? s + 1 : 0].blob, more ? blob_bytes(s + 1) : (next_block ? blob_bytes(0) : 0));
        // accumulators start from the bias plus the residual (ST_RESLN_GLOBAL, only valid as
        // stage 0, finds the rows read on entry in `resid`)
        const bool addres = st.type == ST_RESLN_GLOBAL || st.type == ST_RESLN_REG;
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            acc[mt] = *reinterpret_cast<const f32x4*>(prm + 16 * (mt < st.mtiles ? mt : 0) + 4 * q);
            if (addres) acc[mt] += resid[mt];
        }
        B2H_STAMP(); // 2 + 6s: fetch issued, accumulators initialised
        // the GEMM, branch-free for the three shapes the model has
        if constexpr (H3) {
            if (st.kgroups == 4 && st.mtiles == 8) chain_gemm_h3_db<4, 8>(wl, lane, bh, bl, acc);
            else if (st.kgroups == 1) chain_gemm_h3_db<1, 8>(wl, lane, bh, bl, acc);
            else chain_gemm_h3_db<4, 3>(wl, lane, bh, bl, acc);
        } else {
            if (st.kgroups == 8 && st.mtiles == 8) chain_gemm<8, 8>(wl, lane, cur, acc);
            else if (st.kgroups == 2) chain_gemm<2, 8>(wl, lane, cur, acc);
            else chain_gemm<8, 3>(wl, lane, cur, acc);
        }
        B2H_STAMP(); // 3 + 6s: GEMM issued
        // next blob into the other buffer (free since the previous barrier), before this stage's
        // stores are issued: the wait for it then covers nothing younger
        put_blob(wn, wreg);
        B2H_STAMP(); // 4 + 6s: blob in LDS
        // epilogue
        if (addres && !(B2H_ABLATE & 256)) {
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) sum += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
            sum = quad_sum(sum);
            const float mean = sum * (1.0f / kTencD);
            float var = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[m][r] - mean;
                    var += d * d;
                }
            var = quad_sum(var);
            const float rstd = 1.0f / sqrtf(var * (1.0f / kTencD) + 1e-5f);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(prm + kTencD + 16 * m + 4 * q);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(prm + 2 * kTencD + 16 * m + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[m][r] = (acc[m][r] - mean) * rstd * g4[r] + b4[r];
                resid[m] = cur[m];
            }
        } else if (st.type == ST_RELU) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[m][r] = fmaxf(acc[m][r], 0.f);
        } else if (st.type == ST_SET || (B2H_ABLATE & 256)) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { cur[m] = acc[m]; resid[m] = acc[m]; }
        } else if constexpr (H3) {
            // ST_STORE: the operand that carries over is (bh, bl); defining `cur` on this path too makes
            // it dead across the back edge, so its 32 registers are free during the GEMM (the kernel
            // spilled 14 without this)
#pragma unroll
            for (int m = 0; m < 8; ++m) cur[m] = acc[m];
        }
        if constexpr (H3) {
            if (st.type != ST_STORE) chain_split(cur, bh, bl); // ST_STORE leaves the operand as it is
        }
        B2H_STAMP(); // 5 + 6s: epilogue (+ split) done
        // ST_STORE writes the accumulators, other types an optional copy of the stage's result;
        // no output = an empty descriptor (skipping the block instead measured no faster).  Rows are nout floats wide: whole float4 where they fit
        // and the two-float tail of the 42-wide head (nout is 128 or 42).
        {
            const bool on = st.out != nullptr && !(B2H_ABLATE & 4096);
            const __amdgpu_buffer_rsrc_t ors = make_rsrc(on ? st.out + n0 * st.ldo : nullptr, on ? rows * st.ldo * 4 : 0);
            const bool raw = st.type == ST_STORE;
            if (post && raw && st.nout == kOutCh) { // wave-uniform: the output head of a fused forward only
                // (a real branch: the empty asm keeps hipcc from if-converting it into ~80 selects and
                // multiplies that every stage of every launch would then execute)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int m = 0; m < 3; ++m) { // the 42 outputs live in M-tiles 0..2
                    acc[m] = acc[m] * omul;                 // x factor, or x 1.0f (exact)
                    if (odead) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f32x4 v = raw ? acc[m] : cur[m];
                const int o0 = 16 * m + 4 * q;
                const uint32_t off = (uint32_t)(fr * st.ldo + o0) * 4u;
                const bool whole = m < st.mtiles && o0 + 3 < st.nout;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)(whole ? off : kOob), 0, 0);
                if (m == 2) { // the only M-tile a 42-wide row ends in
                    const bool tail = !whole && m < st.mtiles && o0 + 1 < st.nout;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, ors,
                                                          (int)(tail ? off : kOob), 0, 0);
                    // keep the next vector's VALU writes off this store's data registers for two wait
                    // states (store-data write-after-read, see kernel_mfma16.h; hipcc pads only > 64-bit data)
                    asm volatile("s_nop 1" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        B2H_STAMP(); // 6 + 6s: stores issued
        if (!(B2H_ABLATE & 2048)) __syncthreads();
        B2H_STAMP(); // 7 + 6s: past the barrier
    }
    } // blocks
}

} // namespace b2h