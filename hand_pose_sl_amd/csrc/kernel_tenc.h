// TransformerEnc (body2hand/src/models/HandPoseModels.py:118-178) on gfx950, exact fp32.
//
// First, correctness-first generation of this path: every Linear of the
// model is one launch of an fp32 matrix-core kernel over ALL frames of the batch (a Linear is
// per-frame, so M = B*T rows), with bias / ReLU / residual + LayerNorm fused into its
// epilogue; self-attention is one launch per layer with a workgroup per (sequence, head).
// Activations travel through a caller-provided HBM workspace between launches.
//
//   linear : D[feat][frame] += W[feat][k] * X[k][frame]   (v_mfma_f32_16x16x4_f32, exact fp32)
//            A = weights, staged per 128 output features in LDS in fragment order;
//            B = 16 frames per wave, read straight from X (16 B per lane);
//            epilogue on the accumulator tile (lane = frame, registers = features):
//            +bias, ReLU, or +residual then LayerNorm over the 128 features of a frame
//            (32 in-lane values + two cross-lane steps), 16-B stores.
//   attn   : workgroup = (sequence, head), wave = 16 query frames; scores and P.V on the matrix
//            cores, softmax in registers over all T keys (the reference passes no mask,
//            HandPoseModels.py:170); see b2h_attn_mfma_f32.
#pragma once
#include "b2h_common.h"
#include "kernel_mfma.h" // f32x4

namespace b2h {

constexpr int kTencD = 128;      // nhid (d_model and feed-forward width in the reference's CLIs)
constexpr int kTencHeads = 4;
constexpr int kTencHd = 32;      // head dim
constexpr int kLinWaves = 8;     // waves per workgroup of the linear kernel (16 frames each)
constexpr int kLinChunkMT = 8;   // M-tiles (x16 features) of weights staged in LDS at a time

enum { LIN_PLAIN = 0, LIN_RELU = 1, LIN_RES_LN = 2 };

struct LinearArgs {
    const float* x;      // (N, ldx) input rows
    int ldx;
    int kgroups;         // K / 16 rounded up (2 for K = 24, 8 for K = 128)
    int kvalid;          // real K (24 or 128): columns >= kvalid are read as 0
    const float* wfrag;  // [mt][g][lane][4]: W[16mt + (lane&15)][16g + 4(lane>>4) + j]
    const float* bias;   // (mtiles*16) zero padded
    int mtiles;          // ceil(O / 16)
    int nout;            // O
    float* y;            // (N, ldy)
    int ldy;
    int64_t n;           // rows (frames)
    const float* res;    // LIN_RES_LN: residual (N, 128)
    const float* gamma;  // LIN_RES_LN: LayerNorm weight / bias (128)
    const float* beta;
    const float* pe;     // optional positional encoding (max_len, kvalid) added to x; frame n -> pe[n % T]
    int T;
};

template <int EPI>
__global__ __launch_bounds__(64 * kLinWaves) void b2h_linear_f32(LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_lin[];
    f32x4* wl = reinterpret_cast<f32x4*>(smem_lin);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, tcol = lane & 15, q = lane >> 4;
    const int64_t n = ((int64_t)blockIdx.x * kLinWaves + wave) * 16 + tcol;
    const bool valid = n < a.n;

    // B fragments: this lane's frame, features 16g + 4q .. +3 of every k-group
    f32x4 bx[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        bx[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int k0 = 16 * g + 4 * q;
        if (g < a.kgroups && valid && k0 < a.kvalid) {
            bx[g] = *reinterpret_cast<const f32x4*>(a.x + n * a.ldx + k0);
            if (a.pe) bx[g] += *reinterpret_cast<const f32x4*>(a.pe + (n % a.T) * a.kvalid + k0);
        }
    }

    f32x4 keep[EPI == LIN_RES_LN ? kLinChunkMT : 1]; // LIN_RES_LN: the frame's 128 outputs stay in registers
    for (int c0 = 0; c0 < a.mtiles; c0 += kLinChunkMT) {
        const int cm = min(kLinChunkMT, a.mtiles - c0);
        __syncthreads(); // previous chunk's fragments no longer needed
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wfrag) + (size_t)c0 * a.kgroups * 64;
        for (int i = threadIdx.x; i < cm * a.kgroups * 64; i += 64 * kLinWaves) wl[i] = src[i];
        __syncthreads();
#pragma unroll 1
        for (int m = 0; m < cm; ++m) {
            const int mt = c0 + m;
            f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + 16 * mt + 4 * q);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g >= a.kgroups) break;
                const f32x4 aw = wl[(m * a.kgroups + g) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[j], bx[g][j], acc, 0, 0, 0);
            }
            if constexpr (EPI == LIN_RES_LN) {
                keep[m] = acc; // mtiles == 8: single chunk
            } else {
                if constexpr (EPI == LIN_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
                }
                if (valid) {
                    const int o0 = 16 * mt + 4 * q;
                    float* yr = a.y + n * a.ldy + o0;
                    if (o0 + 3 < a.nout && (a.ldy & 3) == 0) {
                        *reinterpret_cast<f32x4*>(yr) = acc;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (o0 + r < a.nout) yr[r] = acc[r];
                    }
                }
            }
        }
    }
    if constexpr (EPI == LIN_RES_LN) {
        // h = LayerNorm(residual + linear) over the frame's 128 features
        // (torch.nn.LayerNorm: biased variance, eps 1e-5)
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < kLinChunkMT; ++m) {
            if (valid) keep[m] += *reinterpret_cast<const f32x4*>(a.res + n * kTencD + 16 * m + 4 * q);
            s += keep[m][0] + keep[m][1] + keep[m][2] + keep[m][3];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / kTencD);
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < kLinChunkMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = keep[m][r] - mean;
                v += d * d;
            }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const float rstd = 1.0f / sqrtf(v * (1.0f / kTencD) + 1e-5f);
        if (valid) {
#pragma unroll
            for (int m = 0; m < kLinChunkMT; ++m) {
                const int o0 = 16 * m + 4 * q;
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.gamma + o0);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.beta + o0);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (keep[m][r] - mean) * rstd * g4[r] + b4[r];
                *reinterpret_cast<f32x4*>(a.y + n * a.ldy + o0) = o;
            }
        }
    }
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
// K and V of the head live in LDS ([key][32] fp32, rows T..16*ntiles zero).
constexpr int kAttnMaxTiles = 8; // T <= 128

__global__ __launch_bounds__(64 * kAttnMaxTiles) void b2h_attn_mfma_f32(const float* __restrict__ qkv,
                                                                      float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem_attn2[];
    float* Ks = reinterpret_cast<float*>(smem_attn2);
    const int nt = (T + 15) >> 4;
    float* Vs = Ks + nt * 16 * kTencHd;
    const int b = blockIdx.x / kTencHeads, h = blockIdx.x % kTencHeads;
    const float* base = qkv + (int64_t)b * T * (3 * kTencD) + h * kTencHd;
    // stage K, V: (nt*16) rows x 8 float4, zero beyond T
    for (int i = threadIdx.x; i < nt * 16 * 8; i += blockDim.x) {
        const int t = i >> 3, c = i & 7;
        f32x4 k4 = f32x4{0.f, 0.f, 0.f, 0.f}, v4 = k4;
        if (t < T) {
            k4 = *reinterpret_cast<const f32x4*>(base + (int64_t)t * (3 * kTencD) + kTencD + 4 * c);
            v4 = *reinterpret_cast<const f32x4*>(base + (int64_t)t * (3 * kTencD) + 2 * kTencD + 4 * c);
        }
        reinterpret_cast<f32x4*>(Ks)[i] = k4;
        reinterpret_cast<f32x4*>(Vs)[i] = v4;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nt) return;
    const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const int tq = wave * 16 + col;
    // B operand of S^T: this lane's query, d = 16g + 4q + j, pre-scaled (torch scales q, not the scores)
    f32x4 qb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        qb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tq < T) qb[g] = *reinterpret_cast<const f32x4*>(base + (int64_t)tq * (3 * kTencD) + 16 * g + 4 * q) * 0.17677669529663687f;
    }
    f32x4 sc[kAttnMaxTiles];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < kAttnMaxTiles; ++kt) {
        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < nt) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // A operand: key row kt*16 + col, d = 16g + 4q + j
                const f32x4 ka = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + col) * kTencHd + 16 * g + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j], qb[g][j], sc[kt], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { // D row 4q + r = key index within the tile
                if (kt * 16 + 4 * q + r >= T) sc[kt][r] = -INFINITY;
                mx = fmaxf(mx, sc[kt][r]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < kAttnMaxTiles; ++kt)
        if (kt < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] = expf(sc[kt][r] - mx); // masked keys: exp(-inf) = 0
                l += sc[kt][r];
            }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    // O^T[d][query]: for step (kt, r) lane q supplies P^T[kt*16 + 4q + r][query] = sc[kt][r];
    // the A operand is V[kt*16 + 4q + r][16mt + col]
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < kAttnMaxTiles; ++kt)
        if (kt < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = Vs + (kt * 16 + 4 * q + r) * kTencHd + col;
                o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[0], sc[kt][r], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16], sc[kt][r], o[1], 0, 0, 0);
            }
    if (tq < T) {
        const float inv = 1.0f / l;
        float* orow = out + ((int64_t)b * T + tq) * kTencD + h * kTencHd + 4 * q; // D rows 16mt + 4q + r = d
        *reinterpret_cast<f32x4*>(orow) = o[0] * inv;
        *reinterpret_cast<f32x4*>(orow + 16) = o[1] * inv;
    }
}

} // namespace b2h
