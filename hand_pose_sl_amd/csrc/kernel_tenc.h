// TransformerEnc (body2hand/src/models/HandPoseModels.py:118-178) on gfx950, exact fp32
// (SURVEY.md 8f N3).  Two kernels, both on v_mfma_f32_16x16x4_f32:
//
//   b2h_attn_mfma_f32   self-attention, workgroup = (sequence, head), wave = 16 query frames:
//                       scores and P.V on the matrix cores, softmax in registers over all T keys
//                       (the reference passes no mask, HandPoseModels.py:170).
//   b2h_tenc_chain_f32  everything between two attention calls, which is all per-frame: a wave
//                       carries 16 frames through a chain of Linear layers in registers (the
//                       accumulator layout IS the next GEMM's operand layout), with bias, ReLU,
//                       residual + LayerNorm fused between them; weights double-buffered in LDS.
//
// Per forward: 1 front chain (src + pe -> pose2hidden -> Q,K,V) and per layer 1 attention + 1
// chain launch; only the residual stream, the attention output and Q,K,V cross HBM (2.5 KB per
// frame of caller-provided workspace).
#pragma once
#include "b2h_common.h"
#include "kernel_mfma.h" // f32x4

namespace b2h {

constexpr int kTencD = 128;      // nhid (d_model and feed-forward width in the reference's CLIs)
constexpr int kTencHeads = 4;
constexpr int kTencHd = 32;      // head dim
constexpr int kLinWaves = 8;     // waves per workgroup of the linear kernel (16 frames each)
constexpr int kLinChunkMT = 8;   // M-tiles (x16 features) of weights staged in LDS at a time

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

template <int NT> // number of 16-frame tiles = ceil(T / 16): compile-time so that the loops are branch-free
__global__ __launch_bounds__(64 * NT) void b2h_attn_mfma_f32(const float* __restrict__ qkv,
                                                          float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem_attn2[];
    float* Ks = reinterpret_cast<float*>(smem_attn2);
    constexpr int nt = NT;
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
    f32x4 sc[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
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
    for (int kt = 0; kt < NT; ++kt)
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
    for (int kt = 0; kt < NT; ++kt)
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
    const float* blob;   // [mtiles*kgroups*64 float4 weight fragments][bias 128][gamma 128][beta 128]
    float* out;          // ST_STORE: destination; other types: optional copy of the stage result (or nullptr)
    int type, mtiles, kgroups, ldo, nout;
};
struct ChainArgs {
    const float* x;      // (N, ldx) rows entering the chain
    int ldx, kgroups0, kvalid;
    const float* pe;     // optional positional encoding added to x (frame n -> pe[n % T])
    int T;
    const float* res;    // residual rows (N, 128) for ST_RESLN_GLOBAL
    int64_t n;
    int nstages;
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
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt][j], cur[g][j], acc[mt], 0, 0, 0);
    }
}

__global__ __launch_bounds__(64 * kLinWaves, 2) void b2h_tenc_chain_f32(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_chain[];
    f32x4* buf0 = reinterpret_cast<f32x4*>(smem_chain);
    f32x4* buf1 = buf0 + kStageBlobMax / 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, tcol = lane & 15, q = lane >> 4;
    const int64_t n = ((int64_t)blockIdx.x * kLinWaves + wave) * 16 + tcol;
    const bool valid = n < a.n;
    constexpr int kPerThread = (kStageBlobMax / 4 + 64 * kLinWaves - 1) / (64 * kLinWaves); // float4 per thread: 9

    auto blob_f4 = [&](int s) { return a.st[s].mtiles * a.st[s].kgroups * 64 + kStageParams / 4; };
    // stage 0's blob straight into buffer 0
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.st[0].blob);
        const int cnt = blob_f4(0);
        for (int i = threadIdx.x; i < cnt; i += 64 * kLinWaves) buf0[i] = src[i];
    }
    // rows entering the chain: features 16g + 4q .. +3 per k-group
    f32x4 cur[8], resid[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        cur[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        resid[g] = cur[g];
        const int k0 = 16 * g + 4 * q;
        if (g < a.kgroups0 && valid && k0 < a.kvalid) {
            cur[g] = *reinterpret_cast<const f32x4*>(a.x + n * a.ldx + k0);
            if (a.pe) cur[g] += *reinterpret_cast<const f32x4*>(a.pe + (n % a.T) * a.kvalid + k0);
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int s = 0; s < a.nstages; ++s) {
        const ChainStage st = a.st[s];
        f32x4* wl = (s & 1) ? buf1 : buf0;
        f32x4* wn = (s & 1) ? buf0 : buf1;
        const float* prm = reinterpret_cast<const float*>(wl + st.mtiles * st.kgroups * 64);
        // accumulators start from the bias (LDS) plus the residual (HBM or registers)
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            acc[mt] = *reinterpret_cast<const f32x4*>(prm + 16 * (mt < st.mtiles ? mt : 0) + 4 * q);
            if (st.type == ST_RESLN_GLOBAL && valid)
                acc[mt] += *reinterpret_cast<const f32x4*>(a.res + n * kTencD + 16 * mt + 4 * q);
            if (st.type == ST_RESLN_REG) acc[mt] += resid[mt];
        }
        // request the next stage's blob AFTER the residual rows (so that waiting for those is a
        // counted wait that leaves these loads in flight); it lands under this stage's MFMAs
        f32x4 wreg[kPerThread];
        const bool more = s + 1 < a.nstages;
        const int ncnt = more ? blob_f4(s + 1) : 0;
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.st[more ? s + 1 : s].blob);
#pragma unroll
            for (int u = 0; u < kPerThread; ++u) {
                const int i = threadIdx.x + u * 64 * kLinWaves;
                wreg[u] = (i < ncnt) ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // the GEMM, branch-free for the three shapes the model has; eight independent accumulator
        // chains interleaved, fragment reads free to run ahead
        if (st.kgroups == 8 && st.mtiles == 8) chain_gemm<8, 8>(wl, lane, cur, acc);
        else if (st.kgroups == 2) chain_gemm<2, 8>(wl, lane, cur, acc);
        else chain_gemm<8, 3>(wl, lane, cur, acc);
        // epilogue
        if (st.type == ST_RESLN_GLOBAL || st.type == ST_RESLN_REG) {
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) sum += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / kTencD);
            float var = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[m][r] - mean;
                    var += d * d;
                }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
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
        } else if (st.type == ST_SET) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { cur[m] = acc[m]; resid[m] = acc[m]; }
        }
        if (st.out && valid) { // ST_STORE: the accumulators; otherwise a copy of the stage's result
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m < st.mtiles) {
                    const f32x4 v = (st.type == ST_STORE) ? acc[m] : cur[m];
                    const int o0 = 16 * m + 4 * q;
                    float* yr = st.out + n * st.ldo + o0;
                    if (o0 + 3 < st.nout && (st.ldo & 3) == 0) {
                        *reinterpret_cast<f32x4*>(yr) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (o0 + r < st.nout) yr[r] = v[r];
                    }
                }
            }
        }
        // next stage's blob into the other buffer; everyone is done with it since the previous barrier
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = threadIdx.x + u * 64 * kLinWaves;
            if (i < ncnt) wn[i] = wreg[u];
        }
        __syncthreads();
    }
}

} // namespace b2h
