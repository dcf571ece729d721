// fp32 vector-ALU kernel: the generic-width path (any conv_channels <= 128; the only kernel
// above 64) and the device-side cross-check for the MFMA kernels.
//
// One 256-thread workgroup computes 64 output frames of one sequence through
// all four layers (HandPoseModels.py:55-58).  It loads the 80 input frames it
// depends on (+-8 halo) straight from the native (B,T,24) layout -- the
// reference's permute/view (:43-46) is a stride change only -- keeps both
// activation buffers and ONE TAP of the current layer's weights in LDS, and writes
// (B,T,42) == (B,T,21,2) contiguous.  Activations outside [0,T) are forced to
// zero after EVERY layer, which is what per-layer `padding=2` means.
#pragma once
#include "b2h_common.h"

namespace b2h {

constexpr int kValuTile = 64;
constexpr int kValuRows = kValuTile + 2 * kHalo; // 80

// WIDE: in-channel loop unrolled by 4 -- pays at 57..64 channels (7.9 -> 5.8 ms at C = 64), costs 10 % at 30.
// ITEMS: work items per thread; 38 row pairs x (conv_channels / 8) groups / 256 threads = 2 up to 104
// channels, 3 up to 128.
template <bool WIDE, int ITEMS = 2>
__global__ __launch_bounds__(256) void b2h_fwd_f32_valu(const float* __restrict__ x,
                                                        float* __restrict__ y, int T,
                                                        int tiles_per_seq, ValuParams p,
                                                        FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float smem_valu[];
    const int AS = p.act_stride;
    float* act0 = smem_valu;
    float* act1 = smem_valu + kValuRows * AS;
    float* wbuf = smem_valu + 2 * kValuRows * AS;

    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / tiles_per_seq;
    const int tile = blockIdx.x % tiles_per_seq;
    const int t0 = tile * kValuTile;
    const int tbase = t0 - kHalo;
    const float* xb = x + b * (int64_t)T * kInCh;
    const int c0 = p.pos_emb ? 1 : 0; // keypoints start at channel 1 with pos_emb (:78-84)

    // ---- input rows [tbase, tbase+80), zero outside the sequence
    for (int i = tid; i < kValuRows * (kInCh / 4); i += 256) {
        const int r = i / (kInCh / 4), c4 = i % (kInCh / 4);
        const int t = tbase + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) {
            v = *reinterpret_cast<const float4*>(xb + (int64_t)t * kInCh + c4 * 4);
            if (fa.flags & kPreChest) { // body -= body[:,1]  (steps/utils.py:203-210)
                const float2 ch = *reinterpret_cast<const float2*>(xb + (int64_t)t * kInCh + 2);
                v.x -= ch.x; v.y -= ch.y; v.z -= ch.x; v.w -= ch.y;
            }
            if (fa.flags & kPreNorm) { // body / factor     (steps/utils.py:180-190)
                v.x = v.x / fa.factor; v.y = v.y / fa.factor;
                v.z = v.z / fa.factor; v.w = v.w / fa.factor;
            }
        }
        float* dst = act0 + r * AS + c0 + c4 * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    if (p.pos_emb) // channel 0 = t/100 (HandPoseModels.py:71-75); zero outside the sequence
        for (int r = tid; r < kValuRows; r += 256) {
            const int t = tbase + r;
            act0[r * AS] = (t >= 0 && t < T) ? (float)t / 100.0f : 0.f;
        }

    float* in = act0;
    float* out = act1;
    int64_t nvalid = T;
    if ((fa.flags & kPostMask) && fa.n_frames) nvalid = fa.n_frames[b];

    // work item = TWO consecutive rows x one 8-channel group (nrows is even in every layer): each
    // pair of float4 weight reads feeds 16 FMAs.  <= 38 row pairs x 8 (16) groups / 256 threads = 2 (3) items.
    constexpr int kMaxItems = ITEMS;
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const ValuLayer L = p.L[l];
        const int ng = L.opad / 8;
        const int rlo = 2 * (l + 1), nrows = kValuRows - 4 * (l + 1);
        const int nitems = (nrows / 2) * ng;
        // accumulators of this thread's items start from the bias
        float acc[kMaxItems][2][8];
#pragma unroll
        for (int it = 0; it < kMaxItems; ++it) {
            const int item = tid + 256 * it;
            const int g = (item < nitems) ? item % ng : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[it][0][j] = acc[it][1][j] = L.b[g * 8 + j];
        }
        // The layer's weights are staged ONE TAP at a time (cin x opad floats: 16 KB at 64 channels
        // instead of 82 KB for all five), so that several workgroups fit a CU at every width.  The
        // accumulation order (tap outer, in-channel inner) is the one the single-stage version had.
#pragma unroll 1
        for (int k = 0; k < kTaps; ++k) {
            __syncthreads(); // previous tap / layer done with wbuf, `in` complete
            const float* wsrc = L.w + (size_t)k * L.cin * L.opad;
            for (int i = tid * 4; i < L.cin * L.opad; i += 256 * 4)
                *reinterpret_cast<float4*>(wbuf + i) = *reinterpret_cast<const float4*>(wsrc + i);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < kMaxItems; ++it) {
                const int item = tid + 256 * it;
                if (item >= nitems) continue;
                const int r = rlo + 2 * (item / ng), g = item % ng;
                const float* arow = in + (r + k - kPad) * AS;
                const float* wk = wbuf + g * 8;
#define B2H_VALU_CHANNEL(i)                                                                   \
    {                                                                                         \
        const float a0 = arow[i], a1 = arow[AS + (i)];                                        \
        const float4 w0 = *reinterpret_cast<const float4*>(wk + (i) * L.opad);                \
        const float4 w1 = *reinterpret_cast<const float4*>(wk + (i) * L.opad + 4);            \
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                       \
            acc[it][0][j] = fmaf(a0, w[j], acc[it][0][j]);                                    \
            acc[it][1][j] = fmaf(a1, w[j], acc[it][1][j]);                                    \
        }                                                                                     \
    }
                if constexpr (WIDE) {
#pragma unroll 4
                    for (int i = 0; i < L.cin; ++i) B2H_VALU_CHANNEL(i)
                } else {
                    for (int i = 0; i < L.cin; ++i) B2H_VALU_CHANNEL(i)
                }
#undef B2H_VALU_CHANNEL
            }
        }
#pragma unroll
        for (int it = 0; it < kMaxItems; ++it) {
            const int item = tid + 256 * it;
            if (item >= nitems) continue;
            const int g = item % ng;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = rlo + 2 * (item / ng) + h;
                const int t = tbase + r;
                const bool inside = (t >= 0 && t < T);
                if (l < 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        out[r * AS + g * 8 + j] = inside ? fmaxf(acc[it][h][j], 0.f) : 0.f;
                } else if (inside) {
                    float* yr = y + (b * (int64_t)T + t) * kOutCh + g * 8;
                    const bool dead = (int64_t)t >= nvalid;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (g * 8 + j < kOutCh) {
                            float v = acc[it][h][j];
                            if (fa.flags & kPostDenorm) v *= fa.factor; // traintest.py:387-388
                            yr[j] = dead ? 0.f : v;                     // utils.py:309-312
                        }
                }
            }
        }
        float* tmp = in; in = out; out = tmp;
    }
}

// hand_out = (hand - body[:,4]) / factor   (steps/utils.py:194-201,180-190)
__global__ __launch_bounds__(256) void b2h_target_transform_kernel(const float* __restrict__ body,
                                                                   const float* __restrict__ hand,
                                                                   float* __restrict__ out,
                                                                   int64_t frames, int flags,
                                                                   float factor) {
    // one thread per (frame, joint): float2
    const int64_t n = frames * 21;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / 21;
        float2 v = *reinterpret_cast<const float2*>(hand + i * 2);
        if (flags & 1) {
            const float2 w = *reinterpret_cast<const float2*>(body + f * kInCh + 4 * 2);
            v.x -= w.x; v.y -= w.y;
        }
        if (flags & 2) { v.x = v.x / factor; v.y = v.y / factor; }
        *reinterpret_cast<float2*>(out + i * 2) = v;
    }
}

// maskedPoseL1 (steps/utils.py:413-428): per sequence the mean of |pred - target| over its
// first n_frames[i] frames x 21 joints x 2, then the mean over the batch.
// WEIGHTED = poderatedPoseL1 (steps/utils.py:431-452): |pred * s - target * s| with one score per
// (frame, joint), both products rounded to fp32 before the subtraction as torch does; the batch
// reduction is then a SUM (b2h_mean_kernel's `divide` = 0).
// Pass 1: one workgroup per sequence -> per_seq[i] (fixed summation order: reproducible).
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void b2h_masked_l1_seq_kernel(const float* __restrict__ pred,
                                                                const float* __restrict__ target,
                                                                const float* __restrict__ scores,
                                                                const int64_t* __restrict__ n_frames,
                                                                float* __restrict__ per_seq, int T) {
    __shared__ float part[4];
    const int64_t b = blockIdx.x;
    int64_t n = n_frames ? n_frames[b] : T;
    n = n < 0 ? 0 : (n > T ? T : n);
    const int64_t cnt = n * kOutCh;                       // floats of this sequence that count
    const float2* p = reinterpret_cast<const float2*>(pred + b * (int64_t)T * kOutCh);
    const float2* t = reinterpret_cast<const float2*>(target + b * (int64_t)T * kOutCh);
    const float* sc = WEIGHTED ? scores + b * (int64_t)T * (kOutCh / 2) : nullptr;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < cnt / 2; i += 256) { // one (frame, joint) per step
        const float2 a = p[i], c = t[i];
        if constexpr (WEIGHTED) {
            const float w = sc[i];
            acc += fabsf(__fmul_rn(a.x, w) - __fmul_rn(c.x, w)) + fabsf(__fmul_rn(a.y, w) - __fmul_rn(c.y, w));
        } else {
            acc += fabsf(a.x - c.x) + fabsf(a.y - c.y);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) per_seq[b] = (part[0] + part[1] + part[2] + part[3]) / (float)cnt; // 0/0 = NaN like torch
}
// Pass 2: one workgroup, mean (divide != 0) or sum of per_seq over the batch.
__global__ __launch_bounds__(256) void b2h_mean_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t n,
                                                       int divide) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += v[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1] + part[2] + part[3]) / (divide ? (float)n : 1.0f);
}

} // namespace b2h
