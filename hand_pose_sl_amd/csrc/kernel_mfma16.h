// Persistent bf16/f16 matrix-core kernel: the throughput path.
//
// Path: ConvModel.forward, HandPoseModels.py:40-64 (four Conv1d k=5 + ReLU).
//
// Launch shape: one 512-thread workgroup per CU (8 waves = 2 per SIMD), each
// WAVE an independent pipeline over whole sequences (chunks of <= 208 frames):
//
//   once per workgroup : all four layers' weight fragments + biases -> LDS (46.6 KB)
//   per chunk, per wave:
//     commit   : the chunk's (T,24) fp32 rows, already waiting in registers, are
//                cast and written to this wave's LDS image [time][32 ch] (64-B rows)
//     prefetch : the NEXT chunk's rows are requested from HBM into the same
//                registers (up to 21 x 16 B per lane) and fly during the math
//     layers   : per 16-frame tile 5 ds_read_b128 (one per tap) feed 10 (15 for the
//                head) v_mfma_f32_16x16x32; D = W[chan][(tap,ch)] x Act[(tap,ch)][time]
//                starts from the bias fragment; ReLU (integer max), zero-padding
//                mask (last tile only) and the 16-bit cast stay in registers; one
//                ds_write_b128 per lane puts the tile back, 2 rows lower (in-place
//                image, see kernel_mfma.h); the head stores fp32 straight to y.
//   No workgroup barrier after the weight copy; waves never exchange data.
//
// HBM traffic per frame = 96 B read + 168 B written (the algorithmic minimum);
// weights are read once per workgroup.
#pragma once
#include "b2h_common.h"
#include "kernel_mfma.h"

namespace b2h {

constexpr int kRows16 = 224;                       // LDS rows (64 B) per wave
constexpr int kWaves16 = 8;                        // waves per persistent workgroup
constexpr int kWaveLds16 = kRows16 * 64;           // 14336 B
constexpr int kWFrag16 = 64 * 16;                  // one (mt,tap) fragment: 64 lanes x 16 B
constexpr int kWLayerOff16[4] = {0, 10 * kWFrag16, 20 * kWFrag16, 30 * kWFrag16};
constexpr int kWBytes16 = 45 * kWFrag16;           // 46080
constexpr int kBiasOff16[4] = {kWBytes16, kWBytes16 + 128, kWBytes16 + 256, kWBytes16 + 384};
constexpr int kPacked16 = kWBytes16 + 9 * 64;      // + bias [L][mt][q][4] fp32 = 46656
constexpr int kLds16 = kPacked16 + kWaves16 * kWaveLds16; // 161344 <= 163840
constexpr int kChunkWhole16 = 208;                 // a sequence up to this long is one chunk
constexpr int kChunkSplit16 = 192;                 // longer sequences: chunks of 192 (+-8 halo)
constexpr int kInRegs = 20;                        // ceil(208 * 6 / 64) float4 per lane (<= 208 input frames)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int PREC> struct Pack2;
template <> struct Pack2<PREC_BF16> { using v2 = bf16x2; };
template <> struct Pack2<PREC_F16> { using v2 = f16x2; };

template <int PREC> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    using v2 = typename Pack2<PREC>::v2;
    v2 p = __builtin_convertvector(f32x2{a, b}, v2);
    return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ float relu_bits(float v) { // max(v,0) as one v_max_i32
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
}

struct Geom16 {
    int64_t seq;
    int s, e;        // output frames [s, e)
    int in_lo, nf4;  // first input frame, number of float4 to load
};

__device__ __forceinline__ Geom16 geom16(int64_t chunk, int cps, int TT, int T) {
    Geom16 g;
    g.seq = chunk / cps;
    const int c = (int)(chunk - g.seq * cps);
    g.s = c * TT;
    g.e = min(g.s + TT, T);
    g.in_lo = max(g.s - kHalo, 0);
    g.nf4 = (min(g.e + kHalo, T) - g.in_lo) * (kInCh / 4);
    return g;
}

struct InRegs { float4 v[kInRegs]; };

__device__ __forceinline__ void issue_loads16(InRegs& R, const float* __restrict__ x, const Geom16& g,
                                              int T, int lane) {
    const float4* src = reinterpret_cast<const float4*>(x + (g.seq * (int64_t)T + g.in_lo) * kInCh);
#pragma unroll
    for (int j = 0; j < kInRegs; ++j) {
        const int i = lane + 64 * j;
        R.v[j] = (i < g.nf4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// registers -> LDS image of the layer-1 input (P(t,0) = t - s + 8).
// Three load iterations (192 float4) cover exactly 32 rows, so a lane needs only
// three (row, column) pairs; every other address is one of those plus a multiple
// of 32 rows = 2048 B, which also leaves the swizzle term unchanged.
template <int PREC, bool FUSED>
__device__ __forceinline__ void commit16(const InRegs& R, char* lds, const float* __restrict__ x,
                                         const Geom16& g, int T, int lane, int pos_emb,
                                         const FusedArgs& fa) {
    const int P0 = g.in_lo + 8 - g.s; // physical row of the first loaded frame: 0, or 8 at s == 0
    char* base = lds + P0 * 64;      // (P0 is a multiple of 8: swizzle term unchanged)
    int rr[3], off[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int u = lane + 64 * jj;
        rr[jj] = u / 6;
        const int c4 = u - rr[jj] * 6;
        off[jj] = lds_off<64>(rr[jj], c4 >> 1) + (c4 & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < kInRegs; ++j) {
        constexpr int kGroupBytes = 32 * 64;
        const int G = j / 3, jj = j % 3;
        const int i = lane + 64 * j;
        if (i < g.nf4) {
            float4 v = R.v[j];
            if constexpr (FUSED) {
                const int t = g.in_lo + 32 * G + rr[jj];
                if (fa.flags & kPreChest) { // body -= body[:,1] (steps/utils.py:203-210)
                    const float2 ch = *reinterpret_cast<const float2*>(x + (g.seq * (int64_t)T + t) * kInCh + 2);
                    v.x -= ch.x; v.y -= ch.y; v.z -= ch.x; v.w -= ch.y;
                }
                if (fa.flags & kPreNorm) { // body / factor (steps/utils.py:180-190)
                    v.x = v.x / fa.factor; v.y = v.y / fa.factor;
                    v.z = v.z / fa.factor; v.w = v.w / fa.factor;
                }
            }
            uint2 o = {pack2<PREC>(v.x, v.y), pack2<PREC>(v.z, v.w)};
            *reinterpret_cast<uint2*>(base + G * kGroupBytes + off[jj]) = o;
        }
    }
    // channels 24..31: zero (pos_emb: slot 24 = t/100, HandPoseModels.py:71-75)
    const int nrows = g.nf4 / 6;
    const int padoff = lds_off<64>(lane, 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = lane + 64 * k;
        if (r < nrows) {
            uint4 z = {0u, 0u, 0u, 0u};
            if (pos_emb) z.x = pack2<PREC>((float)(g.in_lo + r) / 100.0f, 0.f);
            *reinterpret_cast<uint4*>(base + k * 64 * 64 + padoff) = z;
        }
    }
    const uint4 z4 = {0u, 0u, 0u, 0u};
    if (g.s == 0 && lane < 32) // t in [-8,0): zero padding of every layer
        *reinterpret_cast<uint4*>(lds + lds_off<64>(lane >> 2, lane & 3)) = z4;
    if (g.in_lo + nrows == T && lane < 8) // t = T, T+1
        *reinterpret_cast<uint4*>(base + lds_off<64>(nrows + (lane >> 2), lane & 3)) = z4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int PREC, int L, bool FUSED>
__device__ __forceinline__ void layer16p(char* lds, const char* wlds, const Geom16& g, int T,
                                         int lane, float* __restrict__ yseq, const FusedArgs& fa,
                                         int64_t nvalid) {
    using P = Prec<PREC>;
    using vec8 = typename P::vec8;
    constexpr int MT = (L == 3) ? 3 : 2;
    constexpr int h = 6 - 2 * L;
    const int tcol = lane & 15, q = lane >> 4;
    const int lo = max(g.s - h, 0), hi = min(g.e + h, T);
    const int ntiles = (hi - lo + 15) >> 4;

    vec8 A[MT][kTaps];
    f32x4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < kTaps; ++s)
            A[mt][s] = *reinterpret_cast<const vec8*>(wlds + kWLayerOff16[L] + (mt * kTaps + s) * kWFrag16 + lane * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        bias[mt] = *reinterpret_cast<const f32x4*>(wlds + kBiasOff16[L] + (mt * 4 + q) * 16);

    const int pin = 8 - 2 * L - g.s; // P(t, L)   = t + pin
    const int pout = pin - 2;        // P(t, L+1) = t + pout
    // fragment addresses of tile 0; a tile step is 16 rows = 1024 B and leaves the
    // swizzle term ((P>>1)&3) unchanged
    int rd[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) rd[s] = lds_off<64>(lo + tcol + s - kPad + pin, q);
    int wr = lds_off<64>(lo + tcol + pout, q);

#pragma unroll 1
    for (int m = 0; m < ntiles; ++m) {
        vec8 Bf[kTaps];
#pragma unroll
        for (int s = 0; s < kTaps; ++s) Bf[s] = *reinterpret_cast<const vec8*>(lds + rd[s] + m * 1024);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = P::mfma(A[mt][s], Bf[s], acc[mt]);

        const int tau = lo + 16 * m;
        if constexpr (L < 3) {
            float v[8];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mt * 4 + r] = relu_bits(acc[mt][r]);
            if (tau + 16 > T) { // only the last tile can hold frames >= T (zero padding of the next layer)
                const bool inside = tau + tcol < T;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = inside ? v[k] : 0.f;
            }
            uint4 o = {pack2<PREC>(v[0], v[1]), pack2<PREC>(v[2], v[3]), pack2<PREC>(v[4], v[5]),
                       pack2<PREC>(v[6], v[7])};
            *reinterpret_cast<uint4*>(lds + wr + m * 1024) = o;
        } else {
            const int t = tau + tcol;
            if (t < g.e) {
                float* yr = yseq + (int64_t)t * kOutCh + 4 * q;
                bool dead = false;
                if constexpr (FUSED) dead = (int64_t)t >= nvalid;
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    f32x4 v = acc[mt];
                    if constexpr (FUSED) {
                        if (fa.flags & kPostDenorm) v = v * fa.factor; // traintest.py:387-388
                        if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};      // utils.py:309-312
                    }
                    if (mt < 2 || q < 2) {
                        *reinterpret_cast<float2*>(yr + 16 * mt) = float2{v[0], v[1]};
                        *reinterpret_cast<float2*>(yr + 16 * mt + 2) = float2{v[2], v[3]};
                    } else if (q == 2) {
                        *reinterpret_cast<float2*>(yr + 16 * mt) = float2{v[0], v[1]};
                    }
                }
            }
        }
    }
    if constexpr (L < 3) {
        if (hi == T) { // sequence end: next layer reads frames T, T+1 as zeros
            const int t = T + (lane >> 2);
            if (lane < 8 && t >= lo + 16 * ntiles)
                *reinterpret_cast<uint4*>(lds + lds_off<64>(t + pout, lane & 3)) = uint4{0u, 0u, 0u, 0u};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int PREC, bool FUSED>
__global__ __launch_bounds__(64 * kWaves16, 2) void b2h_fwd_mfma16(
    const float* __restrict__ x, float* __restrict__ y, int T, int cps, int TT, int64_t nchunks,
    const void* __restrict__ wpacked, int pos_emb, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    // weights + biases of all four layers: one copy per workgroup
    for (int i = threadIdx.x; i < kPacked16 / 16; i += 64 * kWaves16)
        reinterpret_cast<uint4*>(smem16)[i] = reinterpret_cast<const uint4*>(wpacked)[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    char* lds = smem16 + kPacked16 + wave * kWaveLds16;
    const int64_t stride = (int64_t)gridDim.x * kWaves16;
    int64_t chunk = blockIdx.x + (int64_t)gridDim.x * wave; // consecutive chunks -> different CUs
    if (chunk >= nchunks) return;

    InRegs R;
    Geom16 g = geom16(chunk, cps, TT, T);
    issue_loads16(R, x, g, T, lane);
    while (true) {
        commit16<PREC, FUSED>(R, lds, x, g, T, lane, pos_emb, fa);
        const int64_t next = chunk + stride;
        const bool more = next < nchunks;
        Geom16 gn = g;
        if (more) {
            gn = geom16(next, cps, TT, T);
            issue_loads16(R, x, gn, T, lane); // flies under the four layers below
        }
        float* yseq = y + g.seq * (int64_t)T * kOutCh;
        int64_t nvalid = T;
        if constexpr (FUSED)
            if ((fa.flags & kPostMask) && fa.n_frames) nvalid = fa.n_frames[g.seq];
        layer16p<PREC, 0, FUSED>(lds, smem16, g, T, lane, yseq, fa, nvalid);
        layer16p<PREC, 1, FUSED>(lds, smem16, g, T, lane, yseq, fa, nvalid);
        layer16p<PREC, 2, FUSED>(lds, smem16, g, T, lane, yseq, fa, nvalid);
        layer16p<PREC, 3, FUSED>(lds, smem16, g, T, lane, yseq, fa, nvalid);
        if (!more) break;
        chunk = next;
        g = gn;
    }
}

} // namespace b2h
