// fp32-grade ConvModel on the f16 matrix cores: 3 x f16 split (B2H_KERNEL_F16X3_MFMA).
//
// Path: ConvModel.forward, HandPoseModels.py:40-64.  Same mapping as the exact-fp32 kernel
// (kernel_mfma.h: one wave owns one chunk of one sequence and carries it through all four layers,
// weights in registers, in-place LDS image, no workgroup barrier), different arithmetic:
//
//   every activation and weight is split   x = hi + lo,  hi = f16(x),  lo = f16(x - hi)
//   (22 significant bits) and every product is three v_mfma_f32_16x16x32_f16 with fp32 accumulation
//       W.x  ~=  Wlo.xhi + Whi.xlo + Whi.xhi          (the dropped Wlo.xlo term is ~2^-22 relative)
//   -- 3/16 of the matrix cycles of v_mfma_f32_16x16x4_f32 for fp32-grade results, valid while
//   |activation|, |weight| < 65504 (f16 range; raw pixel keypoints and their hidden activations
//   are far inside it).
//
// LDS per wave: two images (hi, lo) of the 16-bit kernels' swizzled 64-B rows ([time][32 ch] f16),
// together the 128 B per row of the fp32 image.  One 16x16x32 k-step = the 32 padded channels of one
// tap; with the hidden-layer channel map 8q + 4mt + r a lane's two accumulator tiles are the 8
// consecutive channels of its own 16-byte chunk, so the write-back is one ds_write_b128 per image.
#pragma once
#include "kernel_mfma.h"
#include "kernel_mfma16.h" // relu_bits
#include "dev/b2h_dev.h"  // B2H_STAMP3: empty in the shipped build

namespace b2h {

constexpr int kImg3 = kRows * 64; // bytes of one image (hi or lo) of a wave


// hi = f16(v) packed two per instruction (v_cvt_pk_f16_f32), residual v - hi as ONE mixed-precision
// FMA per value (v_fma_mix_f32 reads the f16 half straight out of the packed register; written as
// asm because hipcc otherwise converts hi back with v_cvt_f32_f16 and subtracts), lo = f16(residual)
// packed: 16 VALU for 8 values.
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
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

template <int L>
__device__ __forceinline__ void layer3(const ChunkCtx& cx, const MfmaParams& mp) {
    constexpr int MT = (L == 3) ? 3 : 2;
    constexpr int h = 6 - 2 * L;
    const int lo = max(cx.s - h, 0), hi = min(cx.e + h, cx.T);
    const int ntiles = (hi - lo + 15) >> 4;
    char* img_h = cx.lds;
    char* img_l = cx.lds + kImg3;

    f16x8 Ah[MT][kTaps], Al[MT][kTaps]; // [.][tap]: in-channels 8q + j of out-channel slot (lane & 15)
    f32x4 bias[MT];
    {
        const f16x8* wp = reinterpret_cast<const f16x8*>(mp.w[L]); // [mt][tap][hi|lo][lane]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s) {
                Ah[mt][s] = wp[((B2H_ABLATE & 64) ? 0 : ((mt * kTaps + s) * 2 + 0) * 64) + cx.lane];
                Al[mt][s] = wp[((B2H_ABLATE & 64) ? 64 : ((mt * kTaps + s) * 2 + 1) * 64) + cx.lane];
            }
        const f32x4* bp = reinterpret_cast<const f32x4*>(mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + cx.q];
    }
    B2H_STAMP3(cx, 2 + 2 * L); // weight fragments requested
    const int pin = 8 - 2 * L - cx.s;
    const int pout = pin - 2;
    // Tiles advance by 16 rows and the image's chunk swizzle has period 8, so the swizzled byte
    // offsets of this lane's five tap rows (and of its write-back row) are those of tile 0 plus
    // m * 1024: computed once per layer, one add per tile.
    int roff[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) roff[s] = lds_off<64>(lo + cx.tcol + s - kPad + pin, cx.q);
    int woff = lds_off<64>(lo + cx.tcol + pout, cx.q);
    HeadStore hs;
    if constexpr (L == 3) hs.init(cx, lo);

#pragma unroll 1
    for (int m = 0; m < ntiles; ++m) {
        const int tau = lo + 16 * m;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(img_h + roff[s]);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(img_l + roff[s]);
            roff[s] += 16 * 64;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mt][s], bh, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], bl, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], bh, acc[mt], 0, 0, 0);
        }
        const int t = tau + cx.tcol;
        if constexpr (L < 3) {
            float v[8]; // channels 8q + 4mt + r = slot j = 4mt + r of this lane's chunk
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
            if (tau + 16 > cx.T) { // only the tile that crosses the sequence end: frames >= T are padding
                const bool inside = t < cx.T;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
            }
            f16x8 oh, ol;
            split8(v, oh, ol);
            *reinterpret_cast<f16x8*>(img_h + woff) = oh;
            *reinterpret_cast<f16x8*>(img_l + woff) = ol;
            woff += 16 * 64;
        } else {
            hs.store(cx, acc, m);
        }
    }
    B2H_STAMP3(cx, 3 + 2 * L); // tiles done
    if constexpr (L < 3) {
        if (hi == cx.T) { // rows T, T+1 of the next layer's input: zero unless a tile covered them
            const int covered = lo + 16 * ntiles;
            const int t = cx.T + ((cx.lane >> 2) & 1);
            if (cx.lane < 16 && t >= covered)
                *reinterpret_cast<f32x4*>((cx.lane < 8 ? img_h : img_l) + lds_off<64>(t + pout, cx.lane & 3)) =
                    f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- input staging: (T,24) fp32 rows -> hi / lo images of layer-1 input ------------------------
__device__ __forceinline__ void stage_input3(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb) {
    char* img_h = cx.lds;
    char* img_l = cx.lds + kImg3;
    const int in_lo = max(cx.s - kHalo, 0), in_hi = min(cx.e + kHalo, cx.T);
    const int pin = 8 - cx.s; // P(t,0) = t + pin
    const int nf4 = (in_hi - in_lo) * (kInCh / 4);
    const float4* src = reinterpret_cast<const float4*>(xs + (int64_t)in_lo * kInCh);
    for (int i0 = cx.lane; i0 < nf4; i0 += 64 * 8) { // 8 loads in flight per lane
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u;
            v[u] = (i < nf4 && !(B2H_ABLATE & 128)) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u;
            if (i >= nf4) continue;
            const int rr = i / 6, c4 = i - rr * 6;
            const int t = in_lo + rr;
            float4 w = v[u];
            if (cx.fa.flags & kPreChest) { // body -= body[:,1]  (steps/utils.py:203-210)
                const float2 ch = *reinterpret_cast<const float2*>(xs + (int64_t)t * kInCh + 2);
                w.x -= ch.x; w.y -= ch.y; w.z -= ch.x; w.w -= ch.y;
            }
            if (cx.fa.flags & kPreNorm) { // body / factor     (steps/utils.py:180-190)
                w.x = w.x / cx.fa.factor; w.y = w.y / cx.fa.factor;
                w.z = w.z / cx.fa.factor; w.w = w.w / cx.fa.factor;
            }
            const float e[4] = {w.x, w.y, w.z, w.w};
            f16x4 wh, wl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 a = (_Float16)e[j];
                wh[j] = a;
                wl[j] = (_Float16)(e[j] - (float)a);
            }
            // channels 4c4 .. 4c4+3: half (c4 & 1) of 16-B chunk c4 >> 1
            const int off = lds_off<64>(t + pin, c4 >> 1) + (c4 & 1) * 8;
            *reinterpret_cast<f16x4*>(img_h + off) = wh;
            *reinterpret_cast<f16x4*>(img_l + off) = wl;
        }
    }
    // channel padding 24..31 = chunk 3 (pos_emb: channel 24 = t/100, HandPoseModels.py:71-75;
    // the layer-1 weights are packed with the position channel moved to slot 24)
    const int nrows = in_hi - in_lo;
    for (int r = cx.lane; r < nrows; r += 64) {
        const int t = in_lo + r, off = lds_off<64>(t + pin, 3);
        const float pe = pos_emb ? (float)t / 100.0f : 0.f;
        const _Float16 ph = (_Float16)pe;
        f16x8 zh, zl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { zh[j] = (_Float16)0.f; zl[j] = (_Float16)0.f; }
        zh[0] = ph;
        zl[0] = (_Float16)(pe - (float)ph);
        *reinterpret_cast<f16x8*>(img_h + off) = zh;
        *reinterpret_cast<f16x8*>(img_l + off) = zl;
    }
    // zero rows: t in [-8,0) at the sequence start (all layers' low padding) and t = T, T+1 at
    // the sequence end; 4 chunks per row per image
    if (cx.s == 0) {
        const int i = cx.lane; // 8 rows x 4 chunks x 2 images
        *reinterpret_cast<f32x4*>((i < 32 ? img_h : img_l) + lds_off<64>((i & 31) >> 2, i & 3)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (in_hi == cx.T && cx.lane < 16) {
        const int i = cx.lane; // 2 rows x 4 chunks x 2 images
        *reinterpret_cast<f32x4*>((i < 8 ? img_h : img_l) + lds_off<64>(cx.T + ((i & 7) >> 2) + pin, i & 3)) =
            f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per (sequence, chunk); no workgroup barrier anywhere.
// (two 4-wave workgroups per CU by LDS: 2 waves per SIMD, so each may use 256 VGPRs)
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void b2h_fwd_mfma_f16x3(
    const float* __restrict__ x, float* __restrict__ y, int T, int chunks_per_seq, int chunk_len,
    int64_t nchunks, MfmaParams mp, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem_mfma3[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (chunk >= nchunks) return;

    ChunkCtx cx;
    cx.lds = smem_mfma3 + (size_t)wave * 2 * kImg3;
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
    B2H_STAMP3(cx, 0);
    stage_input3(cx, x + cx.seq * (int64_t)T * kInCh, mp.pos_emb);
    B2H_STAMP3(cx, 1); // input staged
    layer3<0>(cx, mp); layer3<1>(cx, mp); layer3<2>(cx, mp); layer3<3>(cx, mp);
}

} // namespace b2h
