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

// One layer over this wave's chunk.  Software pipeline over 16-frame tiles, two deep: fragments are
// read two tiles ahead (ping-pong B0/B1), a tile's epilogue runs one tile late (ping-pong accA/accB)
// beside the following tile's MFMAs, and sched_group_barrier spreads that vector / LDS work between
// the MFMAs one at a time (tools/mfma_mix_bench.hip: per MFMA one VALU is free on this chip, a run of
// VALU after a run of MFMAs costs the sum).  Legal in the in-place image: tile m writes rows
// [tau-2, tau+14) of the next image, every fragment read issued before that write belongs to tiles
// <= m+2, and tiles > m read rows >= tau+14.  Only a layer's LAST tile can hold frames >= T and the
// loop never runs a last tile's epilogue, so its body has no padding mask and no branch.
// FUSED: the output-side item transforms (x factor, tail mask) exist only in that instantiation.
template <int L, bool FUSED>
__device__ __forceinline__ void layer3(const ChunkCtx& cx, const MfmaParams& mp) {
    constexpr int MT = (L == 3) ? 3 : 2;
    constexpr int h = 6 - 2 * L;
    const int lo = max(cx.s - h, 0), hi = min(cx.e + h, cx.T);
    const int ntiles = (hi - lo + 15) >> 4;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) const f16x8 lds_cf16x8;
    typedef __attribute__((address_space(3))) f16x8 lds_f16x8;

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
    // Fragment addresses as 32-bit LDS pointers into the hi image (the lo image is kImg3 bytes further;
    // a tile step is 16 rows = 1024 B and leaves the swizzle unchanged).  They advance once per loop
    // iteration and are made opaque there, so every access is register + immediate offset.
    lds_char* rp[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) rp[s] = (lds_char*)(cx.lds + lds_off<64>(lo + cx.tcol + s - kPad + pin, cx.q));
    lds_char* wp = (lds_char*)(cx.lds + lds_off<64>(lo + cx.tcol + pout, cx.q));
    int tq = lo + cx.tcol; // this lane's frame in the tile the loop stands at
    HeadStore<FUSED> hs;
    if constexpr (L == 3) hs.init(cx, lo);

    auto mma = [&](f32x4 (&acc)[MT], const f16x8 (&Bh)[kTaps], const f16x8 (&Bl)[kTaps]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mt][s], Bh[s], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bl[s], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bh[s], acc[mt], 0, 0, 0);
        }
    };
    auto epi = [&](const f32x4 (&acc)[MT], int k, bool mask) { // tile k past the one the loop stands at
        if constexpr (L < 3) {
            float v[8]; // channels 8q + 4mt + r = slot j = 4mt + r of this lane's chunk
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
            if (mask) { // frames >= T are the zero padding of the next layer
                const bool inside = tq + 16 * k < cx.T;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
            }
            f16x8 oh, ol;
            split8(v, oh, ol);
            *(lds_f16x8*)(wp + k * 1024) = oh;
            *(lds_f16x8*)(wp + k * 1024 + kImg3) = ol;
        } else {
            hs.store(cx, acc, k);
        }
    };
    auto fetch = [&](f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps], int k) { // unconditional, see kernel_mfma16.h
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            Bh[s] = *(lds_cf16x8*)(rp[s] + k * 1024);
            Bl[s] = *(lds_cf16x8*)(rp[s] + k * 1024 + kImg3);
        }
    };
    auto advance2 = [&]() {
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            rp[s] += 2048;
            asm volatile("" : "+v"(rp[s]));
        }
        wp += 2048;
        asm volatile("" : "+v"(wp));
        tq += 32;
        if constexpr (L == 3) hs.advance2();
    };
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 5 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); // VALU
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); // VMEM write (head) ...
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // ... or DS write
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
    };
    f16x8 B0h[kTaps], B0l[kTaps], B1h[kTaps], B1l[kTaps];
    f32x4 accA[MT], accB[MT];
    fetch(B0h, B0l, 0);
    fetch(B1h, B1l, 1);
    mma(accA, B0h, B0l); // tile 0
    fetch(B0h, B0l, 2);
    int m = 1; // the pointers stand at tile m - 1
#pragma unroll 1
    for (; m + 1 < ntiles; m += 2) {
        mma(accB, B1h, B1l); epi(accA, 0, false); fetch(B1h, B1l, 3); interleave();
        mma(accA, B0h, B0l); epi(accB, 1, false); fetch(B0h, B0l, 4); interleave();
        advance2();
    }
    if (m < ntiles) { mma(accB, B1h, B1l); epi(accA, 0, false); epi(accB, 1, true); }
    else epi(accA, 0, true);
    B2H_STAMP3(cx, 3 + 2 * L); // tiles done
    if constexpr (L < 3) {
        char* img_h = cx.lds;
        char* img_l = cx.lds + kImg3;
        if (hi == cx.T) { // rows T, T+1 of the next layer's input: zero unless a tile covered them
            const int covered = lo + 16 * ntiles;
            const int t = cx.T + ((cx.lane >> 2) & 1);
            if (cx.lane < 16 && t >= covered)
                *reinterpret_cast<f32x4*>((cx.lane < 8 ? img_h : img_l) + lds_off<64>(t + pout, cx.lane & 3)) =
                    f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wave_lds_sync();
    }
}

// ---- input staging: (T,24) fp32 rows -> hi / lo images of layer-1 input ------------------------
__device__ __forceinline__ void split4(const float4& w, f16x4& hi, f16x4& lo) {
    const float e[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 a = (_Float16)e[j];
        hi[j] = a;
        lo[j] = (_Float16)(e[j] - (float)a);
    }
}

__device__ __forceinline__ void stage_input3(const ChunkCtx& cx, const float* __restrict__ xs, int pos_emb) {
    char* img_h = cx.lds;
    char* img_l = cx.lds + kImg3;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const StagedRows st = stage_rows<64>(
        cx, xs, pos_emb, // channels 4c4 .. 4c4+3: half (c4 & 1) of 16-B chunk c4 >> 1
        [&](int P, int c4) { return lds_off<64>(P, c4 >> 1) + (c4 & 1) * 8; },
        [&](int off, float4 w) {
            f16x4 wh, wl;
            split4(w, wh, wl);
            *reinterpret_cast<f16x4*>(img_h + off) = wh;
            *reinterpret_cast<f16x4*>(img_l + off) = wl;
        },
        [&](int P, float pe) { // channel padding 24..31 = chunk 3
            const _Float16 ph = (_Float16)pe;
            f16x8 zh, zl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { zh[j] = (_Float16)0.f; zl[j] = (_Float16)0.f; }
            zh[0] = ph;
            zl[0] = (_Float16)(pe - (float)ph);
            *reinterpret_cast<f16x8*>(img_h + lds_off<64>(P, 3)) = zh;
            *reinterpret_cast<f16x8*>(img_l + lds_off<64>(P, 3)) = zl;
        });
    // zero rows: t in [-8,0) at the sequence start (all layers' low padding) and t = T, T+1 at its end;
    // 4 chunks per row per image
    if (cx.s == 0) { // 8 rows x 4 chunks x 2 images
        const int i = cx.lane;
        *reinterpret_cast<f32x4*>((i < 32 ? img_h : img_l) + lds_off<64>((i & 31) >> 2, i & 3)) = z4;
    }
    if (st.at_end && cx.lane < 16) { // 2 rows x 4 chunks x 2 images
        const int i = cx.lane;
        *reinterpret_cast<f32x4*>((i < 8 ? img_h : img_l) + lds_off<64>(st.P0 + st.nrows + ((i & 7) >> 2), i & 3)) = z4;
    }
    wave_lds_sync();
}

// One wave per (sequence, chunk); no workgroup barrier anywhere.
// (two 4-wave workgroups per CU by LDS: 2 waves per SIMD, so each may use 256 VGPRs)
template <bool FUSED>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void b2h_fwd_mfma_f16x3(
    const float* __restrict__ x, float* __restrict__ y, int T, int chunks_per_seq, int chunk_len,
    int64_t nchunks, MfmaParams mp, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem_mfma3[];
    ChunkCtx cx;
    if (!chunk_ctx(cx, smem_mfma3, 2 * kImg3, y, T, chunks_per_seq, chunk_len, nchunks, fa)) return;
    B2H_STAMP3(cx, 0);
    stage_input3(cx, x + cx.seq * (int64_t)T * kInCh, mp.pos_emb);
    B2H_STAMP3(cx, 1); // input staged
    layer3<0, FUSED>(cx, mp); layer3<1, FUSED>(cx, mp); layer3<2, FUSED>(cx, mp); layer3<3, FUSED>(cx, mp);
}

} // namespace b2h
