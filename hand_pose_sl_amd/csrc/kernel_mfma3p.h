// Persistent fp32-grade matrix-core kernel: 3 x f16 split (B2H_KERNEL_F16X3_MFMA), the
// throughput form of kernel_mfma3.h.
//
// Path: ConvModel.forward, HandPoseModels.py:40-64.  Arithmetic as in kernel_mfma3.h (every
// activation and weight split x = hi + lo in f16, three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulate: 22 significant bits per operand); structure as in kernel_mfma16.h:
//
//   launch     : one 256-thread workgroup per CU, ONE wave per SIMD (so a wave may use the whole
//                512-entry register file), each wave an independent pipeline over chunks
//   once per WG: all four layers' hi + lo weight fragments + biases -> LDS (90.6 KB).  The
//                wave-per-chunk form re-read them from L2 for every chunk and layer: 90 KB per
//                112 frames, 6.5 TB/s of L2 traffic at 7 G frames/s, and every layer started
//                behind that latency (tools/ablate_conv3.sh: 7 % of the time, the synchronous
//                input staging another 11 %).
//   per chunk  : commit   the chunk's rows, waiting in registers as fp32, are split and written
//                         to this wave's hi / lo images [time][32 ch] (64-B rows, swizzled)
//                prefetch the NEXT chunk's rows are requested from HBM (12 x 16 B per lane) and
//                         fly under the four layers
//                layers   per 16-frame tile 10 ds_read_b128 (hi, lo per tap) feed 30 (45 for the
//                         head) MFMAs; fragments are read two tiles ahead and a tile's epilogue
//                         (ReLU, padding mask, split, two ds_write_b128 / the fp32 stores of the
//                         head) runs one tile late, beside the next tile's MFMAs
//   LDS        : 92 736 B weights + 4 x 2 x 8 704 B images = 162 368 B; the images hold 136 rows =
//                112 output frames + the +-8 halo, so T = 200 runs as two chunks (112 + 88:
//                55 tile-layers against 52 without a halo).
//   No workgroup barrier after the weight copy; waves never exchange data.
#pragma once
#include "kernel_mfma16.h" // Geom16, lane_rows16, Fused16, relu_bits, make_rsrc
#include "kernel_mfma3.h"  // split8
#include "dev/b2h_dev.h"   // B2H_ABLATE hooks: constant-false in the shipped build

namespace b2h {

constexpr int kWaves3p = 4;                        // waves per persistent workgroup (one per SIMD)
constexpr int kChunk3p = 112;                      // output frames per chunk (a shorter sequence is one chunk)
constexpr int kRows3p = kChunk3p + 24;             // 8 low halo + chunk + 8 high halo + 8 spare
constexpr int kImg3p = kRows3p * 64;               // bytes of one image (hi or lo) of a wave: 8704
constexpr int kWFrag3p = 64 * 16;                  // one fragment: 64 lanes x 16 B
constexpr int kWLayerOff3p[4] = {0, 20 * kWFrag3p, 40 * kWFrag3p, 60 * kWFrag3p}; // [mt][tap][hi|lo][lane]
constexpr int kWBytes3p = 90 * kWFrag3p;           // 92160
constexpr int kBiasOff3p[4] = {kWBytes3p, kWBytes3p + 128, kWBytes3p + 256, kWBytes3p + 384};
constexpr int kPacked3p = kWBytes3p + 9 * 64;      // + bias [L][mt][q][4] fp32 = 92736
constexpr int kLds3p = kPacked3p + kWaves3p * 2 * kImg3p; // 162368 <= 163840
constexpr int kInRegs3p = 12;                      // (112 + 16) frames x 6 float4 / 64 lanes

struct InRegs3p { float4 v[kInRegs3p]; };

// Request a chunk's input rows: 12 x 16 B per lane, lane-contiguous (1 KiB per instruction);
// bytes == 0 (nothing left to prefetch) issues no memory traffic.
__device__ __forceinline__ void issue_loads3p(InRegs3p& R, const float* base, int bytes, int lane) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);
#pragma unroll
    for (int j = 0; j < kInRegs3p; ++j) {
        const i32x4 r = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, 0));
        R.v[j] = __builtin_bit_cast(float4, r);
    }
}

// fp32 registers -> hi / lo images of the layer-1 input (P(t,0) = t - s + 8); the reference's item
// transforms are applied here when fused.  Every address is one of three per-lane offsets plus a
// multiple of 32 rows = 2048 B, which leaves the swizzle term unchanged.
template <bool FUSED>
__device__ __forceinline__ void commit3p(const InRegs3p& R, char* img_h, char* img_l, const Geom16& g, int T, int lane,
                                         int pos_emb, const float* __restrict__ xrow0, const Fused16& fu) {
    const int P0 = g.in_lo + 8 - g.s; // physical row of the first loaded frame: 0, or 8 at s == 0
    int rr[3], c4[3], off[3];
    lane_rows16(lane, rr, c4);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) off[jj] = P0 * 64 + lds_off<64>(rr[jj], c4[jj] >> 1) + (c4[jj] & 1) * 8;
    // No lane predicate: lanes past the chunk's last float4 hold the zeros the buffer load's range
    // check returned; their rows are the zero padding after the sequence end or rows no tile of this
    // chunk depends on (12 iterations cover 128 rows <= kRows3p - 8).
#pragma unroll
    for (int j = 0; j < kInRegs3p; ++j) {
        float4 v = R.v[j];
        if constexpr (FUSED) {
            if (fu.chest) { // wave-uniform branch
                float2 ch = make_float2(0.f, 0.f);
                if (lane + 64 * j < g.nf4) // row of this float4, channels 2..3 = joint 1 (chest)
                    ch = *reinterpret_cast<const float2*>(xrow0 + (32 * (j / 3) + rr[j % 3]) * kInCh + 2);
                v.x -= ch.x; v.y -= ch.y; v.z -= ch.x; v.w -= ch.y;
            }
            if (fu.norm) { // wave-uniform branch; true division like the reference
                v.x = v.x / fu.factor; v.y = v.y / fu.factor;
                v.z = v.z / fu.factor; v.w = v.w / fu.factor;
            }
        }
        const float e[4] = {v.x, v.y, v.z, v.w};
        f16x4 wh, wl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const _Float16 a = (_Float16)e[k];
            wh[k] = a;
            wl[k] = (_Float16)(e[k] - (float)a);
        }
        constexpr int kGroupBytes = 32 * 64;
        *reinterpret_cast<f16x4*>(img_h + (j / 3) * kGroupBytes + off[j % 3]) = wh;
        *reinterpret_cast<f16x4*>(img_l + (j / 3) * kGroupBytes + off[j % 3]) = wl;
    }
    // channels 24..31 = chunk 3: zero (pos_emb: slot 24 = t/100, HandPoseModels.py:71-75)
    const int nrows = g.nf4 / 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = lane + 64 * k;
        if (r < nrows) {
            const float pe = pos_emb ? (float)(g.in_lo + r) / 100.0f : 0.f;
            const _Float16 ph = (_Float16)pe;
            f16x8 zh, zl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { zh[j] = (_Float16)0.f; zl[j] = (_Float16)0.f; }
            zh[0] = ph;
            zl[0] = (_Float16)(pe - (float)ph);
            const int o = lds_off<64>(P0 + r, 3);
            *reinterpret_cast<f16x8*>(img_h + o) = zh;
            *reinterpret_cast<f16x8*>(img_l + o) = zl;
        }
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (g.s == 0) // t in [-8,0): zero padding of every layer; 8 rows x 4 chunks x 2 images
        *reinterpret_cast<f32x4*>((lane < 32 ? img_h : img_l) + lds_off<64>((lane & 31) >> 2, lane & 3)) = z4;
    if (g.in_lo + nrows == T && lane < 16) // t = T, T+1; 2 rows x 4 chunks x 2 images
        *reinterpret_cast<f32x4*>((lane < 8 ? img_h : img_l) + lds_off<64>(P0 + nrows + ((lane & 7) >> 2), lane & 3)) = z4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Pin the wait for the prefetched rows (and nothing younger) to the point of the call: see
// pin_loads16 in kernel_mfma16.h -- left alone, the wait lands behind the head layer and drains its stores.
__device__ __forceinline__ void pin_loads3p(InRegs3p& R) {
    static_assert(kInRegs3p == 12, "operand groups below assume 12 float4");
#define B2H_PIN4(TXT, a, b, c, d)                                                                        \
    asm volatile(TXT : "+v"(R.v[a].x), "+v"(R.v[a].y), "+v"(R.v[a].z), "+v"(R.v[a].w), "+v"(R.v[b].x),  \
                 "+v"(R.v[b].y), "+v"(R.v[b].z), "+v"(R.v[b].w), "+v"(R.v[c].x), "+v"(R.v[c].y),        \
                 "+v"(R.v[c].z), "+v"(R.v[c].w), "+v"(R.v[d].x), "+v"(R.v[d].y), "+v"(R.v[d].z),        \
                 "+v"(R.v[d].w)::"memory")
    B2H_PIN4("s_waitcnt vmcnt(0)", 0, 1, 2, 3);
    B2H_PIN4("", 4, 5, 6, 7);
    B2H_PIN4("", 8, 9, 10, 11);
#undef B2H_PIN4
}

template <int L, bool FUSED>
__device__ __forceinline__ void layer3p(char* img_h, char* img_l, const char* wlds, const Geom16& g, int T, int lane,
                                        float* __restrict__ yseq, float mul, int nvalid) {
    constexpr int MT = (L == 3) ? 3 : 2;
    constexpr int h = 6 - 2 * L;
    const int tcol = lane & 15, q = lane >> 4;
    const int lo = max(g.s - h, 0), hi = min(g.e + h, T);
    const int ntiles = (hi - lo + 15) >> 4;
    f16x8 Ah[MT][kTaps], Al[MT][kTaps]; // in-channels 8q + j of out-channel slot (lane & 15)
    f32x4 bias[MT];

    const int pin = 8 - 2 * L - g.s; // P(t, L)   = t + pin
    const int pout = pin - 2;        // P(t, L+1) = t + pout
    // Fragment addresses: one pointer per tap into the hi image (the lo image is kImg3p bytes further,
    // a tile step is 16 rows = 1024 B and leaves the swizzle unchanged), advanced once per loop
    // iteration, so every read and write-back below is pointer + immediate offset.
    // (byte offsets from img_h, made opaque once per iteration: left to itself hipcc's loop strength
    // reduction keeps one address register PER ACCESS and spends 37 v_add_u32 per two tiles on them)
    int rp[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) rp[s] = lds_off<64>(lo + tcol + s - kPad + pin, q);
    int wp = lds_off<64>(lo + tcol + pout, q);

    // head only: this chunk's output rows [s, e) as a buffer, lane byte offset of the current tile
    __amdgpu_buffer_rsrc_t yrs;
    int yoff = 0;
    if constexpr (L == 3) { // lo == s here
        yrs = make_rsrc(yseq + (int64_t)lo * kOutCh, (g.e - lo) * (kOutCh * 4));
        yoff = tcol * (kOutCh * 4) + 16 * q;
    }
    int tq = lo + tcol; // this lane's frame in the tile whose epilogue runs next
    // M: the 30 (45) MFMAs of one tile on fragments already in registers
    auto mma = [&](f32x4 (&acc)[MT], const f16x8 (&Bh)[kTaps], const f16x8 (&Bl)[kTaps]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            if ((B2H_ABLATE & 1024) && T > 0) { // timing probe: one MFMA per tap and M-tile instead of three
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bh[s] + Bl[s], acc[mt], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mt][s], Bh[s], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bl[s], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bh[s], acc[mt], 0, 0, 0);
        }
    };
    // E: epilogue of the tile `k` tiles past the write pointer (k is a compile-time constant inside
    // the loop).  MASK: only the last tile of a layer can hold frames >= T, and the loop below never
    // runs a last tile's epilogue, so its body carries no padding mask and no branch.
    auto epi = [&](const f32x4 (&acc)[MT], int k, bool mask) {
        if ((B2H_ABLATE & 256) && T > 0) { // timing probe: no epilogue (VALU, LDS write-back, stores)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(acc[mt]));
            return;
        }
        if constexpr (L < 3) {
            float v[8]; // channels 8q + 4mt + r = slot 4mt + r of this lane's chunk
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mt * 4 + r] = relu_bits(acc[mt][r]);
            if (mask) { // frames >= T are the zero padding of the next layer
                const bool inside = tq + 16 * k < T;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
            }
            f16x8 oh, ol;
            split8(v, oh, ol);
            *reinterpret_cast<f16x8*>(img_h + (wp + k * 1024)) = oh;
            *reinterpret_cast<f16x8*>(img_h + (wp + k * 1024 + kImg3p)) = ol;
        } else {
            // lane (tcol,q) owns channels 16mt + 4q .. +3 of its frame: 16 B at row offset
            // 168 (t - s) + 64 mt + 16 q; frames >= e fall outside the descriptor.  All varying offsets
            // sit in the VGPR offset (immediate soffset), so hipcc pads the store-data hazard itself.
            const bool dead = FUSED && (tq + 16 * k >= nvalid); // tail mask (per lane)
            const int vo = yoff + k * (16 * kOutCh * 4);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                f32x4 v = acc[mt];
                if constexpr (FUSED) {
                    v = v * mul;                               // x factor, or x 1.0f (exact)
                    if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (mt < 2 || q < 2)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, vo + 64 * mt, 0, 0);
                else if (q == 2) // channels 40, 41 (elements passed BY VALUE, see kernel_mfma16.h)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, yrs,
                                                          vo + 64 * mt, 0, 0);
            }
        }
    };
    // F: the ten fragments of the tile `k` tiles past the read pointers.  Unconditional: past the last
    // tile it reads rows that nobody uses (LDS reads beyond the allocation return 0).
    auto fetch = [&](f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps], int k) {
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            if ((B2H_ABLATE & 512) && k >= 2) continue; // timing probe: no fragment reads past the prologue
            Bh[s] = *reinterpret_cast<const f16x8*>(img_h + (rp[s] + k * 1024));
            Bl[s] = *reinterpret_cast<const f16x8*>(img_h + (rp[s] + k * 1024 + kImg3p));
        }
    };
    auto advance = [&](int tiles) {
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            rp[s] += tiles * 1024;
            asm volatile("" : "+v"(rp[s]));
        }
        wp += tiles * 1024;
        asm volatile("" : "+v"(wp));
        yoff += tiles * (16 * kOutCh * 4);
        tq += tiles * 16;
    };
    // One MFMA at a time with the vector and LDS work of the neighbouring tile spread between them:
    // on this chip a run of MFMAs followed by a run of VALU costs the SUM of both (the compiler's own
    // order left the matrix pipe 52 % busy); 1 MFMA : ~1.5 VALU keeps both issuing.
    auto interleave = [&]() {
        if (B2H_ABLATE & 2048) return; // timing probe: the compiler's own order
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
    // Software pipeline over tiles, two deep (as kernel_mfma16.h): fragments are read two tiles ahead
    // (ping-pong B0/B1) and a tile's epilogue runs one tile late (ping-pong accA/accB), beside the
    // following tile's MFMAs.  Legal in the in-place image: tile m writes rows [tau-2, tau+14) of the
    // next image, every fragment read issued before that write belongs to tiles <= m+2, and tiles > m
    // read rows >= tau+14.  Pointers sit at tile m-1 at the top of the loop body.
    f16x8 B0h[kTaps], B0l[kTaps], B1h[kTaps], B1l[kTaps];
    f32x4 accA[MT], accB[MT];
    fetch(B0h, B0l, 0);
    fetch(B1h, B1l, 1);
    // This layer's weight fragments (hi, lo) and biases, LDS -> registers, ALL requested back to back
    // behind the first two tiles' fragments and waited for ONCE: left alone hipcc sinks each read next
    // to the MFMA that first uses it and the first tile's 30 MFMAs each wait out a full LDS latency
    // (~3 000 of a layer's ~6 300 cycles in the s_memtime stamps, tools/conv3_stamps.py).
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            Ah[mt][s] = *reinterpret_cast<const f16x8*>(wlds + kWLayerOff3p[L] + ((mt * kTaps + s) * 2 + 0) * kWFrag3p + lane * 16);
            Al[mt][s] = *reinterpret_cast<const f16x8*>(wlds + kWLayerOff3p[L] + ((mt * kTaps + s) * 2 + 1) * kWFrag3p + lane * 16);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        bias[mt] = *reinterpret_cast<const f32x4*>(wlds + kBiasOff3p[L] + (mt * 4 + q) * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma(accA, B0h, B0l); // tile 0
    fetch(B0h, B0l, 2);
    int m = 1;
#pragma unroll 1
    for (; m + 1 < ntiles; m += 2) {
        mma(accB, B1h, B1l); epi(accA, 0, false); fetch(B1h, B1l, 3); interleave();
        mma(accA, B0h, B0l); epi(accB, 1, false); fetch(B0h, B0l, 4); interleave();
        advance(2);
    }
    if (m < ntiles) { mma(accB, B1h, B1l); epi(accA, 0, false); epi(accB, 1, true); }
    else epi(accA, 0, true);
    if constexpr (L < 3) {
        if (hi == T) { // sequence end: next layer reads frames T, T+1 as zeros
            const int t = T + ((lane >> 2) & 1);
            if (lane < 16 && t >= lo + 16 * ntiles)
                *reinterpret_cast<f32x4*>((lane < 8 ? img_h : img_l) + lds_off<64>(t + pout, lane & 3)) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <bool FUSED>
__global__ __launch_bounds__(64 * kWaves3p, 1) void b2h_fwd_mfma_f16x3p(
    const float* __restrict__ x, float* __restrict__ y, int T, int cps, int TT, int64_t nchunks,
    const void* __restrict__ wpacked, int pos_emb, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem3p[];
    // hi + lo weights and biases of all four layers: one copy per workgroup
    for (int i = threadIdx.x; i < kPacked3p / 16; i += 64 * kWaves3p)
        reinterpret_cast<uint4*>(smem3p)[i] = reinterpret_cast<const uint4*>(wpacked)[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    char* img_h = smem3p + kPacked3p + wave * (2 * kImg3p);
    char* img_l = img_h + kImg3p;
    const int64_t stride = (int64_t)gridDim.x * kWaves3p;
    int64_t chunk = blockIdx.x + (int64_t)gridDim.x * wave; // consecutive chunks -> different CUs
    if (chunk >= nchunks) return;

    auto src_of = [&](const Geom16& gg) { return x + (gg.seq * (int64_t)T + gg.in_lo) * kInCh; };
    Fused16 fu;
    {
        const int flags = FUSED ? __builtin_amdgcn_readfirstlane(fa.flags) : 0;
        fu.chest = flags & kPreChest;
        fu.norm = flags & kPreNorm;
        fu.factor = fa.factor;
        fu.mul = (flags & kPostDenorm) ? fa.factor : 1.0f;
        fu.mask = (flags & kPostMask) && fa.n_frames;
    }
    InRegs3p R;
    Geom16 g = geom16(chunk, cps, TT, T);
    issue_loads3p(R, src_of(g), g.nf4 * 16, lane);
    int dbg_iter = 0; // development stamps only (dev/b2h_dev.h); dead in the shipped build
    (void)dbg_iter;
    while (true) {
        B2H_STAMP3P(0);
        commit3p<FUSED>(R, img_h, img_l, g, T, lane, pos_emb, src_of(g), fu);
        B2H_STAMP3P(1);
        const int64_t next = chunk + stride;
        const bool more = next < nchunks;
        // prefetch the next chunk; unconditional (an empty buffer when nothing is left) so that the
        // register lifetimes below do not depend on control flow
        const Geom16 gn = more ? geom16(next, cps, TT, T) : g;
        issue_loads3p(R, src_of(gn), more ? gn.nf4 * 16 : 0, lane); // flies under the four layers
        float* yseq = y + g.seq * (int64_t)T * kOutCh;
        int nvalid = T;
        if constexpr (FUSED)
            if (fu.mask) nvalid = (int)min((int64_t)T, max((int64_t)0, fa.n_frames[g.seq]));
        B2H_STAMP3P(2);
        layer3p<0, FUSED>(img_h, img_l, smem3p, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP3P(3);
        layer3p<1, FUSED>(img_h, img_l, smem3p, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP3P(4);
        layer3p<2, FUSED>(img_h, img_l, smem3p, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP3P(5);
        // The prefetch has had three layers to land.  Wait for it HERE: the only vector-memory
        // operations in flight are those loads and the previous chunk's (older) stores, so the wait
        // covers nothing younger; behind the head it would drain this chunk's output stores.
        pin_loads3p(R);
        B2H_STAMP3P(6);
        layer3p<3, FUSED>(img_h, img_l, smem3p, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP3P(7);
        ++dbg_iter;
        if (!more) break;
        chunk = next;
        g = gn;
    }
}

} // namespace b2h
