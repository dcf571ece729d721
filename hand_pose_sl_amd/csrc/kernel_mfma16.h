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
//     prefetch : the NEXT chunk's rows are requested from HBM into registers (20 x 16 B
//                per lane), fly under layers 1-2, and are cast to 16 bit in registers
//                before the register-hungry head
//     layers   : per 16-frame tile 5 ds_read_b128 (one per tap) feed 10 (15 for the
//                head) v_mfma_f32_16x16x32; D = W[chan][(tap,ch)] x Act[(tap,ch)][time]
//                starts from the bias fragment; ReLU (integer max), zero-padding
//                mask (last tile only) and the 16-bit cast stay in registers; one
//                ds_write_b128 per lane puts the tile back, 2 rows lower (in-place
//                image, see kernel_mfma.h); the head's fp32 tile (16 x 168 B) turns once through
//                already-consumed rows of the image and leaves as three lane-linear stores.
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
constexpr int kLds16 = kPacked16 + kWaves16 * kWaveLds16; // 161344; + 16 for the chunk queue <= 163840
constexpr int kLdsAlloc16 = kLds16 + 16;
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

// (chunk indices fit 32 bits: the host refuses launches with 2^31 chunks or more, and a whole-sequence
// chunking -- every T <= 208 -- needs no division at all; the 64-bit one cost ~150 scalar instructions
// per chunk)
__device__ __forceinline__ Geom16 geom16(int64_t chunk, int cps, int TT, int T) {
    Geom16 g;
    const unsigned ch = (unsigned)chunk;
    const unsigned sq = cps == 1 ? ch : ch / (unsigned)cps;
    g.seq = sq;
    const int c = (int)(ch - sq * (unsigned)cps);
    g.s = c * TT;
    g.e = min(g.s + TT, T);
    g.in_lo = max(g.s - kHalo, 0);
    g.nf4 = (min(g.e + kHalo, T) - g.in_lo) * (kInCh / 4);
    return g;
}

struct InRegs { float4 v[kInRegs]; };   // next chunk's rows as loaded (fp32)
struct InRegs16 { uint2 p[kInRegs]; };  // the same, cast to 4 x 16-bit

// Request a chunk's input rows: 20 x 16 B per lane, lane-contiguous (coalesced 1 KiB
// per instruction).  bytes == 0 (nothing left to prefetch) issues no memory traffic.
// AUX: cache policy (kLdStream for a streaming launch; 0 when the rows are read again: the fused
// chest difference re-reads two floats of every row, and tiny launches gain nothing from it).
template <int AUX>
__device__ __forceinline__ void issue_loads16(InRegs& R, const float* base, int bytes, int lane) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);
#pragma unroll
    for (int j = 0; j < kInRegs; ++j) {
        const i32x4 r = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, AUX));
        R.v[j] = __builtin_bit_cast(float4, r);
    }
}

__device__ __forceinline__ void pin_loads16(InRegs& R) {
    static_assert(kInRegs == 20, "operand groups below assume 20 float4");
#define B2H_PIN4(TXT, a, b, c, d)                                                                        \
    asm volatile(TXT : "+v"(R.v[a].x), "+v"(R.v[a].y), "+v"(R.v[a].z), "+v"(R.v[a].w), "+v"(R.v[b].x),  \
                 "+v"(R.v[b].y), "+v"(R.v[b].z), "+v"(R.v[b].w), "+v"(R.v[c].x), "+v"(R.v[c].y),        \
                 "+v"(R.v[c].z), "+v"(R.v[c].w), "+v"(R.v[d].x), "+v"(R.v[d].y), "+v"(R.v[d].z),        \
                 "+v"(R.v[d].w)::"memory")
    B2H_PIN4("s_waitcnt vmcnt(0)", 0, 1, 2, 3);
    B2H_PIN4("", 4, 5, 6, 7);
    B2H_PIN4("", 8, 9, 10, 11);
    B2H_PIN4("", 12, 13, 14, 15);
    B2H_PIN4("", 16, 17, 18, 19);
#undef B2H_PIN4
}

__device__ __forceinline__ void pin_regs16(InRegs16& Q) {
#pragma unroll
    for (int j = 0; j < kInRegs; j += 4)
        asm volatile("" : "+v"(Q.p[j].x), "+v"(Q.p[j].y), "+v"(Q.p[j + 1].x), "+v"(Q.p[j + 1].y),
                     "+v"(Q.p[j + 2].x), "+v"(Q.p[j + 2].y), "+v"(Q.p[j + 3].x), "+v"(Q.p[j + 3].y)::"memory");
}

// Three load iterations (192 float4) cover exactly 32 rows, so a lane needs only
// three (row, column) pairs: iteration j = 3G + jj touches row 32G + rr[jj].
__device__ __forceinline__ void lane_rows16(int lane, int (&rr)[3], int (&c4)[3]) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int u = lane + 64 * jj;
        rr[jj] = u / 6;
        c4[jj] = u - rr[jj] * 6;
    }
}

// Fused pre/post-processing parameters, resolved to wave-uniform scalars ONCE at kernel
// entry (no per-lane select on a uniform flag anywhere): see FusedArgs.
struct Fused16 {
    bool chest;   // body -= body[:,1]             ChestDifference, steps/utils.py:203-210
    bool norm;    // body /= factor                NormalizeFixedFactor, steps/utils.py:180-190
    float factor;
    float mul;    // factor if de-normalising, else 1.0f (x1.0f is exact)   traintest.py:387-388
    bool mask;    // pred[i, n_frames[i]:] = 0     mask_output, steps/utils.py:309-312
};

// fp32 registers -> packed 16-bit registers (runs mid-chunk, when the loads have long
// landed); the reference's item transforms are applied here when fused.
template <int PREC, bool FUSED>
__device__ __forceinline__ void convert16(const InRegs& R, InRegs16& Q, const float* __restrict__ xrow0,
                                          int nf4, int lane, const Fused16& fu) {
    int rr[3], c4[3];
    if constexpr (FUSED) lane_rows16(lane, rr, c4);
#pragma unroll
    for (int j = 0; j < kInRegs; ++j) {
        float4 v = R.v[j];
        if constexpr (FUSED) {
            if (fu.chest) { // wave-uniform branch
                float2 ch = make_float2(0.f, 0.f);
                if (lane + 64 * j < nf4) // row of this float4, channels 2..3 = joint 1 (chest)
                    ch = *reinterpret_cast<const float2*>(xrow0 + (32 * (j / 3) + rr[j % 3]) * kInCh + 2);
                v.x -= ch.x; v.y -= ch.y; v.z -= ch.x; v.w -= ch.y;
            }
            if (fu.norm) { // wave-uniform branch; true division like the reference
                v.x = v.x / fu.factor; v.y = v.y / fu.factor;
                v.z = v.z / fu.factor; v.w = v.w / fu.factor;
            }
        }
        Q.p[j] = uint2{pack2<PREC>(v.x, v.y), pack2<PREC>(v.z, v.w)};
    }
}

// packed registers -> LDS image of the layer-1 input (P(t,0) = t - s + 8).  Every
// address is one of three per-lane offsets plus a multiple of 32 rows = 2048 B,
// which leaves the swizzle term unchanged.
template <int PREC>
__device__ __forceinline__ void commit16(const InRegs16& Q, char* lds, const Geom16& g, int T,
                                         int lane, int pos_emb) {
    const int P0 = g.in_lo + 8 - g.s; // physical row of the first loaded frame: 0, or 8 at s == 0
    char* base = lds + P0 * 64;      // (P0 is a multiple of 8: swizzle term unchanged)
    int rr[3], c4[3], off[3];
    lane_rows16(lane, rr, c4);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) off[jj] = lds_off<64>(rr[jj], c4[jj] >> 1) + (c4[jj] & 1) * 8;
    // No lane predicate: lanes past the chunk's last float4 hold the zeros the buffer load's
    // range check returned, and their rows (<= row 221 of 224) are either the zero padding
    // after the sequence end or rows no tile of this chunk depends on.
#pragma unroll
    for (int j = 0; j < kInRegs; ++j) {
        constexpr int kGroupBytes = 32 * 64;
        *reinterpret_cast<uint2*>(base + (j / 3) * kGroupBytes + off[j % 3]) = Q.p[j];
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

template <int PREC, int L, bool FUSED, bool STREAM>
__device__ __forceinline__ void layer16p(char* lds, const char* wlds, const Geom16& g, int T,
                                         int lane, float* __restrict__ yseq, float mul, int nvalid) {
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
    // Fragment byte offsets of the tile the loop stands at; a tile step is 16 rows = 1024 B and
    // leaves the swizzle term ((P>>1)&3) unchanged.  They advance once per loop iteration and are made
    // opaque there, so every LDS access below is register + immediate offset (left to itself hipcc's
    // loop strength reduction keeps one address register per access and re-adds each per iteration).
    // (Held as 32-bit LDS pointers, so the registers ARE the addresses and no base add remains.)
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) const vec8 lds_vec8;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    lds_char* rd[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) rd[s] = (lds_char*)(lds + lds_off<64>(lo + tcol + s - kPad + pin, q));
    lds_char* wr = (lds_char*)(lds + lds_off<64>(lo + tcol + pout, q));
    int tq = lo + tcol;  // this lane's frame in the tile the loop stands at
    int sbase = 0;       // head: byte offset of that tile's rows in the output buffer (wave-uniform)

    // head only: this chunk's output rows [s, e) as a buffer (so that byte offsets stay small
    // however long the sequence is), lane byte offset within tile 0
    // The 16 x 168 B of a head tile go through LDS once more so that they leave as three lane-linear
    // stores (1 KiB contiguous per instruction) instead of 16 row segments of 64 B per instruction:
    // the same bytes, 7-8 % faster next to the input stream (tools/membench_seq.hip, "mixed seq linear"
    // vs "head-pattern").  Staging area = bytes [0, 2688 + 16) of the wave's own image: when tile j's
    // epilogue runs, the fragments of tiles <= j + 2 are in registers and tile j + 3 reads rows
    // >= 16 (j + 3), i.e. bytes >= 3072 (lo + pin - 2 == 0 in the head).  LDS operations of one wave
    // execute in order, so neither the read-back nor the next tile's staging needs a wait.
    typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
    __amdgpu_buffer_rsrc_t yrs;
    int yoff = 0, yoff2 = 0;
    lds_char *stw = nullptr, *stw2a = nullptr, *stw2b = nullptr, *strd = nullptr;
    if constexpr (L == 3) { // lo == s here
        yrs = make_rsrc(yseq + (int64_t)lo * kOutCh, (g.e - lo) * (kOutCh * 4));
        yoff = lane * 16;
        yoff2 = lane < (16 * kOutCh * 4 - 2048) / 16 ? lane * 16 : 0x40000000; // third store: 640 B = 40 lanes
        stw = (lds_char*)lds + tcol * (kOutCh * 4) + 16 * q;
        lds_char* dummy = (lds_char*)lds + 16 * kOutCh * 4;   // lanes without channels park their 8 B here
        stw2a = q < 3 ? stw + 128 : dummy;                    // channels 32 + 4q, +1 (q = 2: 40, 41)
        stw2b = q < 2 ? stw + 136 : dummy + 8;                // channels 34 + 4q, +1
        strd = (lds_char*)lds + lane * 16;
    }
    // M: the 10 (15) MFMAs of one tile on fragments already in registers.
    auto mma = [&](f32x4 (&acc)[MT], const vec8 (&Bf)[kTaps]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
#pragma unroll
        for (int s = 0; s < kTaps; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = P::mfma(A[mt][s], Bf[s], acc[mt]);
            }
    };
    // E: epilogue of the tile k tiles past the one the loop stands at (k is a constant at every call).
    // `mask`: only the LAST tile of a layer can hold frames >= T, and the loop below never runs a last
    // tile's epilogue, so its body carries no padding mask and no branch.
    auto epi = [&](const f32x4 (&acc)[MT], int k, bool mask) {
        if constexpr (L < 3) {
            float v[8];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mt * 4 + r] = relu_bits(acc[mt][r]);
            if (mask) { // frames >= T are the zero padding of the next layer
                const bool inside = tq + 16 * k < T;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
            }
            const u32x4 o = {pack2<PREC>(v[0], v[1]), pack2<PREC>(v[2], v[3]), pack2<PREC>(v[4], v[5]),
                             pack2<PREC>(v[6], v[7])};
            *(lds_u32x4*)(wr + k * 1024) = o;
        } else {
            // lane (tcol,q) owns channels 16mt + 4q .. +3 of its frame: 16 B at byte 168 (t - lo) + 64 mt
            // + 16 q of the tile's rows (8-byte aligned: 8-byte LDS writes)
            const bool dead = FUSED && (tq + 16 * k >= nvalid); // tail mask (per lane)
            f32x4 v[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                v[mt] = acc[mt];
                if constexpr (FUSED) {
                    v[mt] = v[mt] * mul;                               // x factor, or x 1.0f (exact)
                    if (dead) v[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                *(lds_f32x2*)(stw + 64 * mt) = f32x2{v[mt][0], v[mt][1]};
                *(lds_f32x2*)(stw + 64 * mt + 8) = f32x2{v[mt][2], v[mt][3]};
            }
            *(lds_f32x2*)stw2a = f32x2{v[2][0], v[2][1]};
            *(lds_f32x2*)stw2b = f32x2{v[2][2], v[2][3]};
            // back out, lane-linear; frames >= e are bytes past the descriptor (range check per dword)
            const int so = sbase + k * (16 * kOutCh * 4);
            u32x4 o[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) o[i] = *(lds_u32x4*)(strd + 1024 * i);
            __builtin_amdgcn_raw_buffer_store_b128(o[0], yrs, yoff, so, STREAM ? kStStream : 0);
            __builtin_amdgcn_raw_buffer_store_b128(o[1], yrs, yoff, so + 1024, STREAM ? kStStream : 0);
            __builtin_amdgcn_raw_buffer_store_b128(o[2], yrs, yoff2, so + 2048, STREAM ? kStStream : 0);
        }
    };
    // F: the five fragments of the tile k tiles ahead.  Unconditional: past the last tile it reads
    // rows that nobody uses (LDS reads beyond the allocation return 0), which keeps the loop free
    // of branches between the MFMAs and the epilogue they overlap with.
    auto fetch = [&](vec8 (&Bf)[kTaps], int k) {
#pragma unroll
        for (int s = 0; s < kTaps; ++s) Bf[s] = *(lds_vec8*)(rd[s] + k * 1024);
    };
    auto advance2 = [&]() { // two tiles on
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            rd[s] += 2048;
            asm volatile("" : "+v"(rd[s]));
        }
        wr += 2048;
        asm volatile("" : "+v"(wr));
        tq += 32;
        sbase += 2 * (16 * kOutCh * 4);
    };
    // Software pipeline over tiles, two deep: fragments are read two tiles ahead (ping-pong
    // B0/B1) and a tile's epilogue runs one tile late (ping-pong accA/accB), next to the
    // following tile's MFMAs, so the matrix pipe does not wait for VALU/LDS work.
    // Legal in the in-place image: tile m writes rows [tau-2, tau+14) of the next image,
    // every fragment read issued before that write belongs to tiles <= m+2, and tiles > m
    // read rows >= tau+14.
    vec8 B0[kTaps], B1[kTaps];
    f32x4 accA[MT], accB[MT];
    fetch(B0, 0);
    fetch(B1, 1);
    mma(accA, B0); // tile 0
    fetch(B0, 2);
    int m = 1; // the offsets stand at tile m - 1
#pragma unroll 1
    for (; m + 1 < ntiles; m += 2) {
        mma(accB, B1); epi(accA, 0, false); fetch(B1, 3);
        mma(accA, B0); epi(accB, 1, false); fetch(B0, 4);
        advance2();
    }
    if (m < ntiles) { mma(accB, B1); epi(accA, 0, false); epi(accB, 1, true); }
    else epi(accA, 0, true);
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

// Work distribution of one launch (filled by the host, b2h_api.hip).
//   pool == nullptr  STATIC: workgroup b (of G) owns the chunks b + G k and its waves draw them in order from a
//                    counter in LDS (small launches, launches under stream capture);
//   pool != nullptr  DYNAMIC: after a first deal (wave w of workgroup b: chunks (8 b + w) run .. + run - 1) every
//                    wave claims RUNS of `run` consecutive chunks from `pool`, a device word that is 0 at launch:
//                    a claim returns c, the run is chunks 8 G run + c .. + run - 1.  pool[1] counts finished
//                    workgroups; the last one to finish resets both words for the next launch on the same stream.
struct Sched16 {
    unsigned* pool;
    unsigned run;
};

// STREAM: the launch is a stream (its rows are touched once and exceed the caches): non-temporal input loads
// (unless FUSED) and non-temporal device-scope output stores, see kLdStream / kStStream in kernel_mfma.h.  The
// host picks it for launches of >= 1 MiB of traffic; a single short sequence is 0.9 us FASTER with the default
// policy (its few stores complete in L2 instead of in memory before the kernel can end).
template <int PREC, bool FUSED, bool STREAM>
__global__ __launch_bounds__(64 * kWaves16, 2) void b2h_fwd_mfma16(
    const float* __restrict__ x, float* __restrict__ y, int T, int cps, int TT, int64_t nchunks,
    const void* __restrict__ wpacked, int pos_emb, FusedArgs fa, Sched16 sched) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    // weights + biases of all four layers: one copy per workgroup
    for (int i = threadIdx.x; i < kPacked16 / 16; i += 64 * kWaves16)
        reinterpret_cast<uint4*>(smem16)[i] = reinterpret_cast<const uint4*>(wpacked)[i];
    // Chunks are CLAIMED, not dealt.  Round 3, in three steps:
    // (1) the two waves of a SIMD do not run at the same speed -- the scheduler arbitrates by age, and
    //     tools/conv16_stamps.py shows waves 4-7 needing 1.55x the cycles of waves 0-3 per chunk -- so with every
    //     eighth chunk dealt to each wave the older waves ran out of work at 77 % of the kernel and left each SIMD
    //     to one wave: the waves of a workgroup draw from a counter in LDS instead (STATIC launches still do);
    // (2) the XCDs do not run at the same speed either (workgroup end times differed by 7 %, by blockIdx % 8);
    // (3) and the memory system rewards a chip that walks through x and y as ONE tight front: the same bytes
    //     stream 8 % faster through a non-persistent grid, whose workgroups the dispatcher hands out in order,
    //     than through persistent workgroups with a fixed stride (tools/membench_sched.hip).  So large launches
    //     are DYNAMIC: every wave claims runs of two consecutive chunks from one device-wide counter, in order --
    //     runs of one make that counter the bottleneck (262 144 same-address atomics per launch: 3.1 ms), runs of
    //     four already spread the front.  A claim is made two chunks ahead (its answer is needed when the chunk
    //     after the next one is prefetched), so neither the LDS atomic nor the global one is ever waited for.
    //     Which wave computes a chunk never changes its result.
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* const queue = (lds_u32*)(smem16 + kLds16);
    if (threadIdx.x == 0) {
        queue[0] = 2 * kWaves16; // the first sixteen are dealt: wave w starts with chunks w and w + 8
        queue[1] = 0;            // waves of this workgroup that are done (dynamic launches)
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    char* lds = smem16 + kPacked16 + wave * kWaveLds16;
    const bool dyn = sched.pool != nullptr; // (wave-uniform)
    const int64_t pool_base = (int64_t)gridDim.x * kWaves16 * sched.run; // first claimed chunk (dynamic launches)
    auto chunk_of = [&](unsigned k) { return blockIdx.x + (int64_t)gridDim.x * k; }; // static launches
    [[maybe_unused]] int64_t it = 0;   // chunks this wave has done (development stamps only)
    int rem = (int)sched.run - 2;      // dynamic launches: chunks of the current run after `next` (run >= 2)
    int64_t chunk = dyn ? ((int64_t)blockIdx.x * kWaves16 + wave) * sched.run : chunk_of(wave);
    if (chunk >= nchunks) return;     // (never dynamic: the host asks for >= 256 chunks per workgroup there)
    int64_t next = dyn ? chunk + 1 : chunk_of(wave + kWaves16);
    B2H_SPAN16(wave, lane, 0, __builtin_amdgcn_s_memrealtime());

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
    InRegs R;
    InRegs16 Q;
    Geom16 g = geom16(chunk, cps, TT, T);
    constexpr int kLd = (STREAM && !FUSED) ? kLdStream : 0;
    issue_loads16<kLd>(R, src_of(g), g.nf4 * 16, lane);
    convert16<PREC, FUSED>(R, Q, src_of(g), g.nf4, lane, fu);
    while (true) {
        B2H_STAMP16(wave, lane, it, 0);
        // claim the chunk after the next one: static launches from the workgroup's share ...
        unsigned kn = 0;
        if (!dyn && lane == 0) kn = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        commit16<PREC>(Q, lds, g, T, lane, pos_emb);
        kn = __builtin_amdgcn_readfirstlane(kn);
        // ... dynamic ones, when `next` ends its run, a new run from the device-wide counter.  The atomic is issued BEFORE the prefetch loads, so
        // the wait that the loads need anyway (pin_loads16, after layer 2) covers it: vmcnt counts in order.
        // (asm: hipcc's atomic optimizer wraps a __hip_atomic_fetch_add in a wave reduction whose readfirstlane
        // waits for the answer on the spot -- vmcnt(0) here, i.e. for every store of the previous chunk; for
        // the LDS claim above that wait is ~100 cycles and measured nothing)
        const bool pooled = dyn && rem == 0;
        unsigned pv = 0;
        if (pooled && lane == 0)
            asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(pv) : "v"(0u), "v"(sched.run), "s"(sched.pool));
        const bool more = next < nchunks;
        // prefetch the next chunk; unconditional (an empty buffer when nothing is left)
        // so that the register lifetimes below do not depend on control flow
        const Geom16 gn = more ? geom16(next, cps, TT, T) : g;
        const int nf4n = more ? gn.nf4 : 0;
        B2H_STAMP16(wave, lane, it, 1);
        issue_loads16<kLd>(R, src_of(gn), nf4n * 16, lane); // flies under layers 1-2
        float* yseq = y + g.seq * (int64_t)T * kOutCh;
        int nvalid = T;
        if constexpr (FUSED)
            if (fu.mask) nvalid = (int)min((int64_t)T, max((int64_t)0, fa.n_frames[g.seq]));
        B2H_STAMP16(wave, lane, it, 2);
        layer16p<PREC, 0, FUSED, STREAM>(lds, smem16, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP16(wave, lane, it, 3);
        layer16p<PREC, 1, FUSED, STREAM>(lds, smem16, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP16(wave, lane, it, 4);
        // The prefetch has had two layers to land.  Wait for it HERE -- the only vector-memory
        // operations still in flight are those loads (with the pool claim before them) and the previous
        // chunk's (older) stores, so vmcnt(0) does not wait for anything younger -- and cast it to 16 bit
        // now (80 -> 40 registers before the wide head).  The asm operands pin both the wait and the cast to
        // this point: left alone, hipcc sinks the cast below the head and its wait then also
        // drains this chunk's 52 output stores.
        pin_loads16(R);
        asm volatile("" : "+v"(pv)); // the claim's answer is read after that wait, never before
        const int64_t next2 = pooled ? pool_base + __builtin_amdgcn_readfirstlane(pv) : dyn ? next + 1 : chunk_of(kn);
        rem = pooled ? (int)sched.run - 1 : rem - 1;
        B2H_STAMP16(wave, lane, it, 5);
        convert16<PREC, FUSED>(R, Q, src_of(gn), nf4n, lane, fu);
        pin_regs16(Q);
        layer16p<PREC, 2, FUSED, STREAM>(lds, smem16, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP16(wave, lane, it, 6);
        layer16p<PREC, 3, FUSED, STREAM>(lds, smem16, g, T, lane, yseq, fu.mul, nvalid);
        B2H_STAMP16(wave, lane, it, 7);
        if (!more) break; // claims only grow: next2 is past the end as well
        chunk = next;
        next = next2;
        ++it;
        g = gn;
    }
    B2H_SPAN16(wave, lane, 1, __builtin_amdgcn_s_memrealtime());
    B2H_SPAN16(wave, lane, 2, (unsigned long long)(it + 1));
    // Dynamic launches: the last wave of the last workgroup to finish leaves the pool words at 0 for the next
    // launch on this stream (every claim of every other wave has returned by then: a wave counts itself done
    // only after its last claim was consumed).
    if (sched.pool != nullptr && lane == 0) {
        const unsigned w = __hip_atomic_fetch_add(queue + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w == kWaves16 - 1) {
            const unsigned d = __hip_atomic_fetch_add(sched.pool + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1) {
                __hip_atomic_store(sched.pool, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sched.pool + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

} // namespace b2h
