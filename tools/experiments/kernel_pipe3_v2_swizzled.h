// fp32-grade ConvModel as a LAYER PIPELINE of waves (B2H_KERNEL_F16X3_MFMA on long streams).
//
// Path: ConvModel.forward, HandPoseModels.py:40-64.  Arithmetic: kernel_mfma3.h (every operand split
// into f16 hi + lo, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate) -- the same MFMAs in
// the same order per output, so the results are bit-identical to b2h_fwd_mfma_f16x3.  What changes is
// who does what:
//
//   one 512-thread workgroup per CU = two pipelines of four waves; wave j of a pipeline IS layer j:
//   it loads its layer's hi + lo weight fragments into registers ONCE per launch and then streams
//   16-frame tiles of whole sequences through them, taking its input from a ring of activation rows
//   in LDS that the wave of layer j - 1 fills and handing its output to the ring of layer j + 1
//   (the head stores to HBM).  The front wave (layer 0) also brings the input rows in from HBM, four
//   32-row groups ahead, and announces the sequences of the stream.
//
// Against the wave-per-chunk kernel this removes, per sequence: the weight fragments re-read from L2
// by every wave for every chunk and layer (90 KB per chunk: 12 GB per launch of 65 536 x 200), the
// +-8-frame halo recompute of chunking (+6 % MFMAs; there are no chunks: a sequence of any length is
// one stream), the four pipeline fills per chunk, and the input staging stall.
//
// Stream and rings.  Every stage numbers the pipeline's tiles identically: stream tile 0 is a zero
// tile, then each sequence contributes ceil(T/16) data tiles followed by one zero tile.  A data tile
// holds frames 16i .. 16i+15 of its sequence as rows [frame][32 ch] f16 (hi image and lo image, the
// swizzled 64-B rows of kernel_mfma.h) with frames >= T written as zeros; together with the zero tiles
// these are exactly the zero rows t = -2, -1, T, T+1 that each Conv1d pads with, so no stage ever
// tests a boundary.  Ring r (r = 1..3) holds the kPipeNT = 8 most recent stream tiles of layer r's
// input; ring 0 is the front's own.  Stream tile g lives in slot g mod 8.
//
// Hand-off: two monotonic counters per ring in LDS, each written by one wave only --
//   wr[r] = stream tiles completely written (by stage r-1),   rd[r] = lowest tile its reader still needs.
// A reader computes tile g once wr >= g + 2 (it needs rows of tiles g-1, g, g+1), a writer fills tile g
// once g < rd + 8.  LDS operations of one wave execute in issue order, so a counter is simply stored
// after the rows it covers (no wait in between) and a reader that has seen it reads those rows.
// Every wait is bounded (kPipeSpin polls); running out sets the pipeline's error word, bumps
// `faults` in device memory and makes every stage leave, so a logic error cannot hang the GPU.
#pragma once
#include "kernel_mfma3.h"

namespace b2h {

constexpr int kPipeNT = 8;                        // ring capacity in tiles
constexpr int kPipeImg = kPipeNT * 16 * 64;       // 8192 B: one image (hi or lo) of one ring
constexpr int kPipeRing = 2 * kPipeImg;           // hi | lo
constexpr int kPipeCtlBytes = 256;
constexpr int kPipeLds = 2 * 4 * kPipeRing + 2 * kPipeCtlBytes; // two pipelines: 131 584 B
constexpr int kPipeFifo = 32;                     // announced sequences a pipeline can hold (it holds < 16)
constexpr int kPipeDepth = 4;                     // input groups (32 rows) in flight per front wave
constexpr unsigned kPipeSpin = 1u << 22;          // polls before a wait gives up (~1 s)
constexpr unsigned kRowMask = 0x1FC0u;            // byte offset of a row inside an image: bits 6..12

struct PipeCtl {
    unsigned wr[4];
    unsigned rd[4];
    unsigned seq_pub;     // sequences announced by the front: fifo entries below this are valid
    unsigned error;
    unsigned pad[6];
    int fifo[kPipeFifo];  // sequence index, -1 = end of the stream
};
static_assert(sizeof(PipeCtl) <= kPipeCtlBytes, "control block");

struct PipeArgs {
    const float* x;
    float* y;
    int T;
    int64_t nseq;
    MfmaParams mp;
    FusedArgs fa;
    unsigned* faults;     // device word: waits that ran out (0 after every correct launch)
};

typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) char lds_ch;
typedef __attribute__((address_space(3))) f16x8 lds_h8;
typedef __attribute__((address_space(3))) const f16x8 lds_ch8;
typedef __attribute__((address_space(3))) f16x4 lds_h4;

struct PipeWave {
    lds_ch* lds;        // the workgroup's dynamic LDS (address 0: every offset below is absolute)
    unsigned ring0;     // byte offset of this pipeline's ring 0 (8192-aligned, like every image)
    unsigned ctl;       // byte offset of this pipeline's PipeCtl
    int lane, tcol, q;
    int T, ntseq;
    unsigned* faults;
#if B2H_ABLATE & 131072
    mutable unsigned long long acc[4];
#endif
    __device__ __forceinline__ lds_u32* word(unsigned off) const { return (lds_u32*)(lds + ctl + off); }
    __device__ __forceinline__ lds_u32* wr(int r) const { return word(4 * r); }
    __device__ __forceinline__ lds_u32* rd(int r) const { return word(16 + 4 * r); }
    __device__ __forceinline__ lds_u32* seq_pub() const { return word(32); }
    __device__ __forceinline__ lds_u32* error() const { return word(36); }
    __device__ __forceinline__ lds_i32* fifo(unsigned n) const { return (lds_i32*)(lds + ctl + 64 + 4 * (n & (kPipeFifo - 1))); }
    __device__ __forceinline__ unsigned ring(int r) const { return ring0 + r * kPipeRing; }
};

__device__ __forceinline__ unsigned flag_load(lds_u32* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// one lane stores; the compiler barriers keep the store behind the LDS writes it publishes
// (the hardware executes one wave's LDS operations in issue order)
__device__ __forceinline__ void flag_store(const PipeWave& w, lds_u32* p, unsigned v) {
    asm volatile("" ::: "memory");
    if (w.lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// wait until the monotonic counter *p has reached `need`; `seen` caches its last value
__device__ __forceinline__ bool wait_ge(const PipeWave& w, lds_u32* p, unsigned need, unsigned& seen, int slot = 1) {
    if ((int)(seen - need) >= 0) return true;
    B2H_PIPE_T0();
#pragma unroll 1
    for (unsigned spin = 0; spin < kPipeSpin; ++spin) {
        seen = flag_load(p);
        if ((int)(seen - need) >= 0) {
#if B2H_ABLATE & 131072
            w.acc[slot] += __builtin_amdgcn_s_memtime() - pipe_t0_;
#endif
            return true;
        }
        if ((spin & 255u) == 255u && flag_load(w.error()) != 0) return false; // another stage gave up
        __builtin_amdgcn_s_sleep(1);
    }
    flag_store(w, w.error(), 1u);
    if (w.lane == 0) atomicAdd(w.faults, 1u);
    return false;
}

// zero tile g of ring r: 16 rows x 4 chunks x 2 images, one 16-B write per lane and image
__device__ __forceinline__ void zero_tile(const PipeWave& w, int r, unsigned g) {
    const unsigned off = w.ring(r) + ((g & (kPipeNT - 1)) << 10) + w.lane * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *(__attribute__((address_space(3))) f32x4*)(w.lds + off) = z;
    *(__attribute__((address_space(3))) f32x4*)(w.lds + off + kPipeImg) = z;
}

// A tile of the stream as a stage sees it.
struct PipeTile {
    unsigned g;   // stream tile
    unsigned n;   // sequence number in this pipeline's stream
    int i;        // tile inside the sequence
    int sid;      // sequence index in the batch
};

// The stage that follows `cur`: the next tile of its sequence, or tile 0 of the next announced
// sequence (skipping the zero tile between them).  False at the end of the stream or on a failed wait.
__device__ __forceinline__ bool pipe_next(const PipeWave& w, const PipeTile& cur, PipeTile& nxt, unsigned& pub_seen) {
    nxt = cur;
    ++nxt.i;
    ++nxt.g;
    if (nxt.i < w.ntseq) return true;
    if (!wait_ge(w, w.seq_pub(), cur.n + 2, pub_seen, 3)) return false;
    const int sid = __builtin_amdgcn_readfirstlane(*w.fifo(cur.n + 1));
    if (sid < 0) return false;
    nxt.n = cur.n + 1;
    nxt.i = 0;
    nxt.g = cur.g + 2;
    nxt.sid = sid;
    return true;
}

// ---- stages 1..3 ---------------------------------------------------------------------------------
template <int L, bool FUSED>
__device__ __forceinline__ void pipe_stage(const PipeWave& w, const PipeArgs& a) {
    constexpr int MT = (L == 3) ? 3 : 2;
    f16x8 Ah[MT][kTaps], Al[MT][kTaps];
    f32x4 bias[MT];
    {
        const f16x8* wp = reinterpret_cast<const f16x8*>(a.mp.w[L]); // [mt][tap][hi|lo][lane]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s) {
                Ah[mt][s] = wp[((mt * kTaps + s) * 2 + 0) * 64 + w.lane];
                Al[mt][s] = wp[((mt * kTaps + s) * 2 + 1) * 64 + w.lane];
            }
        const f32x4* bp = reinterpret_cast<const f32x4*>(a.mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + w.q];
    }
    // fragment addresses: row (16 g + tcol + s - 2) mod 128 of ring L, chunk q (swizzled by the row)
    unsigned rbase[kTaps], rk[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) {
        const int r = w.tcol + s - kPad;
        rbase[s] = (unsigned)(r * 64);
        rk[s] = w.ring(L) | (unsigned)((w.q ^ ((r >> 1) & 3)) << 4);
    }
    // write-back address (hidden stages): row (16 g + tcol) mod 128 of ring L + 1, chunk q
    const unsigned wbase = (unsigned)(w.tcol * 64);
    const unsigned wk = w.ring(L < 3 ? L + 1 : L) | (unsigned)((w.q ^ ((w.tcol >> 1) & 3)) << 4);

    unsigned wr_seen = 0, rd_seen = 0, pub_seen = 0;
    auto fetch = [&](f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps], unsigned g) {
        const unsigned gs = g << 10;
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            const unsigned ad = ((rbase[s] + gs) & kRowMask) | rk[s];
            Bh[s] = *(lds_ch8*)(w.lds + ad);
            Bl[s] = *(lds_ch8*)(w.lds + ad + kPipeImg);
        }
    };
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
    // what happens to a finished tile; false = a wait ran out
    auto retire = [&](const f32x4 (&acc)[MT], const PipeTile& t) -> bool {
        const bool last = t.i == w.ntseq - 1;
        if constexpr (L < 3) {
            // room in ring L + 1 for tile t.g (and for the zero tile behind a sequence's last tile)
            const unsigned top = t.g + (last ? 1u : 0u);
            if (top >= (unsigned)kPipeNT && !wait_ge(w, w.rd(L + 1), top - kPipeNT + 1, rd_seen, 2)) return false;
            float v[8]; // channels 8q + 4mt + r = slot j = 4mt + r of this lane's chunk
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
            if (last) { // frames >= T are zero rows of the next layer's input
                const bool inside = 16 * t.i + w.tcol < w.T;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
            }
            f16x8 oh, ol;
            split8(v, oh, ol);
            const unsigned ad = ((wbase + (t.g << 10)) & kRowMask) | wk;
            *(lds_h8*)(w.lds + ad) = oh;
            *(lds_h8*)(w.lds + ad + kPipeImg) = ol;
            flag_store(w, w.wr(L + 1), t.g + 1);
            if (last) {
                zero_tile(w, L + 1, t.g + 1);
                flag_store(w, w.wr(L + 1), t.g + 2);
            }
        } else {
            // lane (tcol, q) owns channels 16mt + 4q .. +3 of frame 16 i + tcol: 16 B at byte 168 t + 64 mt + 16 q of
            // the sequence's output rows; frames >= T fall outside the descriptor and are dropped
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.y + (int64_t)t.sid * w.T * kOutCh, w.T * (kOutCh * 4));
            const int vo = (16 * t.i + w.tcol) * (kOutCh * 4) + 16 * w.q;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const f32x4 v = acc[mt];
                if (mt < 2 || w.q < 2)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 64 * mt, 0, 0);
                else if (w.q == 2) // channels 40, 41 (elements passed BY VALUE: see kernel_mfma16.h)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, rs, vo + 64 * mt, 0, 0);
            }
        }
        return true;
    };

    if constexpr (L < 3) { // stream tile 0
        zero_tile(w, L + 1, 0);
        flag_store(w, w.wr(L + 1), 1u);
    }
    if (!wait_ge(w, w.seq_pub(), 1u, pub_seen, 3)) return;
    PipeTile cur;
    cur.sid = __builtin_amdgcn_readfirstlane(*w.fifo(0));
    if (cur.sid < 0) return;
    cur.g = 1; cur.n = 0; cur.i = 0;

    f16x8 B0h[kTaps], B0l[kTaps], B1h[kTaps], B1l[kTaps];
    PipeTile nxt;
    if (!wait_ge(w, w.wr(L), cur.g + 2, wr_seen)) return;
    fetch(B0h, B0l, cur.g);
    flag_store(w, w.rd(L), cur.g);
    // Two tiles per iteration (static ping-pong B0/B1): the next tile's fragments are requested before this
    // tile's MFMAs.  Hidden stages also ping-pong the accumulators and retire a tile one tile late, beside the
    // following tile's MFMAs (ReLU, the hi/lo split and the ring write are ~40 vector instructions); the head
    // stores straight from its accumulators (a second set would not fit beside its 120 weight registers).
    if constexpr (L < 3) {
        f32x4 accA[MT], accB[MT];
        PipeTile pend;
        bool has_pend = false;
#pragma unroll 1
        while (true) {
            bool more = pipe_next(w, cur, nxt, pub_seen);
            if (more) {
                if (!wait_ge(w, w.wr(L), nxt.g + 2, wr_seen)) return;
                fetch(B1h, B1l, nxt.g);
                flag_store(w, w.rd(L), nxt.g);
            }
            mma(accA, B0h, B0l);
            if (has_pend && !retire(accB, pend)) return;
            pend = cur; has_pend = true;
            if (!more) { if (flag_load(w.error()) == 0) (void)retire(accA, pend); return; }
            cur = nxt;
            more = pipe_next(w, cur, nxt, pub_seen);
            if (more) {
                if (!wait_ge(w, w.wr(L), nxt.g + 2, wr_seen)) return;
                fetch(B0h, B0l, nxt.g);
                flag_store(w, w.rd(L), nxt.g);
            }
            mma(accB, B1h, B1l);
            if (!retire(accA, pend)) return;
            pend = cur;
            if (!more) { if (flag_load(w.error()) == 0) (void)retire(accB, pend); return; }
            cur = nxt;
        }
    } else {
        f32x4 acc[MT];
#pragma unroll 1
        while (true) {
            bool more = pipe_next(w, cur, nxt, pub_seen);
            if (more) {
                if (!wait_ge(w, w.wr(L), nxt.g + 2, wr_seen)) return;
                fetch(B1h, B1l, nxt.g);
                flag_store(w, w.rd(L), nxt.g);
            }
            mma(acc, B0h, B0l);
            (void)retire(acc, cur);
            if (!more) return;
            cur = nxt;
            more = pipe_next(w, cur, nxt, pub_seen);
            if (more) {
                if (!wait_ge(w, w.wr(L), nxt.g + 2, wr_seen)) return;
                fetch(B0h, B0l, nxt.g);
                flag_store(w, w.rd(L), nxt.g);
            }
            mma(acc, B1h, B1l);
            (void)retire(acc, cur);
            if (!more) return;
            cur = nxt;
        }
    }
}

// ---- stage 0: the front --------------------------------------------------------------------------
// One input group = 32 rows of (T, 24) fp32 = 192 float4 = 3 per lane (lane-linear, 1 KiB per instruction;
// rows past the sequence come back as zeros from the range check and are written as such).
struct PipeGroup { float4 v[3]; };

template <bool FUSED>
__device__ __forceinline__ void pipe_front(const PipeWave& w, const PipeArgs& a, int pid, int npipes) {
    constexpr int L = 0, MT = 2;
    f16x8 Ah[MT][kTaps], Al[MT][kTaps];
    f32x4 bias[MT];
    {
        const f16x8* wp = reinterpret_cast<const f16x8*>(a.mp.w[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < kTaps; ++s) {
                Ah[mt][s] = wp[((mt * kTaps + s) * 2 + 0) * 64 + w.lane];
                Al[mt][s] = wp[((mt * kTaps + s) * 2 + 1) * 64 + w.lane];
            }
        const f32x4* bp = reinterpret_cast<const f32x4*>(a.mp.bias[L]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + w.q];
    }
    unsigned rbase[kTaps], rk[kTaps];
#pragma unroll
    for (int s = 0; s < kTaps; ++s) {
        const int r = w.tcol + s - kPad;
        rbase[s] = (unsigned)(r * 64);
        rk[s] = w.ring(0) | (unsigned)((w.q ^ ((r >> 1) & 3)) << 4);
    }
    const unsigned wbase = (unsigned)(w.tcol * 64);
    const unsigned wk = w.ring(1) | (unsigned)((w.q ^ ((w.tcol >> 1) & 3)) << 4);
    // commit addresses: float4 u = lane + 64 jj of a group is channels 4 c4 .. +3 of its row u / 6
    unsigned cbase[3], ck[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int u = w.lane + 64 * jj, row = u / 6, c4 = u - row * 6;
        cbase[jj] = (unsigned)(row * 64);
        ck[jj] = w.ring(0) | (unsigned)((((c4 >> 1) ^ ((row >> 1) & 3)) << 4) + (c4 & 1) * 8);
    }
    const unsigned pk = w.ring(0) | (unsigned)((3 ^ ((w.lane >> 1) & 3)) << 4); // channels 24..31 of row `lane` (< 32)
    const int ngseq = (w.ntseq + 1) >> 1; // groups per sequence

    // -- loader: claims and announces sequences, keeps kPipeDepth groups in flight --
    unsigned ln = 0;       // sequences announced so far
    int lsid = -1, lk = 0; // sequence being loaded and its next group (lk == ngseq: claim the next one)
    bool lend = false;
    auto announce_next = [&]() {
        const int64_t s = (int64_t)pid + (int64_t)npipes * ln;
        lsid = s < a.nseq ? (int)s : -1;
        lend = lsid < 0;
        if (w.lane == 0) *w.fifo(ln) = lsid;
        ++ln;
        flag_store(w, w.seq_pub(), ln);
        lk = 0;
    };
    auto issue = [&](PipeGroup& G) {
        if (!lend && lk == ngseq) announce_next();
        const bool on = !lend;
        const float* base = a.x + ((int64_t)(on ? lsid : 0) * w.T + 32 * (on ? lk : 0)) * kInCh;
        const int rows = on ? min(32, w.T - 32 * lk) : 0;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, rows * (kInCh * 4));
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            G.v[jj] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, w.lane * 16, jj * 1024, 0));
        if (on) ++lk;
    };
    // -- commit: the oldest group in flight -> hi / lo rows of ring 0 --
    unsigned cg0 = 1;  // stream tile of tile 0 of the sequence being committed
    int ck_ = 0;       // its next group
    auto commit = [&](const PipeGroup& G) {
        const unsigned gs = (cg0 + 2 * ck_) << 10;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            f16x4 wh, wl;
            split4(G.v[jj], wh, wl);
            const unsigned ad = ((cbase[jj] + gs) & kRowMask) | ck[jj];
            *(lds_h4*)(w.lds + ad) = wh;
            *(lds_h4*)(w.lds + ad + kPipeImg) = wl;
        }
        if (w.lane < 32) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const unsigned ad = (((unsigned)(w.lane * 64) + gs) & kRowMask) | pk;
            *(__attribute__((address_space(3))) f32x4*)(w.lds + ad) = z;
            *(__attribute__((address_space(3))) f32x4*)(w.lds + ad + kPipeImg) = z;
        }
        ++ck_;
        if (ck_ == ngseq) { // the sequence is in: the zero tile behind it, then on to the next one
            zero_tile(w, 0, cg0 + w.ntseq);
            cg0 += w.ntseq + 1;
            ck_ = 0;
        }
    };

    unsigned rd_seen = 0;
    auto fetch = [&](f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps], unsigned g) {
        const unsigned gs = g << 10;
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            const unsigned ad = ((rbase[s] + gs) & kRowMask) | rk[s];
            Bh[s] = *(lds_ch8*)(w.lds + ad);
            Bl[s] = *(lds_ch8*)(w.lds + ad + kPipeImg);
        }
    };

    zero_tile(w, 0, 0);
    zero_tile(w, 1, 0);
    flag_store(w, w.wr(1), 1u);
    announce_next();
    PipeGroup G[kPipeDepth];
#pragma unroll
    for (int d = 0; d < kPipeDepth; ++d) issue(G[d]);
    if (__builtin_amdgcn_readfirstlane(*w.fifo(0)) < 0) return;

    unsigned g = 1, n = 0; // compute cursor: stream tile, sequence number
    int i = 0;             // next tile of sequence n
    int kk = 0;            // groups of sequence n in ring 0
    bool done = false, failed = false;
    f16x8 Bh[kTaps], Bl[kTaps];
    f32x4 acc[MT];
    auto mma = [&]() {
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
    auto retire = [&](unsigned tg, int ti) -> bool { // acc of stream tile tg = tile ti of the sequence -> ring 1
        const bool last = ti == w.ntseq - 1;
        const unsigned top = tg + (last ? 1u : 0u);
        if (top >= (unsigned)kPipeNT && !wait_ge(w, w.rd(1), top - kPipeNT + 1, rd_seen, 2)) return false;
        float v[8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
        if (last) {
            const bool inside = 16 * ti + w.tcol < w.T;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
        }
        f16x8 oh, ol;
        split8(v, oh, ol);
        const unsigned ad = ((wbase + (tg << 10)) & kRowMask) | wk;
        *(lds_h8*)(w.lds + ad) = oh;
        *(lds_h8*)(w.lds + ad + kPipeImg) = ol;
        flag_store(w, w.wr(1), tg + 1);
        if (last) {
            zero_tile(w, 1, tg + 1);
            flag_store(w, w.wr(1), tg + 2);
        }
        return true;
    };
    // One step = one input group: the oldest group in flight goes into ring 0, its register slot is refilled
    // with the load kPipeDepth groups ahead, and the tiles that group completes are computed: after group kk
    // of a sequence the rows of tiles <= 2 kk + 1 are in, i.e. tiles <= 2 kk can run (a tile needs its right
    // neighbour), and after the last group all that are left.  The slot is a template argument: the groups
    // rotate through the registers by NAME -- moving an in-flight load's destination would wait for it.
    auto step = [&](PipeGroup& slot) {
        commit(slot);
        fetch(Bh, Bl, g);     // tile i is complete now; its fragments travel while the next load is set up
        issue(slot);
        ++kk;
        const int upto = kk == ngseq ? w.ntseq - 1 : 2 * (kk - 1);
        while (true) {
            mma();
            const unsigned tg = g;
            const int ti = i;
            ++i; ++g;
            const bool again = i <= upto;
            if (again) fetch(Bh, Bl, g); // under the retiring tile's vector work
            if (!retire(tg, ti)) { failed = true; return; }
            if (!again) break;
        }
        if (i == w.ntseq) { // the zero tile between sequences, then the next announced sequence
            ++g; ++n; i = 0; kk = 0;
            done = __builtin_amdgcn_readfirstlane(*w.fifo(n)) < 0; // (announced: the loader is a sequence ahead)
        }
    };
    static_assert(kPipeDepth == 4, "the steps below name four register slots");
#pragma unroll 1
    while (true) {
        step(G[0]); if (done || failed) return;
        step(G[1]); if (done || failed) return;
        step(G[2]); if (done || failed) return;
        step(G[3]); if (done || failed) return;
    }
}

template <bool FUSED>
__global__ __launch_bounds__(512, 2) void b2h_fwd_pipe_f16x3(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_pipe[];
    lds_ch* lds = (lds_ch*)smem_pipe;
    // control blocks: everything 0 (counters, error), no sequence announced
    for (int i = threadIdx.x; i < 2 * kPipeCtlBytes / 4; i += 512) ((lds_u32*)(lds + 8 * kPipeRing))[i] = 0u;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PipeWave w;
    w.lds = lds;
    w.lane = threadIdx.x & 63;
    w.tcol = w.lane & 15;
    w.q = w.lane >> 4;
    w.T = a.T;
    w.ntseq = (a.T + 15) >> 4;
    w.faults = a.faults;
    const int p = wave >> 2;
    // waves w and w + 4 share a SIMD: the second pipeline's stages are rotated by two, so that a head
    // (45 MFMAs per tile) never sits beside the front (30 + the input staging)
    const int stage = p == 0 ? (wave & 3) : ((wave + 2) & 3);
    w.ring0 = (unsigned)(p * 4 * kPipeRing);
    w.ctl = (unsigned)(8 * kPipeRing + p * kPipeCtlBytes);
    if ((unsigned)reinterpret_cast<uintptr_t>(lds) != 0u) { // the offsets above are used as absolute LDS addresses
        if (threadIdx.x == 0) atomicAdd(a.faults, 1u);
        return;
    }
    const int pid = blockIdx.x + gridDim.x * p, npipes = 2 * gridDim.x;
#if B2H_ABLATE & 131072
    for (int k = 0; k < 4; ++k) w.acc[k] = 0;
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
    if (stage == 0) pipe_front<FUSED>(w, a, pid, npipes);
    else if (stage == 1) pipe_stage<1, FUSED>(w, a);
    else if (stage == 2) pipe_stage<2, FUSED>(w, a);
    else pipe_stage<3, FUSED>(w, a);
#if B2H_ABLATE & 131072
    if (blockIdx.x == gridDim.x / 2 && w.lane == 0) {
        g_pipe_dbg[wave * 4 + 0] = __builtin_amdgcn_s_memtime() - t_in;
        for (int k = 1; k < 4; ++k) g_pipe_dbg[wave * 4 + k] = w.acc[k];
    }
#endif
}

} // namespace b2h
