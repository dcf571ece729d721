// fp32-grade ConvModel as a LAYER PIPELINE of waves (B2H_KERNEL_F16X3_MFMA on streams of sequences).
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
// holds frames 16i .. 16i+15 of its sequence as rows [frame][32 ch] f16 (a hi image and a lo image)
// with frames >= T written as zeros; together with the zero tiles these are exactly the zero rows
// t = -2, -1, T, T+1 that each Conv1d pads with, so no stage ever tests a boundary.  Ring r (r = 1..3)
// holds the kPipeNT most recent stream tiles of layer r's input, ring 0 (the front's own) kPipeNT0;
// stream tile g lives in slot g mod NT.
//
// Row layout: 96-byte pitch (64 B of channels + 32 B never touched), no swizzle.  On this chip a SIMD
// hides ONE vector instruction per MFMA and pays for every further one (tools/mfma_mix_bench.hip), and the
// first version of this kernel, with the XOR-swizzled 64-B rows of the other kernels, spent 10 vector
// instructions per tile on fragment addresses alone and ran 7 % behind the wave-per-chunk kernel.
// With an affine layout the five taps of a tile are ONE address register plus immediates (s * 96), and
// the 96-B pitch keeps the ds_read_b128 fragment reads conflict-free (the 16 rows of a lane group fall
// on 16 different bank quads).  So that a tile's rows -2 .. 17 never wrap, every image has two guard
// rows on either side: whoever writes rows 14, 15 of the last slot also writes them in front of slot 0,
// and rows 0, 1 of slot 0 also behind the last slot.
//
// Hand-off: two monotonic counters per ring in LDS, each written by one wave only --
//   wr[r] = stream tiles completely written (by stage r-1),   rd[r] = lowest tile its reader still needs.
// A reader computes tile g once wr >= g + 2 (it needs rows of tiles g-1, g, g+1), a writer fills tile g
// once g < rd + NT.  LDS operations of one wave execute in issue order, so a counter is simply stored
// after the rows it covers (no wait in between) and a reader that has seen it reads those rows.
// Every wait is bounded (kPipeSpin polls); running out sets the pipeline's error word, bumps
// `faults` in device memory and makes every stage leave, so a logic error cannot hang the GPU.
#pragma once
#include "kernel_mfma3.h"

namespace b2h {

constexpr int kPipeNT = 6;                        // tiles per ring (rings 1..3)
constexpr int kPipeNT0 = 4;                       // tiles of the front's own ring 0
constexpr int kPipePitch = 96;                    // bytes per row
constexpr int kPipeTileB = 16 * kPipePitch;       // 1536
constexpr int kPipeImg = (16 * kPipeNT + 4) * kPipePitch;   // 9600: one image (hi or lo) incl. guard rows
constexpr int kPipeImg0 = (16 * kPipeNT0 + 4) * kPipePitch; // 6528
constexpr int kPipeRing = 2 * kPipeImg;           // hi | lo
constexpr int kPipeRing0 = 2 * kPipeImg0;
constexpr int kPipePerPipe = kPipeRing0 + 3 * kPipeRing; // 70656
constexpr int kPipeCtlBytes = 256;
constexpr int kPipeLds = 2 * kPipePerPipe + 2 * kPipeCtlBytes; // 141 824 B
constexpr int kPipeFifo = 32;                     // announced sequences a pipeline can hold (it holds < 16)
constexpr int kPipeDepth = 4;                     // input groups (32 rows) in flight per front wave
constexpr unsigned kPipeSpin = 1u << 22;          // polls before a wait gives up (~1 s)

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
typedef __attribute__((address_space(3))) f32x4 lds_f4;

// control block of a pipeline (byte offsets): wr[4] 0.., rd[4] 16.., seq_pub 32, error 36, fifo 64..
struct PipeWave {
    lds_ch* lds;        // the workgroup's dynamic LDS
    unsigned ring0;     // byte offset of this pipeline's ring 0; rings 1..3 follow
    unsigned ctl;       // byte offset of this pipeline's control block
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
    __device__ __forceinline__ unsigned ring(int r) const { return r == 0 ? ring0 : ring0 + kPipeRing0 + (r - 1) * kPipeRing; }
};

__device__ __forceinline__ unsigned flag_load(lds_u32* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// One lane stores (exec = 1 for the one instruction: one vector move for the value instead of the
// compare + mask + branch of `if (lane == 0)`; only ever called from wave-uniform control flow).  The
// hardware executes one wave's LDS operations in issue order, and asm volatile + "memory" keeps the
// compiler from moving the rows' stores past it.
__device__ __forceinline__ void flag_store(lds_u32* p, unsigned v) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"((unsigned)reinterpret_cast<uintptr_t>(p)), "v"(v) : "memory");
}
// wait until the monotonic counter *p has reached `need`; `seen` caches its last value
__device__ __forceinline__ bool wait_slow(const PipeWave& w, lds_u32* p, unsigned need, unsigned& seen, int slot) {
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
    flag_store(w.error(), 1u);
    if (w.lane == 0) atomicAdd(w.faults, 1u);
    return false;
}
__device__ __forceinline__ bool wait_ge(const PipeWave& w, lds_u32* p, unsigned need, unsigned& seen, int slot = 1) {
    if (__builtin_expect((int)(seen - need) >= 0, 1)) return true;
    return wait_slow(w, p, need, seen, slot);
}

// Ring geometry: physical row of (slot, row r) = 2 + 16 slot + r; NT slots; images hi at 0, lo at IMG.
template <int NT> struct RingGeo {
    static constexpr int kImg = (16 * NT + 4) * kPipePitch;
    static __device__ __forceinline__ unsigned next(unsigned slot) { return slot + 1 == (unsigned)NT ? 0u : slot + 1; }
};

// 16 B (or 8 B) of hi and of lo for row r (0..15) of the tile in `slot` of the ring at byte offset `ring`
// -- plus the guard copies of rows 14, 15 of the last slot / rows 0, 1 of slot 0.
// `row_off` = (2 + r) * pitch + byte in the row is a lane constant; slot is wave-uniform.
template <int NT, typename V>
__device__ __forceinline__ void ring_put(const PipeWave& w, unsigned ring, unsigned slot, int r, unsigned row_off,
                                         const V& hi, const V& lo) {
    typedef __attribute__((address_space(3))) V lds_v;
    const unsigned ad = ring + slot * kPipeTileB + row_off;
    *(lds_v*)(w.lds + ad) = hi;
    *(lds_v*)(w.lds + ad + RingGeo<NT>::kImg) = lo;
    if (slot == (unsigned)(NT - 1)) {        // wave-uniform
        if (r >= 14) {
            const unsigned g = ring + row_off - 16 * kPipePitch; // rows 14, 15 -> physical rows 0, 1
            *(lds_v*)(w.lds + g) = hi;
            *(lds_v*)(w.lds + g + RingGeo<NT>::kImg) = lo;
        }
    } else if (slot == 0u) {
        if (r < 2) {
            const unsigned g = ring + row_off + 16 * NT * kPipePitch; // rows 0, 1 -> behind the last slot
            *(lds_v*)(w.lds + g) = hi;
            *(lds_v*)(w.lds + g + RingGeo<NT>::kImg) = lo;
        }
    }
}

// zero tile: 16 rows x 64 B x 2 images, one 16-B write per lane and image (row lane >> 2, chunk lane & 3)
template <int NT>
__device__ __forceinline__ void zero_tile(const PipeWave& w, unsigned ring, unsigned slot) {
    const int r = w.lane >> 2;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ring_put<NT>(w, ring, slot, r, (unsigned)((2 + r) * kPipePitch + (w.lane & 3) * 16), z, z);
}

// the 30 (45) MFMAs of one tile: acc = bias + sum over taps of Wlo.xhi + Whi.xlo + Whi.xhi
template <int MT>
__device__ __forceinline__ void pipe_mma(f32x4 (&acc)[MT], const f32x4 (&bias)[MT], const f16x8 (&Ah)[MT][kTaps],
                                         const f16x8 (&Al)[MT][kTaps], const f16x8 (&Bh)[kTaps], const f16x8 (&Bl)[kTaps]) {
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
}

// fragments of the tile in `slot`: rows tcol - 2 + s (s = 0..4) of chunk q = one address + immediates
template <int NT>
__device__ __forceinline__ void pipe_fetch(const PipeWave& w, unsigned fbase, unsigned slot, f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps]) {
    const unsigned ad = fbase + slot * kPipeTileB;
#pragma unroll
    for (int s = 0; s < kTaps; ++s) {
        Bh[s] = *(lds_ch8*)(w.lds + ad + s * kPipePitch);
        Bl[s] = *(lds_ch8*)(w.lds + ad + s * kPipePitch + RingGeo<NT>::kImg);
    }
}

// the same MFMAs with the fragments of the NEXT tile (at byte offset `nad`, see pipe_fetch) requested tap by tap
// into the registers the taps have just been read from: one fragment set, always a tile ahead
template <int MT, int NT>
__device__ __forceinline__ void pipe_mma_refill(const PipeWave& w, f32x4 (&acc)[MT], const f32x4 (&bias)[MT],
                                                const f16x8 (&Ah)[MT][kTaps], const f16x8 (&Al)[MT][kTaps],
                                                f16x8 (&Bh)[kTaps], f16x8 (&Bl)[kTaps], unsigned nad) {
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
        Bh[s] = *(lds_ch8*)(w.lds + nad + s * kPipePitch);
        Bl[s] = *(lds_ch8*)(w.lds + nad + s * kPipePitch + RingGeo<NT>::kImg);
    }
}

// a hidden layer's accumulators -> ReLU, zero beyond T, hi/lo split -> tile `slot` of the next ring
template <int NT>
__device__ __forceinline__ void pipe_put_hidden(const PipeWave& w, unsigned ring, unsigned slot, unsigned row_off,
                                                const f32x4 (&acc)[2], bool last, int ti) {
    float v[8]; // channels 8q + 4mt + r = slot j = 4mt + r of this lane's chunk
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * mt + r] = relu_bits(acc[mt][r]);
    if (last) { // frames >= T are zero rows of the next layer's input
        const bool inside = 16 * ti + w.tcol < w.T;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.f;
    }
    f16x8 oh, ol;
    split8(v, oh, ol);
    ring_put<NT>(w, ring, slot, w.tcol, row_off, oh, ol);
}

template <int MT>
__device__ __forceinline__ void pipe_weights(const PipeWave& w, const MfmaParams& mp, int L, f16x8 (&Ah)[MT][kTaps],
                                             f16x8 (&Al)[MT][kTaps], f32x4 (&bias)[MT]) {
    const f16x8* wp = reinterpret_cast<const f16x8*>(mp.w[L]); // [mt][tap][hi|lo][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < kTaps; ++s) {
            Ah[mt][s] = wp[((mt * kTaps + s) * 2 + 0) * 64 + w.lane];
            Al[mt][s] = wp[((mt * kTaps + s) * 2 + 1) * 64 + w.lane];
        }
    const f32x4* bp = reinterpret_cast<const f32x4*>(mp.bias[L]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias[mt] = bp[mt * 4 + w.q];
}

// ---- stages 1..3 ---------------------------------------------------------------------------------
// Per sequence: tile 0's fragments are fetched, then every tile's MFMAs refill them for the tile behind it
// (pipe_mma_refill).  The tiles that are not a sequence's last run in PAIRS through a loop whose body has
// no boundary logic at all (two accumulator sets alternate by name, a tile is retired beside the next
// tile's MFMAs); the one or two tiles left over take the general path with the zero-padding mask and the
// zero tile behind the sequence.  Nothing is carried from one sequence to the next: the wave a stage
// shares its SIMD with fills the matrix pipe in the gap.
template <int L, bool FUSED>
__device__ __forceinline__ void pipe_stage(const PipeWave& w, const PipeArgs& a) {
    constexpr int MT = (L == 3) ? 3 : 2;
    constexpr int NT = kPipeNT;
    f16x8 Ah[MT][kTaps], Al[MT][kTaps];
    f32x4 bias[MT];
    pipe_weights<MT>(w, a.mp, L, Ah, Al, bias);
    const unsigned rin = w.ring(L), rout = w.ring(L < 3 ? L + 1 : L);
    const unsigned fbase = rin + (unsigned)(w.tcol * kPipePitch + w.q * 16);       // row tcol - 2 of a slot, chunk q
    const unsigned wrow = (unsigned)((2 + w.tcol) * kPipePitch + w.q * 16);        // row tcol of a slot, chunk q

    unsigned wr_seen = 0, rd_seen = 0, pub_seen = 0;
    unsigned g = 0, slot = 0; // stream tile and its slot (input and output rings have the same geometry)
    int sid = -1, ti = 0;     // sequence in the batch, tile in the sequence
    // a finished tile that is not its sequence's last one
    auto retire = [&](const f32x4 (&acc)[MT], unsigned tg, unsigned ts, int tii) -> bool {
        if constexpr (L < 3) {
            if (tg >= (unsigned)NT && !wait_ge(w, w.rd(L + 1), tg - NT + 1, rd_seen, 2)) return false;
            pipe_put_hidden<NT>(w, rout, ts, wrow, acc, false, tii);
            flag_store(w.wr(L + 1), tg + 1);
        } else {
            // lane (tcol, q) owns channels 16mt + 4q .. +3 of frame 16 i + tcol: 16 B at byte 168 t + 64 mt + 16 q of
            // the sequence's output rows; frames >= T fall outside the descriptor and are dropped
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.y + (int64_t)sid * w.T * kOutCh, w.T * (kOutCh * 4));
            const int vo = (16 * tii + w.tcol) * (kOutCh * 4) + 16 * w.q;
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
    // ... and the last one: frames >= T become zero rows, the zero tile behind the sequence follows
    auto retire_last = [&](const f32x4 (&acc)[MT], unsigned tg, unsigned ts, int tii) -> bool {
        if constexpr (L < 3) {
            if (tg + 1 >= (unsigned)NT && !wait_ge(w, w.rd(L + 1), tg + 1 - NT + 1, rd_seen, 2)) return false;
            pipe_put_hidden<NT>(w, rout, ts, wrow, acc, true, tii);
            flag_store(w.wr(L + 1), tg + 1);
            zero_tile<NT>(w, rout, RingGeo<NT>::next(ts));
            flag_store(w.wr(L + 1), tg + 2);
            return true;
        } else {
            return retire(acc, tg, ts, tii);
        }
    };

    if constexpr (L < 3) { // stream tile 0
        zero_tile<NT>(w, rout, 0u);
        flag_store(w.wr(L + 1), 1u);
    }
    f16x8 Bh[kTaps], Bl[kTaps];
    const int npairs = (w.ntseq - 1) >> 1;     // pairs of tiles none of which is the last
    if constexpr (L < 3) {
        f32x4 accA[MT], accB[MT];
#pragma unroll 1
        for (unsigned n = 0;; ++n) {
            if (!wait_ge(w, w.seq_pub(), n + 1, pub_seen, 3)) return;
            sid = __builtin_amdgcn_readfirstlane(*w.fifo(n));
            if (sid < 0) return;
            ++g; slot = RingGeo<NT>::next(slot);   // past the zero tile: tile 0 of the sequence
            ti = 0;
            if (!wait_ge(w, w.wr(L), g + 2, wr_seen)) return;
            pipe_fetch<NT>(w, fbase, slot, Bh, Bl);
            flag_store(w.rd(L), g);
            bool pendB = false; // accB holds tile ti - 1, not yet retired
#pragma unroll 1
            for (int p = 0; p < npairs; ++p) {
                // tile ti -> accA (tile ti + 1 exists: its fragments replace this tile's)
                const unsigned ns = RingGeo<NT>::next(slot);
                if (!wait_ge(w, w.wr(L), g + 3, wr_seen)) return;
                pipe_mma_refill<MT, NT>(w, accA, bias, Ah, Al, Bh, Bl, fbase + ns * kPipeTileB);
                flag_store(w.rd(L), g + 1);
                if (pendB && !retire(accB, g - 1, slot == 0u ? (unsigned)(NT - 1) : slot - 1, ti - 1)) return;
                // tile ti + 1 -> accB (tile ti + 2 exists)
                const unsigned ns2 = RingGeo<NT>::next(ns);
                if (!wait_ge(w, w.wr(L), g + 4, wr_seen)) return;
                pipe_mma_refill<MT, NT>(w, accB, bias, Ah, Al, Bh, Bl, fbase + ns2 * kPipeTileB);
                flag_store(w.rd(L), g + 2);
                if (!retire(accA, g, slot, ti)) return;
                pendB = true;
                g += 2; ti += 2; slot = ns2;
            }
            // one or two tiles left; Bh/Bl hold tile ti
            const unsigned ps = slot == 0u ? (unsigned)(NT - 1) : slot - 1;
            if (ti == w.ntseq - 1) {
                pipe_mma<MT>(accA, bias, Ah, Al, Bh, Bl);
                if (pendB && !retire(accB, g - 1, ps, ti - 1)) return;
                if (!retire_last(accA, g, slot, ti)) return;
                ++g; slot = RingGeo<NT>::next(slot);
            } else {
                const unsigned ns = RingGeo<NT>::next(slot);
                if (!wait_ge(w, w.wr(L), g + 3, wr_seen)) return;
                pipe_mma_refill<MT, NT>(w, accA, bias, Ah, Al, Bh, Bl, fbase + ns * kPipeTileB);
                flag_store(w.rd(L), g + 1);
                if (pendB && !retire(accB, g - 1, ps, ti - 1)) return;
                pipe_mma<MT>(accB, bias, Ah, Al, Bh, Bl);
                if (!retire(accA, g, slot, ti)) return;
                if (!retire_last(accB, g + 1, ns, ti + 1)) return;
                g += 2; slot = RingGeo<NT>::next(ns);
            }
            // g, slot now stand at the zero tile behind the sequence
        }
    } else {
        // the head: one accumulator set (a second would not fit beside its 120 weight registers), stored as it is
        f32x4 acc[MT];
#pragma unroll 1
        for (unsigned n = 0;; ++n) {
            if (!wait_ge(w, w.seq_pub(), n + 1, pub_seen, 3)) return;
            sid = __builtin_amdgcn_readfirstlane(*w.fifo(n));
            if (sid < 0) return;
            ++g; slot = RingGeo<NT>::next(slot);
            if (!wait_ge(w, w.wr(L), g + 2, wr_seen)) return;
            pipe_fetch<NT>(w, fbase, slot, Bh, Bl);
            flag_store(w.rd(L), g);
#pragma unroll 1
            for (ti = 0; ti < w.ntseq - 1; ++ti) { // every tile but the last has a successor to fetch
                const unsigned ns = RingGeo<NT>::next(slot);
                if (!wait_ge(w, w.wr(L), g + 3, wr_seen)) return;
                pipe_mma_refill<MT, NT>(w, acc, bias, Ah, Al, Bh, Bl, fbase + ns * kPipeTileB);
                flag_store(w.rd(L), g + 1);
                (void)retire(acc, g, slot, ti);
                ++g; slot = ns;
            }
            pipe_mma<MT>(acc, bias, Ah, Al, Bh, Bl);
            (void)retire(acc, g, slot, ti);
            ++g; slot = RingGeo<NT>::next(slot);
        }
    }
}

// ---- stage 0: the front --------------------------------------------------------------------------
// One input group = 32 rows of (T, 24) fp32 = 192 float4 = 3 per lane (lane-linear, 1 KiB per instruction;
// rows past the sequence come back as zeros from the range check and are written as such).
struct PipeGroup { float4 v[3]; };

// hi / lo split of four values with packed converts and one mixed-precision FMA per value (see split8)
__device__ __forceinline__ void pipe_split4(const float4& v, f16x4& hi, f16x4& lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f16x2 h = f16x2{(_Float16)e[2 * i], (_Float16)e[2 * i + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(e[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(e[2 * i + 1]));
        const f16x2 l = f16x2{(_Float16)r0, (_Float16)r1};
        hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
        lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
    }
}

template <bool FUSED>
__device__ __forceinline__ void pipe_front(const PipeWave& w, const PipeArgs& a, int pid, int npipes) {
    constexpr int L = 0, MT = 2, NT0 = kPipeNT0, NT1 = kPipeNT;
    f16x8 Ah[MT][kTaps], Al[MT][kTaps];
    f32x4 bias[MT];
    pipe_weights<MT>(w, a.mp, L, Ah, Al, bias);
    const unsigned r0 = w.ring(0), r1 = w.ring(1);
    const unsigned fbase = r0 + (unsigned)(w.tcol * kPipePitch + w.q * 16);
    const unsigned wrow = (unsigned)((2 + w.tcol) * kPipePitch + w.q * 16);
    // commit: float4 u = lane + 64 jj of a group is channels 4 c4 .. +3 of its row u / 6 (0..31): tile row >> 4
    int crow[3];
    unsigned coff[3]; // (2 + (row & 15)) * pitch + c4 * 8
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int u = w.lane + 64 * jj, row = u / 6, c4 = u - row * 6;
        crow[jj] = row;
        coff[jj] = (unsigned)((2 + (row & 15)) * kPipePitch + c4 * 8);
    }
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
        flag_store(w.seq_pub(), ln);
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
    // -- commit: the oldest group in flight -> hi / lo rows of ring 0 (channels 24..31 of ring 0 stay the
    //    zeros the whole ring is initialised with) --
    unsigned cslot = 1; // slot of the first tile of the next group to commit
    int ck_ = 0;        // that group's index in its sequence
    auto commit = [&](const PipeGroup& G) {
        const unsigned sa = cslot, sb = (cslot + 1) & (NT0 - 1);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            f16x4 wh, wl;
            pipe_split4(G.v[jj], wh, wl);
            if (jj == 0) ring_put<NT0>(w, r0, sa, crow[jj], coff[jj], wh, wl);               // rows 0..10
            else if (jj == 2) ring_put<NT0>(w, r0, sb, crow[jj] & 15, coff[jj], wh, wl);     // rows 21..31
            else { // rows 10..21: both tiles; the select is per lane, the guard-row logic per slot
                const bool second = crow[jj] >= 16;
                typedef __attribute__((address_space(3))) f16x4 lds_v;
                const unsigned ad = r0 + (second ? sb : sa) * kPipeTileB + coff[jj];
                *(lds_v*)(w.lds + ad) = wh;
                *(lds_v*)(w.lds + ad + kPipeImg0) = wl;
                // guard copies: rows 14, 15 of a tile in the last slot (first tile of the pair: lanes with row 14, 15),
                // rows 0, 1 of a tile in slot 0 (second tile of the pair: lanes with row 16, 17)
                if (sa == (unsigned)(NT0 - 1)) {
                    if (crow[jj] == 14 || crow[jj] == 15) {
                        const unsigned gd = r0 + coff[jj] - 16 * kPipePitch;
                        *(lds_v*)(w.lds + gd) = wh;
                        *(lds_v*)(w.lds + gd + kPipeImg0) = wl;
                    }
                }
                if (sb == 0u) {
                    if (crow[jj] == 16 || crow[jj] == 17) {
                        const unsigned gd = r0 + coff[jj] + 16 * NT0 * kPipePitch;
                        *(lds_v*)(w.lds + gd) = wh;
                        *(lds_v*)(w.lds + gd + kPipeImg0) = wl;
                    }
                }
            }
        }
        ++ck_;
        cslot = (cslot + 2) & (NT0 - 1);
        if (ck_ == ngseq) { // the sequence is in; its zero tile (stream tile (tile 0) + ntseq) is written by the compute
            // cursor just before the last tile needs it -- here it could still hold tile ntseq - 4, which tile
            // ntseq - 3 has yet to read.  cslot stands at (tile 0) + 2 ngseq: one past the zero tile when ntseq is odd
            const unsigned zs = (w.ntseq & 1) ? ((cslot + NT0 - 1) & (NT0 - 1)) : cslot;
            cslot = (zs + 1) & (NT0 - 1);
            ck_ = 0;
        }
    };

    unsigned rd_seen = 0;
    // ring 0 starts as zeros (its channel padding is never written again); stream tile 0 of ring 1
    for (int o = w.lane * 16; o < kPipeRing0; o += 64 * 16) *(lds_f4*)(w.lds + r0 + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    zero_tile<NT1>(w, r1, 0u);
    flag_store(w.wr(1), 1u);
    announce_next();
    PipeGroup G[kPipeDepth];
#pragma unroll
    for (int d = 0; d < kPipeDepth; ++d) issue(G[d]);
    if (__builtin_amdgcn_readfirstlane(*w.fifo(0)) < 0) return;

    unsigned g = 1, n = 0;   // compute cursor: stream tile, sequence number
    unsigned s0 = 1, s1 = 1; // its slots in rings 0 and 1
    int i = 0;               // next tile of sequence n
    int kk = 0;              // groups of sequence n in ring 0
    bool done = false, failed = false;
    f16x8 Bh[kTaps], Bl[kTaps];
    f32x4 acc[MT];
    auto retire = [&](unsigned tg, int ti, unsigned ts1) -> bool { // acc of stream tile tg = tile ti -> ring 1
        const bool last = ti == w.ntseq - 1;
        const unsigned top = tg + (last ? 1u : 0u);
        if (top >= (unsigned)NT1 && !wait_ge(w, w.rd(1), top - NT1 + 1, rd_seen, 2)) return false;
        pipe_put_hidden<NT1>(w, r1, ts1, wrow, acc, last, ti);
        flag_store(w.wr(1), tg + 1);
        if (last) {
            zero_tile<NT1>(w, r1, RingGeo<NT1>::next(ts1));
            flag_store(w.wr(1), tg + 2);
        }
        return true;
    };
    // One step = one input group: the oldest group in flight goes into ring 0, its register slot is refilled
    // with the load kPipeDepth groups ahead, and the tiles that group completes are computed: after group kk
    // of a sequence the rows of tiles <= 2 kk + 1 are in, i.e. tiles <= 2 kk can run (a tile needs its right
    // neighbour), and after the last group all that are left.  The slots rotate through the registers by
    // NAME -- moving an in-flight load's destination would wait for it.
    auto step = [&](PipeGroup& slot) {
        commit(slot);
        if (i == w.ntseq - 1) zero_tile<NT0>(w, r0, (s0 + 1) & (NT0 - 1)); // the zero tile behind the sequence's last tile
        pipe_fetch<NT0>(w, fbase, s0, Bh, Bl); // tile i is complete now; its fragments travel while the next load is set up
        issue(slot);
        ++kk;
        const int upto = kk == ngseq ? w.ntseq - 1 : 2 * (kk - 1);
        while (true) {
            pipe_mma<MT>(acc, bias, Ah, Al, Bh, Bl);
            const unsigned tg = g, ts1 = s1;
            const int ti = i;
            ++i; ++g;
            s0 = (s0 + 1) & (NT0 - 1);
            s1 = RingGeo<NT1>::next(s1);
            const bool again = i <= upto;
            if (again) {
                if (i == w.ntseq - 1) zero_tile<NT0>(w, r0, (s0 + 1) & (NT0 - 1));
                pipe_fetch<NT0>(w, fbase, s0, Bh, Bl); // under the retiring tile's vector work
            }
            if (!retire(tg, ti, ts1)) { failed = true; return; }
            if (!again) break;
        }
        if (i == w.ntseq) { // the zero tile between sequences, then the next announced sequence
            ++g; ++n; i = 0; kk = 0;
            s0 = (s0 + 1) & (NT0 - 1);
            s1 = RingGeo<NT1>::next(s1);
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
    for (int i = threadIdx.x; i < 2 * kPipeCtlBytes / 4; i += 512) ((lds_u32*)(lds + 2 * kPipePerPipe))[i] = 0u;
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
    w.ring0 = (unsigned)(p * kPipePerPipe);
    w.ctl = (unsigned)(2 * kPipePerPipe + p * kPipeCtlBytes);
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
