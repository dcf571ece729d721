// libb2h.so -- C ABI (include/b2h.h) over the gfx950 kernels.
// Host side: argument checks mirroring the reference's errors, weight repacking
// into the kernels' fragment layouts, launches on the caller's stream.
#include "../../include/b2h.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "b2h_common.h"
#include "kernel_mfma.h"
#include "kernel_mfma16.h"
#include "kernel_mfma16w.h"
#include "kernel_mfma3.h"
#include "kernel_mfma3w.h"
#include "kernel_tenc.h"
#include "kernel_valu.h"

using namespace b2h;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(B2H_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

uint16_t f32_to_bf16(float f) { // round-to-nearest-even
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

uint16_t f32_to_f16(float f) { // round-to-nearest-even, IEEE binary16
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t absu = u & 0x7fffffffu;
    if (absu >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absu > 0x7f800000u) ? 0x200u : 0u));
    if (absu >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // rounds to >= 65520 -> inf
    if (absu < 0x38800000u) { // below 2^-14: subnormal half, grid 2^-24
        float a;
        std::memcpy(&a, &absu, 4);
        const uint32_t m = (uint32_t)std::nearbyintf(a * 16777216.0f);
        return (uint16_t)(sign | m);
    }
    uint32_t r = absu + 0xfffu + ((absu >> 13) & 1u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}

// hipEvent_t that is destroyed on every exit path
struct Event {
    hipEvent_t e = nullptr;
    ~Event() { if (e) (void)hipEventDestroy(e); }
};

// The library binds a model to the device current at creation; a launch from another current
// device would run a kernel there that dereferences this device's weights.
int check_device(int model_device) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != model_device)
        return fail(B2H_ERR_INVALID, "model is bound to HIP device " + std::to_string(model_device) +
                                        ", the current device is " + std::to_string(cur));
    return B2H_OK;
}

// The model-free entry points (metric, target transform) have no device of their own: every pointer
// must be non-NULL device memory of the CURRENT device, or the kernel would fault / run elsewhere.
int check_device_ptr(const void* p, const char* name) {
    if (!p) return fail(B2H_ERR_INVALID, std::string(name) + " is NULL");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(B2H_ERR_INVALID, std::string(name) + " is not a device pointer");
    }
    if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged)
        return fail(B2H_ERR_INVALID, std::string(name) + " is not device memory");
    if (a.device != cur)
        return fail(B2H_ERR_INVALID, std::string(name) + " lives on HIP device " + std::to_string(a.device) +
                                        ", the current device is " + std::to_string(cur));
    return B2H_OK;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const void* host, size_t n) {
        if (p && bytes != n) { (void)hipFree(p); p = nullptr; }
        if (!p) HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        HIP_TRY(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return B2H_OK;
    }
};

} // namespace

constexpr int kPoolSlots = 64;        // streams per model that get a chunk pool (Sched16); further streams run without
constexpr int kPoolSlotBytes = 128;   // one cache line per slot
constexpr int64_t kPoolMinChunks = 256; // chunks per workgroup from which a launch is dynamic (at 64 it measured 2.5 % slower than static)

struct b2h_model {
    int C = 0;
    int pos_emb = 0;
    int device = 0;
    bool has_weights = false;
    int cin[4], cout[4];
    // packed device weights
    DevBuf valu_w[4], valu_b[4];
    DevBuf mf32_w[4], m_bias[4];  // exact-fp32 MFMA kernel: per-layer fragments + bias fragments
    DevBuf m3_w[4];               // f16x3 kernel: per-layer hi / lo f16 fragments (bias shared)
    DevBuf mbf16_all, mf16_all;   // persistent 16-bit kernel: [W L0..L3 | bias L0..L3], kPacked16 bytes
    DevBuf mwbf_w[4], mwh_w[4], mw_bias[4]; // wide 16-bit kernel (33..64 channels): bf16 / f16 fragments, bias
    DevBuf mw3_w[4], mw32_w[4];             // wide f16x3 kernel: hi / lo f16 fragments; wide exact-fp32 kernel: fp32 fragments
    int num_cus = 256;
    // Chunk pools of the persistent 16-bit kernel (kernel_mfma16.h, Sched16): one 128-byte slot per stream
    // that has launched on this model, two words each (claim counter, finished workgroups), zero between
    // launches.  Launches on one stream are ordered, so a slot is never shared by two running kernels.
    DevBuf pools;
    std::mutex pool_mu;
    std::vector<hipStream_t> pool_streams;
    ValuParams vp;
    MfmaParams mp32, mp3, mpw_bf, mpw_h, mpw3, mpw32;
    float w_absmax = 0.f;         // largest |weight| or |bias| (NaN counts as inf): F16X3 needs < 65504
};

namespace {

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// value of weight (layer l, out-channel o, in-channel slot i, tap k); in-channel
// slot order of layer 1 with pos_emb: slots 0..23 = reference channels 1..24
// (keypoints), slot 24 = reference channel 0 (t/100).
struct HostWeights {
    const b2h_model* m;
    std::vector<float> w[4], b[4];
    float at(int l, int o, int slot, int k, bool permute_pos) const {
        if (o >= m->cout[l] || slot >= m->cin[l]) return 0.f;
        int i = slot;
        if (l == 0 && m->pos_emb && permute_pos) i = (slot == 24) ? 0 : slot + 1;
        return w[l][((size_t)o * m->cin[l] + i) * kTaps + k];
    }
    float bias(int l, int o) const { return o < m->cout[l] ? b[l][o] : 0.f; }
};

constexpr float kF16Max = 65504.f;

// largest magnitude of a set of fp32 values; a NaN makes it +inf
float absmax_of(const std::vector<float>& v, float acc) {
    for (float x : v) {
        const float a = std::fabs(x);
        if (!(a <= acc)) acc = std::isnan(a) ? INFINITY : a;
    }
    return acc;
}

// Wide 16-bit kernel (kernel_mfma16w.h): per layer [mt][tap][ks][lane][8], in-position 32ks + 8q + j
// (layer 1: the 24|25 inputs, pos_emb moved to slot 24; hidden layers: position = channel), out slot
// (mt, row) = channel 16(row>>2) + 4mt + (row&3) (head: 16mt + row); bias [mt][q][4] fp32.
int pack_wide(b2h_model* m, const HostWeights& hw) {
    for (int l = 0; l < 4; ++l) {
        const int MT = wide_mt(l), KS = wide_ks(l);
        auto chan = [&](int mt, int row) { return l == 3 ? last_chan_of(mt, row) : wide_chan_of(mt, row); };
        std::vector<uint16_t> wb((size_t)MT * kTaps * KS * 64 * 8), wh(wb.size());
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const float v = hw.at(l, chan(mt, lane & 15), 32 * ks + 8 * (lane >> 4) + j, k, true);
                            const size_t idx = ((((size_t)mt * kTaps + k) * KS + ks) * 64 + lane) * 8 + j;
                            wb[idx] = f32_to_bf16(v);
                            wh[idx] = f32_to_f16(v);
                        }
        // f16x3: [mt][tap][ks][hi|lo][lane][8], w = hi + lo with hi = f16(w), lo = f16(w - hi)
        std::vector<_Float16> w3((size_t)MT * kTaps * KS * 2 * 64 * 8);
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const float v = hw.at(l, chan(mt, lane & 15), 32 * ks + 8 * (lane >> 4) + j, k, true);
                            const _Float16 hi = (_Float16)v;
                            const size_t at = (((((size_t)mt * kTaps + k) * KS + ks) * 2) * 64 + lane) * 8 + j;
                            w3[at] = hi;
                            w3[at + 64 * 8] = (_Float16)(v - (float)hi);
                        }
        // exact fp32: [mt][tap][g][lane][4], in-position 16g + 4(lane>>4) + j, groups as Geo32<true>
        const int NG = Geo32<true>::groups(l);
        std::vector<float> wf((size_t)MT * kTaps * NG * 64 * 4);
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int g = 0; g < NG; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            wf[((((size_t)mt * kTaps + k) * NG + g) * 64 + lane) * 4 + j] =
                                hw.at(l, chan(mt, lane & 15), 16 * g + 4 * (lane >> 4) + j, k, true);
        std::vector<float> bf((size_t)MT * 16);
        for (int mt = 0; mt < MT; ++mt)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) bf[(mt * 4 + q) * 4 + r] = hw.bias(l, chan(mt, 4 * q + r));
        int rc;
        if ((rc = m->mw32_w[l].upload(wf.data(), wf.size() * 4))) return rc;
        m->mpw32.w[l] = m->mw32_w[l].p;
        if ((rc = m->mw3_w[l].upload(w3.data(), w3.size() * 2))) return rc;
        m->mpw3.w[l] = m->mw3_w[l].p;
        if ((rc = m->mwbf_w[l].upload(wb.data(), wb.size() * 2))) return rc;
        if ((rc = m->mwh_w[l].upload(wh.data(), wh.size() * 2))) return rc;
        if ((rc = m->mw_bias[l].upload(bf.data(), bf.size() * 4))) return rc;
        m->mpw_bf.w[l] = m->mwbf_w[l].p;
        m->mpw_h.w[l] = m->mwh_w[l].p;
        m->mpw_bf.bias[l] = m->mpw_h.bias[l] = m->mpw3.bias[l] = m->mpw32.bias[l] = (const float*)m->mw_bias[l].p;
    }
    m->mpw_bf.pos_emb = m->mpw_h.pos_emb = m->mpw3.pos_emb = m->mpw32.pos_emb = m->pos_emb;
    return B2H_OK;
}

// single-parameter aliases of the wide kernel for the launch macro
template <bool FUSED> constexpr auto b2h_fwd_mfma16w_bf16 = b2h_fwd_mfma16w<PREC_BF16, FUSED>;
template <bool FUSED> constexpr auto b2h_fwd_mfma16w_f16 = b2h_fwd_mfma16w<PREC_F16, FUSED>;
template <bool FUSED> constexpr auto b2h_fwd_mfma_f32_narrow = b2h_fwd_mfma_f32<FUSED, false>;
template <bool FUSED> constexpr auto b2h_fwd_mfma_f32_wide = b2h_fwd_mfma_f32<FUSED, true>;

int pack_all(b2h_model* m, const HostWeights& hw) {
    m->w_absmax = 0.f;
    for (int l = 0; l < 4; ++l) m->w_absmax = absmax_of(hw.b[l], absmax_of(hw.w[l], m->w_absmax));
    // ---- VALU layout: w[k][i][opad], reference channel order
    for (int l = 0; l < 4; ++l) {
        const int opad = round_up(m->cout[l], 8), cin = m->cin[l];
        std::vector<float> w((size_t)kTaps * cin * opad, 0.f), b(opad, 0.f);
        for (int k = 0; k < kTaps; ++k)
            for (int i = 0; i < cin; ++i)
                for (int o = 0; o < m->cout[l]; ++o)
                    w[((size_t)k * cin + i) * opad + o] = hw.at(l, o, i, k, false);
        for (int o = 0; o < m->cout[l]; ++o) b[o] = hw.b[l][o];
        int rc = m->valu_w[l].upload(w.data(), w.size() * 4);
        if (rc) return rc;
        rc = m->valu_b[l].upload(b.data(), b.size() * 4);
        if (rc) return rc;
        m->vp.L[l] = ValuLayer{(const float*)m->valu_w[l].p, (const float*)m->valu_b[l].p, cin,
                               m->cout[l], opad};
    }
    {
        int as = round_up(m->C, 8);
        if (as < m->cin[0]) as = m->cin[0];
        m->vp.act_stride = as | 1;
        int wb = 0;
        for (int l = 0; l < 4; ++l) wb = std::max(wb, m->vp.L[l].cin * m->vp.L[l].opad); // one tap at a time
        m->vp.wbuf_floats = wb;
        m->vp.pos_emb = m->pos_emb;
    }
    if (m->C > kMfmaWideWidth) return B2H_OK;       // 65..128 channels: the VALU kernel only
    if (m->C > kMfmaWidth) return pack_wide(m, hw); // 33..64 channels: the wide matrix-core kernels

    // ---- MFMA layouts
    std::vector<unsigned char> ab(kPacked16, 0), ah(kPacked16, 0); // LDS images of the persistent kernel
    for (int l = 0; l < 4; ++l) {
        const int MT = (l == 3) ? 3 : 2;
        auto chan = [&](int mt, int row) { return l == 3 ? last_chan_of(mt, row) : hidden_chan_of(mt, row); };
        // 16-bit: [mt][tap][lane][8]
        uint16_t* wb = reinterpret_cast<uint16_t*>(ab.data() + kWLayerOff16[l]);
        uint16_t* wh = reinterpret_cast<uint16_t*>(ah.data() + kWLayerOff16[l]);
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = hw.at(l, chan(mt, lane & 15), 8 * (lane >> 4) + j, k, true);
                        const size_t idx = (((size_t)mt * kTaps + k) * 64 + lane) * 8 + j;
                        wb[idx] = f32_to_bf16(v);
                        wh[idx] = f32_to_f16(v);
                    }
        // fp32: [mt][tap][g][lane][4]
        std::vector<float> wf((size_t)MT * kTaps * 2 * 64 * 4);
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int g = 0; g < 2; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            wf[((((size_t)mt * kTaps + k) * 2 + g) * 64 + lane) * 4 + j] =
                                hw.at(l, chan(mt, lane & 15), 16 * g + 4 * (lane >> 4) + j, k, true);
        // bias: [mt][q][4]
        std::vector<float> bf((size_t)MT * 16);
        for (int mt = 0; mt < MT; ++mt)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) bf[(mt * 4 + q) * 4 + r] = hw.bias(l, chan(mt, 4 * q + r));
        std::memcpy(ab.data() + kBiasOff16[l], bf.data(), bf.size() * 4);
        std::memcpy(ah.data() + kBiasOff16[l], bf.data(), bf.size() * 4);
        // f16x3: [mt][tap][hi|lo][lane][8] halves, w = hi + lo with hi = f16(w), lo = f16(w - hi)
        std::vector<_Float16> w3((size_t)MT * kTaps * 2 * 64 * 8);
        for (int mt = 0; mt < MT; ++mt)
            for (int k = 0; k < kTaps; ++k)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = hw.at(l, chan(mt, lane & 15), 8 * (lane >> 4) + j, k, true);
                        const _Float16 hi = (_Float16)v;
                        const size_t at = ((((size_t)mt * kTaps + k) * 2) * 64 + lane) * 8 + j;
                        w3[at] = hi;
                        w3[at + 64 * 8] = (_Float16)(v - (float)hi);
                    }
        int rc;
        if ((rc = m->mf32_w[l].upload(wf.data(), wf.size() * 4))) return rc;
        if ((rc = m->m_bias[l].upload(bf.data(), bf.size() * 4))) return rc;
        if ((rc = m->m3_w[l].upload(w3.data(), w3.size() * 2))) return rc;
        m->mp32.w[l] = m->mf32_w[l].p;
        m->mp32.bias[l] = (const float*)m->m_bias[l].p;
        m->mp3.w[l] = m->m3_w[l].p;
        m->mp3.bias[l] = (const float*)m->m_bias[l].p;
    }
    m->mp32.pos_emb = m->pos_emb;
    m->mp3.pos_emb = m->pos_emb;
    int rc;
    if ((rc = m->mbf16_all.upload(ab.data(), ab.size()))) return rc;
    if ((rc = m->mf16_all.upload(ah.data(), ah.size()))) return rc;
    if (!m->pools.p) { // once per model: a later weight replacement must not touch the words of a running launch
        const std::vector<char> zeros((size_t)kPoolSlots * kPoolSlotBytes, 0);
        if ((rc = m->pools.upload(zeros.data(), zeros.size()))) return rc;
    }
    return B2H_OK;
}

int resolve_kernel(const b2h_model* m, int kernel) {
    // AUTO = the faster of the two exact-fp32 kernels (profiles/r2_bf16/widths_8192x200.txt): the
    // matrix-core kernel costs the same at every width of its geometry (<= 32: 2.66 G frames/s,
    // 33..64: 0.77 G), the VALU kernel's cost grows with the width and is ahead only just above the
    // geometry step (0.88 G at 33 channels, level at 40) and for the narrowest models (3.08 G at 8
    // channels, 2.51 G at 10)
    if (kernel == B2H_KERNEL_AUTO)
        return (m->C <= 8 || (m->C > kMfmaWidth && m->C < 40) || m->C > kMfmaWideWidth) ? B2H_KERNEL_F32_VALU
                                                                                          : B2H_KERNEL_F32_MFMA;
    return kernel;
}

bool kernel_ok(const b2h_model* m, int k) {
    switch (k) {
        case B2H_KERNEL_F32_VALU: return m->C <= kMaxWidth;
        case B2H_KERNEL_F32_MFMA:
        case B2H_KERNEL_BF16_MFMA:
        case B2H_KERNEL_F16_MFMA: return m->C <= kMfmaWideWidth; // > 32 channels: the wide kernels
        case B2H_KERNEL_F16X3_MFMA: return m->C <= kMfmaWideWidth && (!m->has_weights || m->w_absmax < kF16Max);
        default: return false;
    }
}

// Every kernel that may use more than 64 KB of dynamic LDS gets its cap raised ONCE per device, when
// weights are loaded -- not inside launch(), so that the very first b2h_forward is already free of
// runtime calls other than the launch itself and can be captured into a HIP graph.
template <typename K> int raise_lds_cap(K kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return B2H_OK;
}

int set_conv_kernel_attributes() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && done[dev]) return B2H_OK;
    int rc;
    if ((rc = raise_lds_cap(b2h_fwd_f32_valu<true>)) || (rc = raise_lds_cap(b2h_fwd_f32_valu<false>)) ||
        (rc = raise_lds_cap(b2h_fwd_f32_valu<true, 3>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma_f32<false, false>)) || (rc = raise_lds_cap(b2h_fwd_mfma_f32<true, false>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma_f32<false, true>)) || (rc = raise_lds_cap(b2h_fwd_mfma_f32<true, true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma_f16x3<false>)) || (rc = raise_lds_cap(b2h_fwd_mfma_f16x3<true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma_f16x3w<false>)) || (rc = raise_lds_cap(b2h_fwd_mfma_f16x3w<true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_BF16, false, false>)) || (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_BF16, true, false>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_F16, false, false>)) || (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_F16, true, false>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_BF16, false, true>)) || (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_BF16, true, true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_F16, false, true>)) || (rc = raise_lds_cap(b2h_fwd_mfma16<PREC_F16, true, true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16w<PREC_BF16, false>)) || (rc = raise_lds_cap(b2h_fwd_mfma16w<PREC_BF16, true>)) ||
        (rc = raise_lds_cap(b2h_fwd_mfma16w<PREC_F16, false>)) || (rc = raise_lds_cap(b2h_fwd_mfma16w<PREC_F16, true>)))
        return rc;
    if (dev >= 0 && dev < 64) done[dev] = true;
    return B2H_OK;
}

int launch(b2h_model* m, const float* x, float* y, int64_t B, int64_t T, int kernel,
           const FusedArgs& fa, hipStream_t st) {
    if (!m) return fail(B2H_ERR_INVALID, "model is NULL");
    if (!m->has_weights) return fail(B2H_ERR_NO_WEIGHTS, "b2h_forward before b2h_load_weights");
    if (B < 0 || T < 1) return fail(B2H_ERR_SHAPE, "expected B >= 0 and T >= 1");
    if (T > (1 << 24)) return fail(B2H_ERR_SHAPE, "T too large");
    if (m->pos_emb && T != 100)
        return fail(B2H_ERR_SHAPE, "pos_emb model requires T == 100 (LinearPositionalEmbedding max_len, "
                                   "HandPoseModels.py:23,78-84)");
    if (B == 0) return B2H_OK;
    if (!x || !y) return fail(B2H_ERR_INVALID, "x / y is NULL");
    if (int rc = check_device(m->device)) return rc;
    // 16-B vector loads of x rows (96 B each) and 8-B granular stores of y rows (168 B each)
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return fail(B2H_ERR_INVALID, "x and y must be 16-byte aligned (hipMalloc / torch allocations are)");
    {
        const char* xb = reinterpret_cast<const char*>(x);
        const char* yb = reinterpret_cast<const char*>(y);
        const size_t xn = (size_t)B * T * kInCh * 4, yn = (size_t)B * T * kOutCh * 4;
        if (xb < yb + yn && yb < xb + xn) return fail(B2H_ERR_INVALID, "x and y overlap");
    }
    if ((fa.flags & kPostMask) && !fa.n_frames)
        return fail(B2H_ERR_INVALID, "B2H_POST_MASK_TAIL needs n_frames");
    const int k = resolve_kernel(m, kernel);
    if (!kernel_ok(m, k)) {
        if (k == B2H_KERNEL_F16X3_MFMA && m->C <= kMfmaWideWidth)
            return fail(B2H_ERR_UNSUPPORTED, "F16X3 kernel: a weight or bias is outside the f16 range (|w| >= 65504 or "
                                             "not finite); use the exact fp32 kernel");
        return fail(B2H_ERR_UNSUPPORTED, "kernel variant does not support conv_channels=" + std::to_string(m->C) +
                                             " (matrix-core kernels: <= 64; exact fp32 VALU kernel: <= 128)");
    }

    if (k == B2H_KERNEL_F32_VALU) {
        const int tiles = (int)((T + kValuTile - 1) / kValuTile);
        const int64_t grid = B * tiles;
        if (grid > 0x7fffffff) return fail(B2H_ERR_SHAPE, "B*T too large for one launch");
        const size_t lds = ((size_t)2 * kValuRows * m->vp.act_stride + m->vp.wbuf_floats) * 4;
        if (m->C > 104) { // three work items per thread (kernel_valu.h)
            hipLaunchKernelGGL((b2h_fwd_f32_valu<true, 3>), dim3((unsigned)grid), dim3(256), lds, st, x, y, (int)T, tiles, m->vp, fa);
        } else if (m->C > 56) {
            hipLaunchKernelGGL(b2h_fwd_f32_valu<true>, dim3((unsigned)grid), dim3(256), lds, st, x, y, (int)T, tiles, m->vp, fa);
        } else {
            hipLaunchKernelGGL(b2h_fwd_f32_valu<false>, dim3((unsigned)grid), dim3(256), lds, st, x, y, (int)T, tiles, m->vp, fa);
        }
    } else {
        // chunk length of the wave-per-chunk kernels: 112 frames (the LDS image's capacity) unless
        // that leaves most of the chip's wave slots idle (2 workgroups x 4 waves per CU); then 64 or
        // 32 frames, paying the +-8-frame halo recompute for parallelism.  Any chunking computes
        // bit-identical frames.
        const bool wide = m->C > kMfmaWidth; // 33..64 channels: wave-per-chunk 16-bit kernel (kernel_mfma16w.h)
        int chunk_len = kChunk;
        const bool wide3 = wide && (k == B2H_KERNEL_F16X3_MFMA || k == B2H_KERNEL_F32_MFMA); // one 4-wave workgroup per CU
        if (wide || (k != B2H_KERNEL_BF16_MFMA && k != B2H_KERNEL_F16_MFMA)) {
            const int64_t slots = (int64_t)m->num_cus * (wide3 ? 1 : 2) * kWavesPerBlock;
            for (int cand : {64, 32}) {
                if (B * ((T + chunk_len - 1) / chunk_len) * 2 >= slots) break;
                chunk_len = cand;
            }
        }
        const int cps = (int)((T + chunk_len - 1) / chunk_len);
        // (equal chunks -- T = 200 as 100 + 100 instead of 112 + 88 -- measured 3.9 % SLOWER in f16x3 and 1.6 %
        // in exact fp32, same-process A/B, profiles/r3_f16x3/ab_equal_chunks.txt: not done)
        const int64_t nchunks = B * cps;
        const int64_t grid = (nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
        if (grid > 0x7fffffff) return fail(B2H_ERR_SHAPE, "B*T too large for one launch");
        const dim3 g((unsigned)grid), blk(64 * kWavesPerBlock);
        const bool fusedc = fa.flags != 0; // the plain instantiations carry no transform code at all
#define B2H_LAUNCHC(KERN, LDS, MP)                                                                          \
    do {                                                                                                    \
        if (fusedc) hipLaunchKernelGGL((KERN<true>), g, blk, LDS, st, x, y, (int)T, cps, chunk_len, nchunks, MP, fa);  \
        else hipLaunchKernelGGL((KERN<false>), g, blk, LDS, st, x, y, (int)T, cps, chunk_len, nchunks, MP, fa);        \
    } while (0)
        if (wide3 && k == B2H_KERNEL_F16X3_MFMA) {
            const size_t lds = (size_t)kWavesPerBlock * 2 * kImg3W;
            B2H_LAUNCHC(b2h_fwd_mfma_f16x3w, lds, m->mpw3);
        } else if (wide3) {
            const size_t lds = (size_t)kWavesPerBlock * Geo32<true>::kImg;
            B2H_LAUNCHC(b2h_fwd_mfma_f32_wide, lds, m->mpw32);
        } else if (wide) {
            const size_t lds = (size_t)kWavesPerBlock * kImgW;
            if (k == B2H_KERNEL_BF16_MFMA) B2H_LAUNCHC(b2h_fwd_mfma16w_bf16, lds, m->mpw_bf);
            else B2H_LAUNCHC(b2h_fwd_mfma16w_f16, lds, m->mpw_h);
        } else if (k == B2H_KERNEL_F32_MFMA) {
            const size_t lds = (size_t)kWavesPerBlock * Geo32<false>::kImg;
            B2H_LAUNCHC(b2h_fwd_mfma_f32_narrow, lds, m->mp32);
        } else if (k == B2H_KERNEL_F16X3_MFMA) {
            const size_t lds = (size_t)kWavesPerBlock * 2 * kImg3; // hi + lo images = the fp32 image's bytes
            B2H_LAUNCHC(b2h_fwd_mfma_f16x3, lds, m->mp3);
        } else {
            // persistent kernel: one 512-thread workgroup per CU
            // Chunk length: whole sequences (<= 208 frames) or 192-frame chunks keep the halo
            // recompute at zero / 8 %.  When that leaves most of the chip's 2048 wave slots idle
            // (small batches) shorter chunks trade halo work for parallelism; every chunking
            // computes bit-identical frames.
            int TT = (T <= kChunkWhole16) ? kChunkWhole16 : kChunkSplit16;
            int64_t nch = B * ((T + TT - 1) / TT);
            const int64_t slots = (int64_t)m->num_cus * kWaves16;
            for (int cand : {96, 48}) {
                if (nch * 2 >= slots || T <= cand) break;
                TT = cand;
                nch = B * ((T + TT - 1) / TT);
            }
            const int cps16 = (int)((T + TT - 1) / TT);
            const unsigned grid16 = (unsigned)std::min<int64_t>(m->num_cus, nch);
            if (nch >= 0x7fffffff) return fail(B2H_ERR_SHAPE, "B*T too large for one launch");
            // Work distribution (Sched16): with >= 256 chunks per workgroup the launch is DYNAMIC -- waves claim runs
            // of two consecutive chunks from a device-wide counter, so the chip walks through x and y as one front
            // (kernel_mfma16.h).  The counter lives in this stream's slot; no slot (more than kPoolSlots streams)
            // or a stream under capture (a graph may be replayed on any stream, concurrently with this one) means a
            // STATIC launch.
            Sched16 sched{nullptr, 2};
            const int64_t per_wg = nch / grid16;
            if (per_wg >= kPoolMinChunks && m->pools.p) {
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
                if (cap == hipStreamCaptureStatusNone) {
                    std::lock_guard<std::mutex> lock(m->pool_mu);
                    size_t slot = 0;
                    while (slot < m->pool_streams.size() && m->pool_streams[slot] != st) ++slot;
                    if (slot == m->pool_streams.size() && slot < (size_t)kPoolSlots) m->pool_streams.push_back(st);
                    if (slot < (size_t)kPoolSlots)
                        sched.pool = reinterpret_cast<unsigned*>(static_cast<char*>(m->pools.p) + slot * kPoolSlotBytes);
                }
            }
            const bool fused = fa.flags != 0;
            const bool bf = (k == B2H_KERNEL_BF16_MFMA);
            const void* wp = bf ? m->mbf16_all.p : m->mf16_all.p;
            // streaming cache policy (kernel_mfma.h: kLdStream / kStStream) from 1 MiB of traffic up
            const bool stream = B * T * (int64_t)((kInCh + kOutCh) * 4) >= (1 << 20);
#define B2H_LAUNCH16(PR, FU, SM)                                                                          \
    hipLaunchKernelGGL((b2h_fwd_mfma16<PR, FU, SM>), dim3(grid16), dim3(64 * kWaves16), kLdsAlloc16, st, x, y, \
                       (int)T, cps16, TT, nch, wp, m->pos_emb, fa, sched)
#define B2H_LAUNCH16S(PR, FU) do { if (stream) B2H_LAUNCH16(PR, FU, true); else B2H_LAUNCH16(PR, FU, false); } while (0)
            if (bf && !fused) B2H_LAUNCH16S(PREC_BF16, false);
            else if (bf) B2H_LAUNCH16S(PREC_BF16, true);
            else if (!fused) B2H_LAUNCH16S(PREC_F16, false);
            else B2H_LAUNCH16S(PREC_F16, true);
#undef B2H_LAUNCH16S
#undef B2H_LAUNCH16
        }
#undef B2H_LAUNCHC
    }
    HIP_TRY(hipGetLastError());
    return B2H_OK;
}

} // namespace

// ---- TransformerEnc ---------------------------------------------------------------------
// One stage blob of the chain kernel: weight fragments of <= 128 outputs + bias/gamma/beta.
struct TencBlob {
    DevBuf buf;   // fp32 fragments (k-groups of 16)
    DevBuf buf16; // f16 hi + lo fragments (k-groups of 32), B2H_TENC_F16X3
    int mtiles = 0, kgroups = 0, kgroups32 = 0, nout = 0;
};

struct b2h_tenc {
    int nlayers = 0, max_len = 0, device = 0;
    bool has_weights = false;
    int kernel = B2H_TENC_F32;
    float w_absmax = 0.f; // largest |parameter| (NaN counts as inf): B2H_TENC_F16X3 needs < 65504
    DevBuf pe;
    TencBlob in_proj, out_proj;
    struct Layer {
        TencBlob q, k, v, attn_out, ff1, ff2;
        TencBlob qkv_head[kTencHeads]; // rows of Q_h, K_h, V_h of in_proj_weight: the projection inside b2h_attn_qkv_h3
    };
    int num_cus = 256;
    std::vector<Layer> layers;
};

namespace {

// rows [r0, r0 + nout) of W (*, k) row-major fp32 -> [mt][g][lane][4] with
// W[r0 + 16mt + (lane&15)][16g + 4(lane>>4) + j], followed by bias, gamma, beta (128 each)
int pack_blob(TencBlob& B, const float* w, const float* b, int r0, int nout, int k, const float* gamma,
              const float* beta) {
    B.kgroups = (k + 15) / 16;
    B.mtiles = (nout + 15) / 16;
    B.nout = nout;
    const size_t nw = (size_t)B.mtiles * B.kgroups * 64 * 4;
    std::vector<float> blob(nw + kStageParams, 0.f);
    for (int mt = 0; mt < B.mtiles; ++mt)
        for (int g = 0; g < B.kgroups; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int o = 16 * mt + (lane & 15), kk = 16 * g + 4 * (lane >> 4) + j;
                    if (o < nout && kk < k)
                        blob[(((size_t)mt * B.kgroups + g) * 64 + lane) * 4 + j] = w[(size_t)(r0 + o) * k + kk];
                }
    for (int o = 0; o < nout; ++o) blob[nw + o] = b[r0 + o];
    if (gamma) std::memcpy(blob.data() + nw + kTencD, gamma, kTencD * 4);
    if (beta) std::memcpy(blob.data() + nw + 2 * kTencD, beta, kTencD * 4);
    int rc = B.buf.upload(blob.data(), blob.size() * 4);
    if (rc) return rc;
    // f16 hi/lo fragments for v_mfma_f32_16x16x32_f16: [part][mt][g][lane][8] with k-slot (g, q, j)
    // = input feature 32g + 16(j>>2) + 4q + (j&3) (kernel_tenc.h), then the same fp32 parameters
    B.kgroups32 = (k + 31) / 32;
    const size_t nh = (size_t)B.mtiles * B.kgroups32 * 64 * 8; // halves per part
    std::vector<_Float16> frag(2 * nh, (_Float16)0.f);
    for (int mt = 0; mt < B.mtiles; ++mt)
        for (int g = 0; g < B.kgroups32; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = 16 * mt + (lane & 15), kk = 32 * g + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                    if (o < nout && kk < k) {
                        const float x = w[(size_t)(r0 + o) * k + kk];
                        const _Float16 hi = (_Float16)x;
                        const size_t at = (((size_t)mt * B.kgroups32 + g) * 64 + lane) * 8 + j;
                        frag[at] = hi;
                        frag[nh + at] = (_Float16)(x - (float)hi);
                    }
                }
    std::vector<char> blob16(2 * nh * 2 + kStageParams * 4);
    std::memcpy(blob16.data(), frag.data(), 2 * nh * 2);
    std::memcpy(blob16.data() + 2 * nh * 2, blob.data() + nw, kStageParams * 4);
    return B.buf16.upload(blob16.data(), blob16.size());
}

ChainStage stage_of(const b2h_tenc* m, const TencBlob& B, int type, float* out, int ldo) {
    if (m->kernel == B2H_TENC_F16X3)
        return ChainStage{(const float*)B.buf16.p, out, type, B.mtiles, B.kgroups32, ldo, B.nout, 2 * B.mtiles * B.kgroups32 * 64};
    return ChainStage{(const float*)B.buf.p, out, type, B.mtiles, B.kgroups, ldo, B.nout, B.mtiles * B.kgroups * 64};
}

int launch_chain(b2h_tenc* m, ChainArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * kStageBlobMax * sizeof(float);
    // persistent: one workgroup per CU (134 KB of LDS each) walks over the 128-frame blocks
    const int64_t blocks = std::min<int64_t>((a.n + 16 * kLinWaves - 1) / (16 * kLinWaves), m->num_cus);
    if (m->kernel == B2H_TENC_F16X3)
        hipLaunchKernelGGL(b2h_tenc_chain<true>, dim3((unsigned)blocks), dim3(64 * kLinWaves), lds, st, a);
    else
        hipLaunchKernelGGL(b2h_tenc_chain<false>, dim3((unsigned)blocks), dim3(64 * kLinWaves), lds, st, a);
    return B2H_OK;
}

} // namespace

extern "C" {

int b2h_tenc_create(int ninp, int nhead, int nhid, int nout, int nlayers, int max_len, b2h_tenc** out) {
    if (!out) return fail(B2H_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (ninp != kInCh || nhead != kTencHeads || nhid != kTencD || nout != kOutCh)
        return fail(B2H_ERR_UNSUPPORTED, "TransformerEnc: only ninp=24, nhead=4, nhid=128, nout=42 "
                                         "(infer_utterance.py:99-101) is implemented");
    if (nlayers < 1 || nlayers > 16 || max_len < 1 || max_len > 128)
        return fail(B2H_ERR_UNSUPPORTED, "TransformerEnc: 1 <= nlayers <= 16 and 1 <= max_len <= 128");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(B2H_ERR_NO_DEVICE, "no HIP device visible (libb2h has no CPU path)");
    std::unique_ptr<b2h_tenc> m(new b2h_tenc()); // released to the caller only on success
    HIP_TRY(hipGetDevice(&m->device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, m->device));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(B2H_ERR_NO_DEVICE, std::string("device is ") + p.gcnArchName + ", libb2h is built for gfx950 only");
    m->nlayers = nlayers;
    m->max_len = max_len;
    m->num_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    m->layers.resize(nlayers);
    *out = m.release();
    return B2H_OK;
}

int b2h_tenc_destroy(b2h_tenc* m) {
    delete m;
    return B2H_OK;
}

int b2h_tenc_set_kernel(b2h_tenc* m, int kernel) {
    if (!m) return fail(B2H_ERR_INVALID, "model is NULL");
    if (kernel != B2H_TENC_F32 && kernel != B2H_TENC_F16X3) return fail(B2H_ERR_INVALID, "unknown TransformerEnc kernel");
    m->kernel = kernel;
    return B2H_OK;
}

int b2h_tenc_load_weights(b2h_tenc* m, const float* const* tensors, int count, int on_device) {
    if (!m || !tensors) return fail(B2H_ERR_INVALID, "NULL argument");
    if (count != 5 + 12 * m->nlayers) return fail(B2H_ERR_INVALID, "expected 5 + 12*nlayers tensors");
    const int D = kTencD;
    std::vector<size_t> sizes = {(size_t)m->max_len * kInCh, (size_t)D * kInCh, (size_t)D};
    for (int l = 0; l < m->nlayers; ++l)
        for (size_t s : {(size_t)3 * D * D, (size_t)3 * D, (size_t)D * D, (size_t)D, (size_t)D * D, (size_t)D,
                         (size_t)D * D, (size_t)D, (size_t)D, (size_t)D, (size_t)D, (size_t)D})
            sizes.push_back(s);
    sizes.push_back((size_t)kOutCh * D);
    sizes.push_back((size_t)kOutCh);
    std::vector<std::vector<float>> h(count);
    for (int i = 0; i < count; ++i) {
        if (!tensors[i]) return fail(B2H_ERR_INVALID, "tensor pointer is NULL");
        h[i].resize(sizes[i]);
        if (on_device) HIP_TRY(hipMemcpy(h[i].data(), tensors[i], sizes[i] * 4, hipMemcpyDeviceToHost));
        else std::memcpy(h[i].data(), tensors[i], sizes[i] * 4);
    }
    HIP_TRY(hipDeviceSynchronize());
    if (int rc0 = check_device(m->device)) return rc0;
    // LDS caps are raised here, not in b2h_tenc_forward: the first forward is already capture-safe
    constexpr int kChainLds = 2 * kStageBlobMax * (int)sizeof(float);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(b2h_tenc_chain<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(b2h_tenc_chain<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds));
#define B2H_AQ_CAP(N) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(b2h_attn_qkv_h3<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    B2H_AQ_CAP(1) B2H_AQ_CAP(2) B2H_AQ_CAP(3) B2H_AQ_CAP(4) B2H_AQ_CAP(5) B2H_AQ_CAP(6) B2H_AQ_CAP(7) B2H_AQ_CAP(8)
#undef B2H_AQ_CAP
    m->w_absmax = 0.f;
    for (int i = 1; i < count; ++i) m->w_absmax = absmax_of(h[i], m->w_absmax); // h[0] is the pe table (|pe| <= 1)
    int rc;
    if ((rc = m->pe.upload(h[0].data(), h[0].size() * 4))) return rc;
    if ((rc = pack_blob(m->in_proj, h[1].data(), h[2].data(), 0, D, kInCh, nullptr, nullptr))) return rc;
    for (int l = 0; l < m->nlayers; ++l) {
        auto& L = m->layers[l];
        const int o = 3 + 12 * l;
        if ((rc = pack_blob(L.q, h[o].data(), h[o + 1].data(), 0, D, D, nullptr, nullptr))) return rc;
        if ((rc = pack_blob(L.k, h[o].data(), h[o + 1].data(), D, D, D, nullptr, nullptr))) return rc;
        if ((rc = pack_blob(L.v, h[o].data(), h[o + 1].data(), 2 * D, D, D, nullptr, nullptr))) return rc;
        for (int hd = 0; hd < kTencHeads; ++hd) { // [Q_h | K_h | V_h]: rows 32 hd .. of each third of in_proj_weight / bias
            std::vector<float> wh((size_t)3 * kTencHd * D), bhd((size_t)3 * kTencHd);
            for (int part = 0; part < 3; ++part)
                for (int r = 0; r < kTencHd; ++r) {
                    std::memcpy(&wh[((size_t)part * kTencHd + r) * D], &h[o][((size_t)part * D + hd * kTencHd + r) * D], D * 4);
                    bhd[part * kTencHd + r] = h[o + 1][part * D + hd * kTencHd + r];
                }
            if ((rc = pack_blob(L.qkv_head[hd], wh.data(), bhd.data(), 0, 3 * kTencHd, D, nullptr, nullptr))) return rc;
        }
        if ((rc = pack_blob(L.attn_out, h[o + 2].data(), h[o + 3].data(), 0, D, D, h[o + 8].data(), h[o + 9].data()))) return rc;
        if ((rc = pack_blob(L.ff1, h[o + 4].data(), h[o + 5].data(), 0, D, D, nullptr, nullptr))) return rc;
        if ((rc = pack_blob(L.ff2, h[o + 6].data(), h[o + 7].data(), 0, D, D, h[o + 10].data(), h[o + 11].data()))) return rc;
    }
    const int o = 3 + 12 * m->nlayers;
    if ((rc = pack_blob(m->out_proj, h[o].data(), h[o + 1].data(), 0, kOutCh, D, nullptr, nullptr))) return rc;
    m->has_weights = true;
    return B2H_OK;
}

size_t b2h_tenc_workspace_bytes(const b2h_tenc* m, int64_t B, int64_t T) {
    if (!m || B < 0 || T < 0) return 0;
    return (size_t)B * T * (5 * kTencD) * sizeof(float); // residual stream XA, attention output OC (128 each) + QKV (384)
}

} // extern "C"

namespace {
int tenc_launch(b2h_tenc* m, const float* x, float* y, int64_t B, int64_t T, const FusedArgs& fa, void* workspace,
                size_t workspace_bytes, void* stream) {
    if (!m) return fail(B2H_ERR_INVALID, "model is NULL");
    if (!m->has_weights) return fail(B2H_ERR_NO_WEIGHTS, "b2h_tenc_forward before b2h_tenc_load_weights");
    if (B < 0 || T < 1) return fail(B2H_ERR_SHAPE, "expected B >= 0 and T >= 1");
    if (T > m->max_len)
        return fail(B2H_ERR_SHAPE, "TransformerEnc: T exceeds the positional encoding's max_len (src + pe[:T], "
                                   "HandPoseModels.py:101,167)");
    if (B == 0) return B2H_OK;
    const int64_t n = B * T;
    // grid limits: attention launches B x heads workgroups, the chain n / 128
    if (B * kTencHeads > 0x7fffffff || n / (16 * kLinWaves) >= 0x7fffffff) return fail(B2H_ERR_SHAPE, "batch too large for one launch");
    if (!x || !y || !workspace) return fail(B2H_ERR_INVALID, "NULL pointer");
    if (int rc = check_device(m->device)) return rc;
    if (m->kernel == B2H_TENC_F16X3 && !(m->w_absmax < kF16Max))
        return fail(B2H_ERR_UNSUPPORTED, "B2H_TENC_F16X3: a parameter is outside the f16 range (|w| >= 65504 or not "
                                         "finite); use B2H_TENC_F32");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return fail(B2H_ERR_INVALID, "x and workspace must be 16-byte aligned");
    if (workspace_bytes < b2h_tenc_workspace_bytes(m, B, T)) return fail(B2H_ERR_INVALID, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* XA = reinterpret_cast<float*>(workspace);
    float* OC = XA + n * kTencD;
    float* QKV = OC + n * kTencD;
    int rc;
    const bool h3 = m->kernel == B2H_TENC_F16X3;
    if (h3) {
        // Round 3: the Q, K, V projection runs inside the attention kernel (b2h_attn_qkv_h3), so only the residual
        // stream and the attention output cross HBM between launches (2.5 KB per frame and layer -> 1.25 KB).
        {   // src + pe -> pose2hidden_projection (HandPoseModels.py:167-169) -> residual stream
            ChainArgs a{};
            a.x = x; a.ldx = kInCh; a.kgroups0 = 2; a.kvalid = kInCh; a.pe = (const float*)m->pe.p; a.T = (int)T;
            a.res = nullptr; a.n = n; a.nstages = 1;
            a.flags = fa.flags & (kPreChest | kPreNorm); a.factor = fa.factor; a.Tseq = (int)T;
            a.st[0] = stage_of(m, m->in_proj, ST_SET, XA, kTencD);
            if ((rc = launch_chain(m, a, st))) return rc;
        }
        const int nt = (int)((T + 15) / 16);
        // persistent: one workgroup per CU, bound to a head (blockIdx = 8 (4 slot + head) + xcd: 32 per sequence slot)
        const unsigned grid = (unsigned)std::max(32, m->num_cus / 32 * 32);
        const size_t qlds = (size_t)kQkvBlobBytes + 2 * ((size_t)2 * nt * 16 * 48 * 2 + (size_t)2 * kTencHd * kAttnVtRow * 2); // K rows of 48 halves
        for (int l = 0; l < m->nlayers; ++l) { // torch.nn.TransformerEncoderLayer, post-norm, ReLU
            auto& L = m->layers[l];
            AttnQkvArgs qa{};
            qa.x = XA; qa.out = OC; qa.T = (int)T; qa.B = B;
            for (int hd = 0; hd < kTencHeads; ++hd) qa.blob[hd] = (const float*)L.qkv_head[hd].buf16.p;
            switch (nt) {
#define B2H_AQ(N) case N: hipLaunchKernelGGL(b2h_attn_qkv_h3<N>, dim3(grid), dim3(64 * N), qlds, st, qa); break;
                B2H_AQ(1) B2H_AQ(2) B2H_AQ(3) B2H_AQ(4) B2H_AQ(5) B2H_AQ(6) B2H_AQ(7) B2H_AQ(8)
#undef B2H_AQ
                default: return fail(B2H_ERR_SHAPE, "TransformerEnc: T > 128");
            }
            // out_proj +res LN1 -> linear1 ReLU -> linear2 +res LN2 -> residual stream | hidden2pose (:171)
            ChainArgs a{};
            a.x = OC; a.ldx = kTencD; a.kgroups0 = 8; a.kvalid = kTencD; a.pe = nullptr; a.T = 1;
            a.res = XA; a.n = n;
            a.st[0] = stage_of(m, L.attn_out, ST_RESLN_GLOBAL, nullptr, kTencD);
            a.st[1] = stage_of(m, L.ff1, ST_RELU, nullptr, kTencD);
            if (l + 1 < m->nlayers) {
                a.st[2] = stage_of(m, L.ff2, ST_RESLN_REG, XA, kTencD); // the next layer's input and residual
                a.nstages = 3;
            } else {
                a.st[2] = stage_of(m, L.ff2, ST_RESLN_REG, nullptr, kTencD);
                a.st[3] = stage_of(m, m->out_proj, ST_STORE, y, kOutCh);
                a.nstages = 4;
                a.flags = fa.flags & (kPostDenorm | kPostMask); a.factor = fa.factor; a.n_frames = fa.n_frames;
            }
            a.Tseq = (int)T;
            if ((rc = launch_chain(m, a, st))) return rc;
        }
        HIP_TRY(hipGetLastError());
        return B2H_OK;
    }
    {   // src + pe -> pose2hidden_projection (HandPoseModels.py:167-169) -> layer 0's Q, K, V
        ChainArgs a{};
        a.x = x; a.ldx = kInCh; a.kgroups0 = 2; a.kvalid = kInCh; a.pe = (const float*)m->pe.p; a.T = (int)T;
        a.res = nullptr; a.n = n; a.nstages = 4;
        a.flags = fa.flags & (kPreChest | kPreNorm); a.factor = fa.factor; a.Tseq = (int)T;
        a.st[0] = stage_of(m, m->in_proj, ST_SET, XA, kTencD);
        a.st[1] = stage_of(m, m->layers[0].q, ST_STORE, QKV, 3 * kTencD);
        a.st[2] = stage_of(m, m->layers[0].k, ST_STORE, QKV + kTencD, 3 * kTencD);
        a.st[3] = stage_of(m, m->layers[0].v, ST_STORE, QKV + 2 * kTencD, 3 * kTencD);
        if ((rc = launch_chain(m, a, st))) return rc;
    }
    for (int l = 0; l < m->nlayers; ++l) { // torch.nn.TransformerEncoderLayer, post-norm, ReLU
        auto& L = m->layers[l];
        const int nt = (int)((T + 15) / 16);
        const dim3 ag((unsigned)(B * kTencHeads)), ab(64 * nt);
        // fp32: K and V rows padded to kAttnRow floats; f16x3: K hi/lo [keys][32] + V^T hi/lo [32][kAttnVtRow]
        const size_t alds = h3 ? ((size_t)2 * nt * 16 * kTencHd + (size_t)2 * kTencHd * kAttnVtRow) * 2
                               : (size_t)nt * 16 * kAttnRow * 8;
        switch (nt) {
#define B2H_ATTN(N)                                                                                   \
    case N:                                                                                           \
        if (h3) hipLaunchKernelGGL(b2h_attn_mfma_h3<N>, ag, ab, alds, st, QKV, OC, (int)T);            \
        else hipLaunchKernelGGL(b2h_attn_mfma_f32<N>, ag, ab, alds, st, QKV, OC, (int)T);              \
        break;
            B2H_ATTN(1) B2H_ATTN(2) B2H_ATTN(3) B2H_ATTN(4) B2H_ATTN(5) B2H_ATTN(6) B2H_ATTN(7) B2H_ATTN(8)
#undef B2H_ATTN
            default: return fail(B2H_ERR_SHAPE, "TransformerEnc: T > 128");
        }
        // out_proj +res LN1 -> linear1 ReLU -> linear2 +res LN2 -> next layer's Q,K,V | hidden2pose (:171)
        ChainArgs a{};
        a.x = OC; a.ldx = kTencD; a.kgroups0 = 8; a.kvalid = kTencD; a.pe = nullptr; a.T = 1;
        a.res = XA; a.n = n;
        a.st[0] = stage_of(m, L.attn_out, ST_RESLN_GLOBAL, nullptr, kTencD);
        a.st[1] = stage_of(m, L.ff1, ST_RELU, nullptr, kTencD);
        if (l + 1 < m->nlayers) {
            a.st[2] = stage_of(m, L.ff2, ST_RESLN_REG, XA, kTencD); // the next layer's residual
            a.st[3] = stage_of(m, m->layers[l + 1].q, ST_STORE, QKV, 3 * kTencD);
            a.st[4] = stage_of(m, m->layers[l + 1].k, ST_STORE, QKV + kTencD, 3 * kTencD);
            a.st[5] = stage_of(m, m->layers[l + 1].v, ST_STORE, QKV + 2 * kTencD, 3 * kTencD);
            a.nstages = 6;
        } else {
            a.st[2] = stage_of(m, L.ff2, ST_RESLN_REG, nullptr, kTencD);
            a.st[3] = stage_of(m, m->out_proj, ST_STORE, y, kOutCh);
            a.nstages = 4;
            a.flags = fa.flags & (kPostDenorm | kPostMask); a.factor = fa.factor; a.n_frames = fa.n_frames;
        }
        a.Tseq = (int)T;
        if ((rc = launch_chain(m, a, st))) return rc;
    }
    HIP_TRY(hipGetLastError());
    return B2H_OK;
}
} // namespace

extern "C" {

int b2h_tenc_forward(b2h_tenc* m, const float* x, float* y, int64_t B, int64_t T, void* workspace,
                     size_t workspace_bytes, void* stream) {
    FusedArgs fa{0, 1.0f, nullptr};
    return tenc_launch(m, x, y, B, T, fa, workspace, workspace_bytes, stream);
}

int b2h_tenc_forward_fused(b2h_tenc* m, const float* body, float* y, int64_t B, int64_t T, int flags, float factor,
                           const int64_t* n_frames, void* workspace, size_t workspace_bytes, void* stream) {
    if (flags & ~(kPreChest | kPreNorm | kPostDenorm | kPostMask)) return fail(B2H_ERR_INVALID, "unknown flag bits");
    if ((flags & (kPreNorm | kPostDenorm)) && !(factor > 0.f)) return fail(B2H_ERR_INVALID, "factor must be > 0");
    if ((flags & kPostMask) && !n_frames) return fail(B2H_ERR_INVALID, "B2H_POST_MASK_TAIL needs n_frames");
    FusedArgs fa{flags, factor, n_frames};
    return tenc_launch(m, body, y, B, T, fa, workspace, workspace_bytes, stream);
}

} // extern "C"

extern "C" {

int b2h_version(void) { return B2H_VERSION; }
int b2h_build_flags(void) { return B2H_ABLATE; }

const char* b2h_last_error(void) { return g_err.c_str(); }

int b2h_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int b2h_create(int conv_channels, const char* activation, int pos_emb, b2h_model** out) {
    if (!out) return fail(B2H_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // HandPoseModels.py:34-37: only "ReLU" is accepted, anything else raises ValueError
    if (!activation || std::strcmp(activation, "ReLU") != 0)
        return fail(B2H_ERR_INVALID, "activation must be \"ReLU\" (HandPoseModels.py:34-37)");
    if (conv_channels < 1 || conv_channels > kMaxWidth)
        return fail(B2H_ERR_INVALID, "conv_channels must be in [1, 128]");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(B2H_ERR_NO_DEVICE, "no HIP device visible (libb2h has no CPU path)");
    std::unique_ptr<b2h_model> m(new b2h_model()); // released to the caller only on success
    HIP_TRY(hipGetDevice(&m->device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, m->device));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(B2H_ERR_NO_DEVICE, std::string("device is ") + p.gcnArchName + ", libb2h is built for gfx950 only");
    m->num_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    m->C = conv_channels;
    m->pos_emb = pos_emb ? 1 : 0;
    const int C = conv_channels;
    const int cin[4] = {kInCh + m->pos_emb, C, C, C}, cout[4] = {C, C, C, kOutCh};
    for (int l = 0; l < 4; ++l) { m->cin[l] = cin[l]; m->cout[l] = cout[l]; }
    *out = m.release();
    return B2H_OK;
}

int b2h_destroy(b2h_model* m) {
    delete m;
    return B2H_OK;
}

int b2h_load_weights(b2h_model* m, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, const float* w4, const float* b4, int on_device) {
    if (!m) return fail(B2H_ERR_INVALID, "model is NULL");
    const float* ws[4] = {w1, w2, w3, w4};
    const float* bs[4] = {b1, b2, b3, b4};
    HostWeights hw;
    hw.m = m;
    for (int l = 0; l < 4; ++l) {
        if (!ws[l] || !bs[l]) return fail(B2H_ERR_INVALID, "weight pointer is NULL");
        hw.w[l].resize((size_t)m->cout[l] * m->cin[l] * kTaps);
        hw.b[l].resize(m->cout[l]);
        if (on_device) {
            HIP_TRY(hipMemcpy(hw.w[l].data(), ws[l], hw.w[l].size() * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(hw.b[l].data(), bs[l], hw.b[l].size() * 4, hipMemcpyDeviceToHost));
        } else {
            std::memcpy(hw.w[l].data(), ws[l], hw.w[l].size() * 4);
            std::memcpy(hw.b[l].data(), bs[l], hw.b[l].size() * 4);
        }
    }
    HIP_TRY(hipDeviceSynchronize()); // no launch may still read the old packed buffers
    int rc = check_device(m->device);
    if (rc) return rc;
    if ((rc = set_conv_kernel_attributes())) return rc;
    if ((rc = pack_all(m, hw))) return rc;
    m->has_weights = true;
    return B2H_OK;
}

int b2h_forward(b2h_model* m, const float* x, float* y, int64_t B, int64_t T, int kernel, void* stream) {
    FusedArgs fa{0, 1.0f, nullptr};
    return launch(m, x, y, B, T, kernel, fa, (hipStream_t)stream);
}

int b2h_forward_fused(b2h_model* m, const float* body, float* y, int64_t B, int64_t T, int flags, float factor,
                      const int64_t* n_frames, int kernel, void* stream) {
    if (flags & ~(kPreChest | kPreNorm | kPostDenorm | kPostMask)) return fail(B2H_ERR_INVALID, "unknown flag bits");
    if ((flags & (kPreNorm | kPostDenorm)) && !(factor > 0.f)) return fail(B2H_ERR_INVALID, "factor must be > 0");
    FusedArgs fa{flags, factor, n_frames};
    return launch(m, body, y, B, T, kernel, fa, (hipStream_t)stream);
}

int b2h_target_transform(const float* body, const float* hand, float* hand_out, int64_t B, int64_t T, int flags,
                         float factor, void* stream) {
    if (B < 0 || T < 0) return fail(B2H_ERR_SHAPE, "negative shape");
    if (B * T == 0) return B2H_OK;
    if (B > 0x7fffffff || T > (1 << 24)) return fail(B2H_ERR_SHAPE, "shape too large");
    if (flags & ~3) return fail(B2H_ERR_INVALID, "unknown flag bits");
    if ((flags & 2) && !(factor > 0.f)) return fail(B2H_ERR_INVALID, "factor must be > 0");
    int rc;
    if ((rc = check_device_ptr(body, "body")) || (rc = check_device_ptr(hand, "hand")) ||
        (rc = check_device_ptr(hand_out, "hand_out")))
        return rc;
    const int64_t n = B * T * 21;
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(b2h_target_transform_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, body,
                       hand, hand_out, B * T, flags, factor);
    HIP_TRY(hipGetLastError());
    return B2H_OK;
}

namespace {
int l1_metric(const float* pred, const float* target, const float* scores, const int64_t* n_frames, int64_t B,
              int64_t T, float* per_seq, float* loss, void* stream) {
    if (B < 1 || T < 1) return fail(B2H_ERR_SHAPE, "the L1 metrics need B >= 1 and T >= 1");
    if (B > 0x7fffffff || T > (1 << 24)) return fail(B2H_ERR_SHAPE, "shape too large");
    int rc;
    if ((rc = check_device_ptr(pred, "pred")) || (rc = check_device_ptr(target, "target")) ||
        (rc = check_device_ptr(per_seq, "per_seq")) || (rc = check_device_ptr(loss, "loss")) ||
        (scores && (rc = check_device_ptr(scores, "scores"))) || (n_frames && (rc = check_device_ptr(n_frames, "n_frames"))))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    if (scores)
        hipLaunchKernelGGL(b2h_masked_l1_seq_kernel<true>, dim3((unsigned)B), dim3(256), 0, st, pred, target, scores,
                           n_frames, per_seq, (int)T);
    else
        hipLaunchKernelGGL(b2h_masked_l1_seq_kernel<false>, dim3((unsigned)B), dim3(256), 0, st, pred, target, scores,
                           n_frames, per_seq, (int)T);
    hipLaunchKernelGGL(b2h_mean_kernel, dim3(1), dim3(256), 0, st, per_seq, loss, B, scores ? 0 : 1);
    HIP_TRY(hipGetLastError());
    return B2H_OK;
}
} // namespace

int b2h_masked_l1(const float* pred, const float* target, const int64_t* n_frames, int64_t B, int64_t T,
                  float* per_seq, float* loss, void* stream) {
    return l1_metric(pred, target, nullptr, n_frames, B, T, per_seq, loss, stream);
}

int b2h_weighted_l1(const float* pred, const float* target, const float* scores, const int64_t* n_frames, int64_t B,
                    int64_t T, float* per_seq, float* loss, void* stream) {
    if (!scores) return fail(B2H_ERR_INVALID, "scores is NULL");
    return l1_metric(pred, target, scores, n_frames, B, T, per_seq, loss, stream);
}

int b2h_model_info(const b2h_model* m, int* conv_channels, int* pos_emb, int* has_weights) {
    if (!m) return fail(B2H_ERR_INVALID, "model is NULL");
    if (conv_channels) *conv_channels = m->C;
    if (pos_emb) *pos_emb = m->pos_emb;
    if (has_weights) *has_weights = m->has_weights ? 1 : 0;
    return B2H_OK;
}

int b2h_kernel_supported(const b2h_model* m, int kernel) {
    if (!m) return 0;
    return kernel_ok(m, resolve_kernel(m, kernel)) ? 1 : 0;
}

const char* b2h_kernel_name(const b2h_model* m, int kernel) {
    if (!m) return "";
    switch (resolve_kernel(m, kernel)) {
        case B2H_KERNEL_F32_VALU: return "b2h_fwd_f32_valu";
        case B2H_KERNEL_F32_MFMA: return m->C > kMfmaWidth ? "b2h_fwd_mfma_f32<false, true>" : "b2h_fwd_mfma_f32<false, false>";
        case B2H_KERNEL_BF16_MFMA: return m->C > kMfmaWidth ? "b2h_fwd_mfma16w<1, false>" : "b2h_fwd_mfma16<1, false, true>";
        case B2H_KERNEL_F16_MFMA: return m->C > kMfmaWidth ? "b2h_fwd_mfma16w<2, false>" : "b2h_fwd_mfma16<2, false, true>";
        case B2H_KERNEL_F16X3_MFMA: return m->C > kMfmaWidth ? "b2h_fwd_mfma_f16x3w<false>" : "b2h_fwd_mfma_f16x3<false>";
        default: return "";
    }
}

int b2h_time_forward(b2h_model* m, const float* x, float* y, int64_t B, int64_t T, int kernel, int iters,
                     void* stream, float* avg_ms) {
    if (iters < 1 || !avg_ms) return fail(B2H_ERR_INVALID, "iters < 1 or avg_ms NULL");
    hipStream_t st = (hipStream_t)stream;
    Event e0, e1;
    HIP_TRY(hipEventCreate(&e0.e));
    HIP_TRY(hipEventCreate(&e1.e));
    FusedArgs fa{0, 1.0f, nullptr};
    int rc = B2H_OK;
    HIP_TRY(hipEventRecord(e0.e, st));
    for (int i = 0; i < iters && rc == B2H_OK; ++i) rc = launch(m, x, y, B, T, kernel, fa, st);
    HIP_TRY(hipEventRecord(e1.e, st));
    HIP_TRY(hipEventSynchronize(e1.e));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0.e, e1.e));
    if (rc) return rc;
    *avg_ms = ms / iters;
    return B2H_OK;
}

int b2h_stream_sync(void* stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return B2H_OK;
}

} // extern "C"

#include "dev/b2h_dev_exports.h" // empty unless a B2H_ABLATE stamp build
