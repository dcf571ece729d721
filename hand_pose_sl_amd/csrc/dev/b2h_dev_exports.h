// Development-only exports (not in include/b2h.h); compiled only into B2H_ABLATE stamp builds.
#pragma once
#include "b2h_dev.h"
#if B2H_ABLATE & 16384
extern "C" int b2h_debug_chain_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2h::g_chain_dbg), 8 * 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
#endif
#if B2H_ABLATE & 32768
extern "C" int b2h_debug_conv3_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2h::g_conv3_dbg), 4 * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
#endif
#if B2H_ABLATE & 65536
extern "C" int b2h_debug_conv16_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2h::g_conv16_dbg), 8 * 32 * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
extern "C" int b2h_debug_conv16_spans(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2h::g_conv16_span), 256 * 8 * 3 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
#endif
