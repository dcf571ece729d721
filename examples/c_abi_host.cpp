// Pure C++/HIP host program over the C ABI (include/b2h.h): no Python, no torch.
// What a compiled caller of this path would do -- create, load weights, forward on its own
// stream -- and the harness tests/test_c_abi_host.py uses to check the boundary end to end:
//
//   c_abi_host <in.bin> <out.bin> <kernel>
//     in.bin : int32 C, pos_emb, B, T, then w1 b1 w2 b2 w3 b3 w4 b4 (fp32, reference state_dict
//              layout), then x (B,T,12,2) fp32
//     out.bin: y (B,T,21,2) fp32
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude -o c_abi_host examples/c_abi_host.cpp \
//         -Lhand_pose_sl_amd/csrc -lb2h -Wl,-rpath,$PWD/hand_pose_sl_amd/csrc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "b2h.h"

#define CK(call)                                                                  \
    do {                                                                          \
        int rc_ = (call);                                                         \
        if (rc_ != B2H_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, b2h_last_error()); return 2; } \
    } while (0)
#define HK(call)                                                                  \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)

static bool read_all(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: %s in.bin out.bin kernel\n", argv[0]); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int hdr[4];
    if (!read_all(f, hdr, sizeof hdr)) return 1;
    const int C = hdr[0], pos_emb = hdr[1], B = hdr[2], T = hdr[3], cin1 = 24 + (pos_emb ? 1 : 0);
    const size_t nw[8] = {(size_t)C * cin1 * 5, (size_t)C, (size_t)C * C * 5, (size_t)C,
                          (size_t)C * C * 5, (size_t)C, (size_t)42 * C * 5, 42};
    std::vector<std::vector<float>> w(8);
    for (int i = 0; i < 8; ++i) { w[i].resize(nw[i]); if (!read_all(f, w[i].data(), nw[i] * 4)) return 1; }
    std::vector<float> x((size_t)B * T * 24), y((size_t)B * T * 42);
    if (!read_all(f, x.data(), x.size() * 4)) return 1;
    std::fclose(f);

    if (b2h_version() != B2H_VERSION) { std::fprintf(stderr, "header/library version mismatch\n"); return 1; }
    if (b2h_device_count() < 1) { std::fprintf(stderr, "no gfx950 device\n"); return 4; }

    b2h_model* m = nullptr;
    // the reference's ValueError for a wrong activation (HandPoseModels.py:34-37)
    if (b2h_create(C, "Tanh", pos_emb, &m) != B2H_ERR_INVALID || m != nullptr) return 5;
    CK(b2h_create(C, "ReLU", pos_emb, &m));
    float *dx, *dy;
    HK(hipMalloc(&dx, x.size() * 4));
    HK(hipMalloc(&dy, y.size() * 4));
    if (b2h_forward(m, dx, dy, B, T, std::atoi(argv[3]), nullptr) != B2H_ERR_NO_WEIGHTS) return 6;
    CK(b2h_load_weights(m, w[0].data(), w[1].data(), w[2].data(), w[3].data(), w[4].data(), w[5].data(),
                        w[6].data(), w[7].data(), /*on_device=*/0));
    hipStream_t st;
    HK(hipStreamCreate(&st));
    HK(hipMemcpyAsync(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice, st));
    CK(b2h_forward(m, dx, dy, B, T, std::atoi(argv[3]), st));
    HK(hipMemcpyAsync(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost, st));
    CK(b2h_stream_sync(st));
    int c = 0, p = 0, has = 0;
    CK(b2h_model_info(m, &c, &p, &has));
    if (c != C || p != pos_emb || !has) return 7;
    std::printf("kernel %s  B=%d T=%d\n", b2h_kernel_name(m, std::atoi(argv[3])), B, T);
    CK(b2h_destroy(m));
    HK(hipFree(dx)); HK(hipFree(dy)); HK(hipStreamDestroy(st));

    FILE* o = std::fopen(argv[2], "wb");
    if (!o || std::fwrite(y.data(), 4, y.size(), o) != y.size()) return 1;
    std::fclose(o);
    return 0;
}
