// Development probe (GPU box): which SIMD does each wave of a 512-thread workgroup land on?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/simd_map tools/simd_map.hip && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    extern __shared__ char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
    unsigned* d;
    const int blocks = 1024;
    hipMalloc(&d, blocks * 8 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 140 * 1024, 0, d);
    static unsigned h[1024 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int b = 0; b < 4; ++b) {
        printf("block %d simd:", b);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
        printf("   wave_id:");
        for (int w = 0; w < 8; ++w) printf(" %u", h[b * 8 + w] & 15);
        printf("\n");
    }
    for (int w = 0; w < 8; ++w) printf("wave %d -> simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    return 0;
}
