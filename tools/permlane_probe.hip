// Hardware / toolchain probe: v_permlane32_swap / v_permlane16_swap (gfx950) as cross-row reductions.
//   hipcc -O3 --offload-arch=gfx950 tools/permlane_probe.hip -o tools/_build/permlane_probe && tools/_build/permlane_probe
// For each lane prints whether sum over lanes {l, l^16, l^32, l^48} came out right, for the builtin used
// naively (both operands the same value) and with the second operand made opaque.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float quad_sum_naive(float s) {
    unsigned u = __builtin_bit_cast(unsigned, s);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    float t = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    unsigned v = __builtin_bit_cast(unsigned, t);
    auto r2 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __builtin_bit_cast(float, r2[0]) + __builtin_bit_cast(float, r2[1]);
}
__device__ __forceinline__ float quad_sum_asm(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float t = a + b, c = t, d = t;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
}
__device__ __forceinline__ float quad_sum_opaque(float s) { // builtin, second operand made opaque to the optimizer
    unsigned u = __builtin_bit_cast(unsigned, s), u2 = u;
    asm volatile("" : "+v"(u2));
    auto r = __builtin_amdgcn_permlane32_swap(u, u2, false, false);
    float t = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    unsigned v = __builtin_bit_cast(unsigned, t), v2 = v;
    asm volatile("" : "+v"(v2));
    auto r2 = __builtin_amdgcn_permlane16_swap(v, v2, false, false);
    return __builtin_bit_cast(float, r2[0]) + __builtin_bit_cast(float, r2[1]);
}
__global__ void k3(const float* in, float* o) { o[threadIdx.x] = quad_sum_opaque(in[threadIdx.x]); }
__global__ void k(const float* in, float* o1, float* o2) {
    const float s = in[threadIdx.x];
    o1[threadIdx.x] = quad_sum_naive(s);
    o2[threadIdx.x] = quad_sum_asm(s);
}
int main() {
    float h[64], r1[64], r2[64], *d, *d1, *d2;
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i >> 4)) * 100.f + i; // rows distinguishable
    hipMalloc(&d, 256); hipMalloc(&d1, 256); hipMalloc(&d2, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d1, d2);
    hipMemcpy(r1, d1, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, d2, 256, hipMemcpyDeviceToHost);
    k3<<<1, 64>>>(d, d1);
    float r3[64];
    hipMemcpy(r3, d1, 256, hipMemcpyDeviceToHost);
    int ok3 = 0;
    for (int i = 0; i < 64; ++i) ok3 += r3[i] == h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48];
    printf("builtin with an opaque second copy: %d / 64 lanes right\n", ok3);
    int ok1 = 0, ok2 = 0;
    for (int i = 0; i < 64; ++i) {
        const float want = h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48];
        ok1 += r1[i] == want; ok2 += r2[i] == want;
    }
    printf("builtin with identical operands: %d / 64 lanes right (lane 0: %g, want %g)\n", ok1, r1[0], h[0] + h[16] + h[32] + h[48]);
    printf("asm (s_nop 1 before) two copies: %d / 64 lanes right (lane 0: %g)\n", ok2, r2[0]);
    for (int row = 0; row < 4; ++row) printf("  row %d: asm result %g (want %g)\n", row, r2[16 * row], h[0] + h[16] + h[32] + h[48]);
    return 0;
}
