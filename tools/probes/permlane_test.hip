// standalone check of the permlane-swap reductions used by attn.hip:  hipcc --offload-arch=gfx950 permlane_test.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void pl_pair16(float v, float& x, float& y) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("" : "+v"(w));
    auto r = __builtin_amdgcn_permlane16_swap(u, w, false, false);
    x = __builtin_bit_cast(float, r[0]); y = __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ void pl_pair32(float v, float& x, float& y) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("" : "+v"(w));
    auto r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
    x = __builtin_bit_cast(float, r[0]); y = __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float v = (float)lane, x, y;
    pl_pair16(v, x, y);
    out[lane] = x; out[64 + lane] = y;
    pl_pair32(v, x, y);
    out[128 + lane] = x; out[192 + lane] = y;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"p16.x", "p16.y", "p32.x", "p32.y"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %2.0f", h[a * 64 + i]); printf("\n"); }
    return 0;
}
