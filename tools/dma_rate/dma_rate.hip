// LDS-DMA fill rate of one MI355X, as the weight-streaming kernels of csrc/ use it (raw_ptr_buffer_load_lds, 16 B per lane, 1 KiB per
// wave instruction): how many bytes per clock a CU gets when
//   mode 0  every workgroup reads the SAME 40 KB chunk sequence in the SAME order        (gemm_rp / mlp_fused / xattn_tail before the rotation)
//   mode 1  the same chunks, every workgroup walking the 40 pieces of a chunk in its own rotation
//   mode 2  every workgroup reads its OWN chunks (distinct addresses, L2 / MALL resident after the first pass)
//   mode 3  as 0, but the four waves issue one piece each, wait, repeat (issue never queues up)
// one workgroup (4 waves) per CU, double-buffered 40 KB chunks, no math.  Build: hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 1) void dma_kernel(const char* src, size_t wg_stride, int nchunks_src, int iters, int mode, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (unsigned)(nchunks_src * 40960), 0x00020000);
    const int rot_w = mode == 1 ? (int)(blockIdx.x & 3) : 0, rot_q = mode == 1 ? (int)((blockIdx.x >> 2) % 10) : 0;
    unsigned vo[10];
    int ldo[10];
#pragma unroll
    for (int p = 0; p < 10; ++p) {
        int q = p + rot_q;
        if (q >= 10) q -= 10;
        const int piece = q * 4 + ((wave + rot_w) & 3);
        vo[p] = (unsigned)(piece * 1024 + lane * 16);
        ldo[p] = __builtin_amdgcn_readfirstlane(piece * 1024);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int c = it % nchunks_src, buf = it & 1;
        if (mode == 3) {
#pragma unroll
            for (int p = 0; p < 10; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(smem + buf * 40960 + ldo[p]), 16, vo[p], (unsigned)c * 40960u, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int p = 0; p < 10; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(smem + buf * 40960 + ldo[p]), 16, vo[p], (unsigned)c * 40960u, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x * 16] == 123 && iters < 0) ticks[0] = 0;  // keep the LDS writes observable
}

int main() {
    int ncu = 256;
    const int nchunks = 15, iters = 600;
    char* src;
    const size_t per_wg = (size_t)nchunks * 40960;
    CHECK(hipMalloc(&src, per_wg * ncu));
    CHECK(hipMemset(src, 1, per_wg * ncu));
    unsigned long long* ticks;
    CHECK(hipMalloc(&ticks, ncu * 8));
    CHECK(hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = {"same chunks, same order", "same chunks, rotated order", "own chunks per workgroup", "same chunks, one piece at a time"};
    for (int grid : {256, 64, 8}) {
        for (int mode = 0; mode < 4; ++mode) {
            const size_t stride = mode == 2 ? per_wg : 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(dma_kernel, dim3(grid), dim3(256), 81920, 0, src, stride, nchunks, iters, mode, ticks);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 0) continue;
                std::vector<unsigned long long> h(grid);
                CHECK(hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost));
                double avg = 0;
                for (auto v : h) avg += (double)v;
                avg /= grid;
                const double bytes = (double)iters * 40960;
                printf("grid %3d  %-34s %8.3f ms  %7.1f GB/s per CU  %7.2f TB/s total  %6.1f ticks per 1 KiB piece per wave  (%.0f ticks per chunk)\n", grid, names[mode], ms,
                       bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12, avg / iters / 10.0, avg / iters);
            }
        }
    }
    return 0;
}
