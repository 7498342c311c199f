// Backward-pass kernels of the LoRA fine-tuning step: everything that is not a GEMM.
// The dense contractions of the backward (conv / linear dgrad, attention products) reuse gemm.hip with transposed or
// flipped weight copies packed at finalize; base weights are frozen, so only dX and the LoRA wgrads are produced.
#include "common.h"
#include "prof.h"

namespace mrisr {

#define TRY_(expr)           \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

template <typename T> struct BVec;
template <> struct BVec<bf16> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct BVec<float> { static constexpr int N = 4; typedef f32x4 type; };

__device__ __forceinline__ float bw_wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float silu_grad(float x) {  // d/dx [x * sigmoid(x)]
    const float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float gelu_grad(float x) {  // d/dx [0.5 x (1 + erf(x / sqrt2))]
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
static inline unsigned bw_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward.  y = act(xhat * gamma + beta), xhat = (x - mean) * rstd per (sample, group).
//   ds = dy * act'(pre);  dxhat = ds * gamma;
//   dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat))
// Stage 1 accumulates S1 = sum dxhat, S2 = sum dxhat*xhat per (b, split, group) (same geometry as the forward stats);
// stage 2 applies.  mean/rstd are re-derived from the forward's saved partial sums (fwd_partial).
// The (concatenated) gradient is written to two destinations (dx0: first c0 channels, dx1: the rest), optionally
// accumulating into them.
// ------------------------------------------------------------------------------------------------
template <typename T, int VPT>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const GroupNormBwdArgs a, int slots, int RL) {
    constexpr int VE = BVec<T>::N;
    typedef typename BVec<T>::type vec_t;
    extern __shared__ float sm[];  // [2][RL][C]
    __shared__ float mean_s[64], rstd_s[64];
    const int C = a.c0 + a.c1;
    const int Cg = C / a.groups;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < a.groups) {
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < a.nsplit; ++q) {
            const float* p = a.fwd_partial + (((size_t)b * a.nsplit + q) * a.groups + tid) * 2;
            s1 += (double)p[0];
            s2 += (double)p[1];
        }
        const double n = (double)a.HW * Cg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    const int slot = tid % slots, rl = tid / slots;
    const int rows_per = (a.HW + a.nsplit - 1) / a.nsplit;
    const int r_beg = sp * rows_per, r_end = min(a.HW, r_beg + rows_per);
    float s1[VPT][VE], s2[VPT][VE];
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int e = 0; e < VE; ++e) s1[v][e] = s2[v][e] = 0.f;
    if (rl < RL) {
        for (int r = r_beg + rl; r < r_end; r += RL) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int ch = (slot * VPT + v) * VE;
                const T* src = ch < a.c0 ? reinterpret_cast<const T*>(a.x0) + ((size_t)b * a.HW + r) * a.c0 + ch
                                         : reinterpret_cast<const T*>(a.x1) + ((size_t)b * a.HW + r) * a.c1 + (ch - a.c0);
                const vec_t x = *reinterpret_cast<const vec_t*>(src);
                const vec_t dy = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(a.dy) + ((size_t)b * a.HW + r) * C + ch);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int c = ch + e;
                    const int gq = c / Cg;
                    const float xh = ((float)x[e] - mean_s[gq]) * rstd_s[gq];
                    float d = (float)dy[e];
                    if (a.silu) d *= silu_grad(xh * a.gamma[c] + a.beta[c]);
                    const float dxh = d * a.gamma[c];
                    s1[v][e] += dxh;
                    s2[v][e] += dxh * xh;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const int ch = (slot * VPT + v) * VE + e;
                sm[(size_t)rl * C + ch] = s1[v][e];
                sm[(size_t)(RL + rl) * C + ch] = s2[v][e];
            }
    }
    __syncthreads();
    if (tid < 2 * a.groups) {
        const int gq = tid % a.groups, which = tid / a.groups;
        double acc = 0.0;
        for (int r = 0; r < RL; ++r) {
            const float* row = sm + (size_t)(which * RL + r) * C + gq * Cg;
            float t = 0.f;
            for (int c = 0; c < Cg; ++c) t += row[c];
            acc += (double)t;
        }
        a.bwd_partial[(((size_t)b * a.nsplit + sp) * a.groups + gq) * 2 + which] = (float)acc;
    }
}

template <typename T, int VPT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GroupNormBwdArgs a, int slots, int RL, int rows_per_block) {
    constexpr int VE = BVec<T>::N;
    typedef typename BVec<T>::type vec_t;
    __shared__ float mean_s[64], rstd_s[64], m1_s[64], m2_s[64];
    const int C = a.c0 + a.c1;
    const int Cg = C / a.groups;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < a.groups) {
        double s1 = 0.0, s2 = 0.0, g1 = 0.0, g2 = 0.0;
        for (int q = 0; q < a.nsplit; ++q) {
            const size_t o = (((size_t)b * a.nsplit + q) * a.groups + tid) * 2;
            s1 += (double)a.fwd_partial[o];
            s2 += (double)a.fwd_partial[o + 1];
            g1 += (double)a.bwd_partial[o];
            g2 += (double)a.bwd_partial[o + 1];
        }
        const double n = (double)a.HW * Cg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        m1_s[tid] = (float)(g1 / n);
        m2_s[tid] = (float)(g2 / n);
    }
    __syncthreads();
    const int slot = tid % slots, rl = tid / slots;
    if (rl >= RL) return;
    const int r_beg = blockIdx.x * rows_per_block;
    const int r_end = min(a.HW, r_beg + rows_per_block);
    for (int r = r_beg + rl; r < r_end; r += RL) {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int ch = (slot * VPT + v) * VE;
            const bool first = ch < a.c0;
            const T* src = first ? reinterpret_cast<const T*>(a.x0) + ((size_t)b * a.HW + r) * a.c0 + ch
                                 : reinterpret_cast<const T*>(a.x1) + ((size_t)b * a.HW + r) * a.c1 + (ch - a.c0);
            T* dst = first ? reinterpret_cast<T*>(a.dx0) + ((size_t)b * a.HW + r) * a.c0 + ch
                           : reinterpret_cast<T*>(a.dx1) + ((size_t)b * a.HW + r) * a.c1 + (ch - a.c0);
            const bool acc = first ? a.acc0 : a.acc1;
            const vec_t x = *reinterpret_cast<const vec_t*>(src);
            const vec_t dy = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(a.dy) + ((size_t)b * a.HW + r) * C + ch);
            vec_t o;
            if (acc) o = *reinterpret_cast<const vec_t*>(dst);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const int c = ch + e;
                const int gq = c / Cg;
                const float xh = ((float)x[e] - mean_s[gq]) * rstd_s[gq];
                float d = (float)dy[e];
                if (a.silu) d *= silu_grad(xh * a.gamma[c] + a.beta[c]);
                const float dxh = d * a.gamma[c];
                float g = rstd_s[gq] * (dxh - m1_s[gq] - xh * m2_s[gq]);
                if (acc) g += (float)o[e];
                o[e] = from_f32<T>(g);
            }
            *reinterpret_cast<vec_t*>(dst) = o;
        }
    }
}

// GroupNorm backward in ONE pass (bf16), the mirror of gn_fused_kernel (norm.hip): a workgroup owns one image and a slab of
// whole groups, keeps x and dy of its part of the image in registers, folds sum(dxhat) and sum(dxhat * xhat) per group through
// LDS in a fixed order and writes dx from registers - x and dy are read once instead of twice, one launch instead of two.
template <int NVM, int VE>
__global__ __launch_bounds__(256, 1) void gn_bwd_fused_kernel(const GroupNormBwdArgs a, int slab, int slots, int RL, int nslab, int xcd_map) {
    typedef bf16 T;
    typedef __attribute__((ext_vector_type(VE))) __bf16 vec_t;
    __shared__ float part[256 * 16];
    __shared__ float seg[2 * 1280];
    __shared__ double gsum[2 * 80];
    __shared__ float mean_s[80], rstd_s[80], m1_s[80], m2_s[80];
    const int C = a.c0 + a.c1;
    const int Cg = C / a.groups;
    int b, sl;
    if (xcd_map) {
        const int L = blockIdx.x;
        b = (L & 7) + 8 * (L / (8 * nslab));
        sl = (L >> 3) % nslab;
    } else {
        b = blockIdx.x / nslab;
        sl = blockIdx.x - b * nslab;
    }
    const int ch0 = sl * slab;
    const int tid = threadIdx.x;
    const int ng = slab / Cg;
    if (tid < ng) {
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < a.nsplit; ++q) {
            const size_t o = (((size_t)b * a.nsplit + q) * a.groups + ch0 / Cg + tid) * 2;
            s1 += (double)a.fwd_partial[o];
            s2 += (double)a.fwd_partial[o + 1];
        }
        const double n = (double)a.HW * Cg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    const int slot = tid % slots, rl = tid / slots;
    const bool active = rl < RL;
    const int ch = ch0 + slot * VE;
    const bool first = ch < a.c0;
    const T* src = first ? reinterpret_cast<const T*>(a.x0) + (size_t)b * a.HW * a.c0 + ch
                         : reinterpret_cast<const T*>(a.x1) + (size_t)b * a.HW * a.c1 + (ch - a.c0);
    const int ld = first ? a.c0 : a.c1;
    const T* dyp = reinterpret_cast<const T*>(a.dy) + (size_t)b * a.HW * C + ch;
    vec_t xv[NVM], dv[NVM];
    if (active) {
#pragma unroll
        for (int i = 0; i < NVM; ++i) {
            const int r = rl + i * RL;
            if (r < a.HW) {
                xv[i] = *reinterpret_cast<const vec_t*>(src + (size_t)r * ld);
                dv[i] = *reinterpret_cast<const vec_t*>(dyp + (size_t)r * C);
            }
        }
    }
    __syncthreads();
    float mu[VE], rs[VE], ga[VE], be[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int gl = (slot * VE + e) / Cg;
        mu[e] = active ? mean_s[gl] : 0.f;
        rs[e] = active ? rstd_s[gl] : 0.f;
        ga[e] = active ? a.gamma[ch + e] : 0.f;
        be[e] = active ? a.beta[ch + e] : 0.f;
    }
    float s1[VE], s2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
#pragma unroll
        for (int i = 0; i < NVM; ++i) {
            if (rl + i * RL < a.HW) {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float xh = ((float)xv[i][e] - mu[e]) * rs[e];
                    float d = (float)dv[i][e];
                    if (a.silu) d *= silu_grad(xh * ga[e] + be[e]);
                    const float dxh = d * ga[e];
                    s1[e] += dxh;
                    s2[e] += dxh * xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[tid * 16 + e] = s1[e];
        part[tid * 16 + 8 + e] = s2[e];
    }
    __syncthreads();
    const int nout = 2 * slab;
    int nseg = 256 / nout;
    if (nseg < 1) nseg = 1;
    if (nseg > RL) nseg = RL;
    const int seg_len = (RL + nseg - 1) / nseg;
    for (int t = tid; t < nout * nseg; t += 256) {
        const int o = t % nout, sg = t / nout;
        const int which = o / slab, c = o - which * slab;
        const int off = (c / VE) * 16 + which * 8 + (c % VE);
        float acc = 0.f;
        const int r1 = min(RL, (sg + 1) * seg_len);
        for (int r = sg * seg_len; r < r1; ++r) acc += part[(r * slots) * 16 + off];
        seg[sg * nout + o] = acc;
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int p = wave; p < 2 * ng; p += 4) {
            const int gl = p % ng, which = p / ng;
            double acc = 0.0;
            for (int t = lane; t < Cg * nseg; t += 64) {
                const int sg = t / Cg, c = t - sg * Cg;
                acc += (double)seg[sg * nout + which * slab + gl * Cg + c];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) gsum[which * 80 + gl] = acc;
        }
    }
    __syncthreads();
    if (tid < ng) {
        const double n = (double)a.HW * Cg;
        m1_s[tid] = (float)(gsum[tid] / n);
        m2_s[tid] = (float)(gsum[80 + tid] / n);
    }
    __syncthreads();
    if (!active) return;
    float m1[VE], m2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int gl = (slot * VE + e) / Cg;
        m1[e] = m1_s[gl];
        m2[e] = m2_s[gl];
    }
    T* dst = first ? reinterpret_cast<T*>(a.dx0) + (size_t)b * a.HW * a.c0 + ch : reinterpret_cast<T*>(a.dx1) + (size_t)b * a.HW * a.c1 + (ch - a.c0);
    const bool acc = first ? a.acc0 : a.acc1;
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
        const int r = rl + i * RL;
        if (r < a.HW) {
            vec_t o;
            if (acc) o = *reinterpret_cast<const vec_t*>(dst + (size_t)r * ld);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float xh = ((float)xv[i][e] - mu[e]) * rs[e];
                float d = (float)dv[i][e];
                if (a.silu) d *= silu_grad(xh * ga[e] + be[e]);
                const float dxh = d * ga[e];
                float g = rs[e] * (dxh - m1[e] - xh * m2[e]);
                if (acc) g += (float)o[e];
                o[e] = (bf16)g;
            }
            *reinterpret_cast<vec_t*>(dst + (size_t)r * ld) = o;
        }
    }
}

template <typename T>
int launch_groupnorm_bwd(const GroupNormBwdArgs& a, hipStream_t st) {
    constexpr int VE = BVec<T>::N;
    const int C = a.c0 + a.c1;
    MRISR_REQUIRE(C % a.groups == 0 && a.groups <= 64 && a.c0 % VE == 0 && a.c1 % VE == 0, "GroupNorm backward geometry");
    const int nvec = C / VE;
    int vpt = 1;
    while (nvec / vpt > 256 || (nvec % vpt) != 0) ++vpt;
    MRISR_REQUIRE(vpt <= 4, "GroupNorm backward: too many channels");
    const int slots = nvec / vpt;
    int RL = 256 / slots;
    if (RL < 1) RL = 1;
    const size_t smem = (size_t)2 * RL * C * sizeof(float);
    const double act_bytes = (double)a.B * a.HW * C * sizeof(T);
    if constexpr (sizeof(T) == 2) {
        int slab = 0, fslots = 0, fRL = 0, nv = 0;
        if (gn_fused_geometry(a.c0, a.c1, a.groups, a.HW, &slab, &fslots, &fRL, &nv, 8)) {
            int ve = 8, s4 = 0, sl4 = 0, rl4 = 0, nv4 = 0;  // 8-byte-vector slabs when that doubles a thin grid (as in the forward)
            static const int ve4_ok = [] { const char* e = getenv("MRISR_GN_VE4"); return e ? atoi(e) : 1; }();
            if (ve4_ok && a.B * (C / slab) < 512 && gn_fused_geometry(a.c0, a.c1, a.groups, a.HW, &s4, &sl4, &rl4, &nv4, 4) &&
                a.B * (C / s4) > a.B * (C / slab)) {
                ve = 4; slab = s4; fslots = sl4; fRL = rl4; nv = nv4;
            }
            const int nslab = C / slab;
            const int xmap = (a.B % 8) == 0 ? 1 : 0;
            ProfScope ps("groupnorm_bwd_fused", 0.0, 3.0 * act_bytes, st);
            const dim3 fg(a.B * nslab);
#define GNB_GO(NV, VEV) hipLaunchKernelGGL((gn_bwd_fused_kernel<NV, VEV>), fg, dim3(256), 0, st, a, slab, fslots, fRL, nslab, xmap)
            if (ve == 8) {
                if (nv <= 2) GNB_GO(2, 8); else if (nv <= 4) GNB_GO(4, 8); else if (nv <= 8) GNB_GO(8, 8); else if (nv <= 16) GNB_GO(16, 8); else GNB_GO(24, 8);
            } else {
                if (nv <= 2) GNB_GO(2, 4); else if (nv <= 4) GNB_GO(4, 4); else if (nv <= 8) GNB_GO(8, 4); else if (nv <= 16) GNB_GO(16, 4); else GNB_GO(24, 4);
            }
#undef GNB_GO
            MRISR_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    {
        ProfScope ps("groupnorm_bwd_stats", 0.0, 2.0 * act_bytes, st);
        dim3 grid(a.nsplit, a.B);
        switch (vpt) {
            case 1: hipLaunchKernelGGL((gn_bwd_stats_kernel<T, 1>), grid, dim3(256), smem, st, a, slots, RL); break;
            case 2: hipLaunchKernelGGL((gn_bwd_stats_kernel<T, 2>), grid, dim3(256), smem, st, a, slots, RL); break;
            case 3: hipLaunchKernelGGL((gn_bwd_stats_kernel<T, 3>), grid, dim3(256), smem, st, a, slots, RL); break;
            default: hipLaunchKernelGGL((gn_bwd_stats_kernel<T, 4>), grid, dim3(256), smem, st, a, slots, RL); break;
        }
    }
    MRISR_CHECK_HIP(hipGetLastError());
    int bx = 2048 / (a.B > 0 ? a.B : 1) + 1;
    int rows_per_block = (a.HW + bx - 1) / bx;
    if (rows_per_block < RL) rows_per_block = RL;
    bx = (a.HW + rows_per_block - 1) / rows_per_block;
    ProfScope ps2("groupnorm_bwd_apply", 0.0, 3.0 * act_bytes, st);
    switch (vpt) {
        case 1: hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 1>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        case 2: hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 2>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        case 3: hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 3>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        default: hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 4>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, row statistics recomputed in registers.
//   dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * gamma;  dx (+)= into dst
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* dx,
                                                            const float* __restrict__ gamma, int M, int C, float eps,
                                                            int accumulate) {
    constexpr int VE = BVec<T>::N;
    typedef typename BVec<T>::type vec_t;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C / VE;
    float xv[MAXV][VE], dv[MAXV][VE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const vec_t t = *reinterpret_cast<const vec_t*>(x + (size_t)row * C + vi * VE);
            const vec_t d = *reinterpret_cast<const vec_t*>(dy + (size_t)row * C + vi * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                xv[i][e] = (float)t[e];
                dv[i][e] = (float)d[e] * gamma[vi * VE + e];
                s += xv[i][e];
            }
        }
    }
    const float mean = bw_wsum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float d = xv[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(bw_wsum(q) / (float)C + eps);
    float g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                xv[i][e] = (xv[i][e] - mean) * rstd;  // xhat
                g1 += dv[i][e];
                g2 += dv[i][e] * xv[i][e];
            }
        }
    }
    g1 = bw_wsum(g1) / (float)C;
    g2 = bw_wsum(g2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            T* dst = dx + (size_t)row * C + vi * VE;
            vec_t o;
            if (accumulate) o = *reinterpret_cast<const vec_t*>(dst);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float g = rstd * (dv[i][e] - g1 - xv[i][e] * g2);
                if (accumulate) g += (float)o[e];
                o[e] = from_f32<T>(g);
            }
            *reinterpret_cast<vec_t*>(dst) = o;
        }
    }
}

template <typename T>
int launch_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, int M, int C, float eps,
                         int accumulate, hipStream_t st) {
    constexpr int VE = BVec<T>::N;
    MRISR_REQUIRE(C % VE == 0, "LayerNorm backward channel alignment");
    const int need = (C / VE + 63) / 64;
    const dim3 grid((M + 3) / 4);
    const T* xi = reinterpret_cast<const T*>(x);
    const T* di = reinterpret_cast<const T*>(dy);
    T* o = reinterpret_cast<T*>(dx);
    ProfScope ps("layernorm_bwd", 0.0, 3.0 * M * (double)C * sizeof(T), st);
    if (need <= 1) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 1>), grid, dim3(256), 0, st, xi, di, o, gamma, M, C, eps, accumulate);
    else if (need <= 2) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 2>), grid, dim3(256), 0, st, xi, di, o, gamma, M, C, eps, accumulate);
    else if (need <= 3) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 3>), grid, dim3(256), 0, st, xi, di, o, gamma, M, C, eps, accumulate);
    else if (need <= 5) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 5>), grid, dim3(256), 0, st, xi, di, o, gamma, M, C, eps, accumulate);
    else MRISR_REQUIRE(false, "LayerNorm backward: row too long");
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GEGLU backward.  Forward (epilogue of ff.net.0.proj): out[m][j] = u * gelu(g) with (u, g) interleaved in blocks of 16
// in the projection's column space.  pre: [M][8C] the saved pre-activation in that interleaved layout;
// dout: [M][4C];  dpre: [M][8C] (same interleaved layout, so the projection's dgrad uses the packed weights as is).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dout, T* __restrict__ dpre, long long M,
                                 int half) {
    const long long total = M * half;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / half;
        const int j = (int)(i - m * half);
        const int cu = (j >> 4) * 32 + (j & 15), cg = cu + 16;
        const float u = to_f32(pre[m * 2 * half + cu]), g = to_f32(pre[m * 2 * half + cg]);
        const float d = to_f32(dout[i]);
        dpre[m * 2 * half + cu] = from_f32<T>(d * gelu_erf_t<T>(g));
        dpre[m * 2 * half + cg] = from_f32<T>(d * u * gelu_grad(g));
    }
}
template <typename T>
int launch_geglu_bwd(const void* pre, const void* dout, void* dpre, long long M, int half, hipStream_t st) {
    ProfScope ps("geglu_bwd", 0.0, (double)M * half * sizeof(T) * 5.0, st);
    hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3(bw_blocks(M * half)), dim3(256), 0, st, reinterpret_cast<const T*>(pre),
                       reinterpret_cast<const T*>(dout), reinterpret_cast<T*>(dpre), M, half);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// GEGLU forward from a stored pre-activation (training keeps `pre` for the backward instead of fusing the gate into
// the projection's epilogue)
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ pre, T* __restrict__ out, long long M, int half) {
    const long long total = M * half;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / half;
        const int j = (int)(i - m * half);
        const int cu = (j >> 4) * 32 + (j & 15);
        out[i] = from_f32<T>(to_f32(pre[m * 2 * half + cu]) * gelu_erf_t<T>(to_f32(pre[m * 2 * half + cu + 16])));
    }
}
template <typename T>
int launch_geglu_fwd(const void* pre, void* out, long long M, int half, hipStream_t st) {
    hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3(bw_blocks(M * half)), dim3(256), 0, st, reinterpret_cast<const T*>(pre),
                       reinterpret_cast<T*>(out), M, half);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// token rows [B][N][ldx] (columns col_off .. col_off + H*hd) -> head-major [B*H][npad][dpad]; pads must be pre-zeroed
template <typename T>
__global__ void rows_to_heads_kernel2(const T* __restrict__ x, int ldx, int col_off, T* __restrict__ dst, int B, int N, int H, int hd,
                                      int npad, int dpad) {
    const long long total = (long long)B * N * H * hd;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int dd = (int)(i % hd);
        const int h = (int)((i / hd) % H);
        const int tok = (int)((i / ((long long)hd * H)) % N);
        const int b = (int)(i / ((long long)hd * H * N));
        dst[(((size_t)b * H + h) * npad + tok) * dpad + dd] = x[((size_t)b * N + tok) * ldx + col_off + h * hd + dd];
    }
}
template <typename T>
int launch_rows_to_heads(const void* x, int ldx, int col_off, void* dst, int B, int N, int H, int hd, int npad, int dpad,
                         hipStream_t st) {
    hipLaunchKernelGGL(rows_to_heads_kernel2<T>, dim3(bw_blocks((long long)B * N * H * hd)), dim3(256), 0, st,
                       reinterpret_cast<const T*>(x), ldx, col_off, reinterpret_cast<T*>(dst), B, N, H, hd, npad, dpad);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// softmax backward on materialised rows:  dS = scale * P * (dP - sum_k dP*P);  one wave per row.
//   P (T) [rows][ld], dP f32 [rows][ld] -> dS (T) [rows][ld]; columns >= nk are written as 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, const float* __restrict__ dp, T* __restrict__ ds,
                                                          int ld, long long rows, int nk, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* pr = p + row * ld;
    const float* dr = dp + row * ld;
    float s = 0.f;
    for (int c = lane; c < nk; c += 64) s += to_f32(pr[c]) * dr[c];
    s = bw_wsum(s);
    T* o = ds + row * ld;
    for (int c = lane; c < ld; c += 64) o[c] = from_f32<T>(c < nk ? scale * to_f32(pr[c]) * (dr[c] - s) : 0.f);
}
// vectorised variant: P and dP rows in registers, 16-byte accesses.  Needs ld % 4 == 0, ld <= 256 * MAXV.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const T* __restrict__ p, const float* __restrict__ dp, T* __restrict__ ds,
                                                              int ld, long long rows, int nk, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* pr = p + row * ld;
    const float* dr = dp + row * ld;
    f32x4 pv[MAXV], dv[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < ld) {
            dv[i] = *reinterpret_cast<const f32x4*>(dr + c);
            if constexpr (sizeof(T) == 2) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(pr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[i][e] = (float)t[e];
            } else {
                pv[i] = *reinterpret_cast<const f32x4*>(pr + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < nk) s += pv[i][e] * dv[i][e];
        }
    }
    s = bw_wsum(s);
    T* o = ds + row * ld;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < ld) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = c + e < nk ? scale * pv[i][e] * (dv[i][e] - s) : 0.f;
            if constexpr (sizeof(T) == 2) {
                bf16x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (bf16)r[e];
                *reinterpret_cast<bf16x4*>(o + c) = t;
            } else {
                *reinterpret_cast<f32x4*>(o + c) = r;
            }
        }
    }
}
template <typename T>
int launch_softmax_bwd(const void* p, const float* dp, void* ds, int ld, long long rows, int nk, float scale, hipStream_t st) {
    ProfScope ps("softmax_bwd", 0.0, (double)rows * ld * (4.0 + 2.0 * sizeof(T)), st);
    const dim3 grid((unsigned)((rows + 3) / 4));
    const T* pp = reinterpret_cast<const T*>(p);
    T* dd = reinterpret_cast<T*>(ds);
    if (ld % 4 == 0 && ld <= 4096) {
        const int need = (ld / 4 + 63) / 64;
        if (need <= 1) hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 1>), grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
        else if (need <= 2) hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 2>), grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
        else if (need <= 4) hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 4>), grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
        else if (need <= 8) hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 8>), grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
        else hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 16>), grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
    } else {
        hipLaunchKernelGGL(softmax_bwd_kernel<T>, grid, dim3(256), 0, st, pp, dp, dd, ld, rows, nk, scale);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// batched transpose  dst[z][c][r] = src[z][r][c]   (R x C -> C x R per batch), 32x32 tiles through LDS.
// Rows >= R_valid of src are treated as zero (so padded head buffers transpose to zero-padded columns).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int C, int ld_src,
                                                        int ld_dst, long long bs_src, long long bs_dst, int r_valid) {
    __shared__ T tile[32][33];
    const int z = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < r_valid && r < R && c < C) ? src[(size_t)z * bs_src + (size_t)r * ld_src + c] : from_f32<T>(0.f);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)z * bs_dst + (size_t)c * ld_dst + r] = tile[tx][i];
    }
}
// 64x64 tiles, 16-byte global accesses on both sides (needs R, C, pitches and batch strides to be multiples of a vector)
template <typename T>
__global__ __launch_bounds__(256) void transpose_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int C, int ld_src,
                                                            int ld_dst, long long bs_src, long long bs_dst, int r_valid) {
    constexpr int VE = BVec<T>::N;
    constexpr int VPR = 64 / VE;
    constexpr int LDT = 64 + (sizeof(T) == 2 ? 2 : 1);
    typedef typename BVec<T>::type vec_t;
    __shared__ T tile[64 * LDT];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    src += (size_t)blockIdx.z * bs_src;
    dst += (size_t)blockIdx.z * bs_dst;
    for (int idx = threadIdx.x; idx < 64 * VPR; idx += 256) {
        const int row = idx / VPR, cv = idx - row * VPR;
        const int r = r0 + row, c = c0 + cv * VE;
        vec_t v;
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = from_f32<T>(0.f);
        if (r < R && r < r_valid && c < C) v = *reinterpret_cast<const vec_t*>(src + (size_t)r * ld_src + c);
#pragma unroll
        for (int e = 0; e < VE; ++e) tile[row * LDT + cv * VE + e] = v[e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * VPR; idx += 256) {
        const int crow = idx / VPR, g = idx - crow * VPR;
        const int c = c0 + crow, r = r0 + g * VE;
        if (c < C && r < R) {
            vec_t v;
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = tile[(g * VE + e) * LDT + crow];
            *reinterpret_cast<vec_t*>(dst + (size_t)c * ld_dst + r) = v;
        }
    }
}
template <typename T>
int launch_transpose(const void* src, void* dst, int R, int C, int ld_src, int ld_dst, long long bs_src, long long bs_dst,
                     int batch, int r_valid, hipStream_t st) {
    constexpr int VE = BVec<T>::N;
    ProfScope ps("transpose", 0.0, 2.0 * batch * (double)R * C * sizeof(T), st);
    const bool vec = R % VE == 0 && C % VE == 0 && ld_src % VE == 0 && ld_dst % VE == 0 && bs_src % VE == 0 && bs_dst % VE == 0 &&
                     ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
    if (vec) {
        dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
        hipLaunchKernelGGL(transpose_vec_kernel<T>, grid, dim3(256), 0, st, reinterpret_cast<const T*>(src), reinterpret_cast<T*>(dst), R,
                           C, ld_src, ld_dst, bs_src, bs_dst, r_valid);
    } else {
        dim3 grid((C + 31) / 32, (R + 31) / 32, batch);
        hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, st, reinterpret_cast<const T*>(src), reinterpret_cast<T*>(dst), R, C,
                           ld_src, ld_dst, bs_src, bs_dst, r_valid);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LoRA weight gradients: rank-r outer-product sums over the M rows of an activation, one streaming pass.
//   mode 0 (dB): out_j[c'][q] += scale * sum_m P[m][c] Q[m][j*r+q],  j = c / secN, c' = c - j*secN   (P = dY, Q = z)
//   mode 1 (dA): out_j[q][c]  += scale * sum_m P[m][c] Q[m][j*r+q]   for every fused module j         (P = x,  Q = dz)
// HBM-bound (8*NQ FMAs per 16 bytes of P).  A block owns a slab of rows and up to 256 16-byte channel chunks; its
// threads tile (row interleave, chunk) so a wave reads contiguous row segments, four rows in flight per thread; Q rows
// are broadcast loads.  The row interleaves fold through LDS atomics into one partial tile per block (plain stores,
// no global atomics: a thousand blocks hitting the same few KB serialise in L2), then a second tiny kernel sums the
// partial tiles and adds them to the gradient vector.
// ------------------------------------------------------------------------------------------------
struct LoraWgradArgs {
    const void* P = nullptr;
    int ldp = 0;
    const float* Q = nullptr;
    int ldq = 0;
    int M = 0, C = 0, mode = 0, r = 0, nmod = 1, secN = 0, qbase = 0;
    float* out[3] = {nullptr, nullptr, nullptr};
    float scale = 1.0f;
    float* partial = nullptr;  // [gy][gx * cxb * VE * NQ]
    int cxb = 0, RL = 1, rpb = 0, gx = 1, gy = 1;
};
static void lora_wgrad_geom(LoraWgradArgs& a, int VE) {
    const int cx = a.C / VE;
    a.gx = (cx + 255) / 256;
    a.cxb = (cx + a.gx - 1) / a.gx;
    a.RL = 256 / a.cxb;
    a.rpb = a.RL * 16;
    a.gy = (a.M + a.rpb - 1) / a.rpb;
    if (a.gy > 512) {
        a.rpb = ((a.M + 511) / 512 + a.RL - 1) / a.RL * a.RL;
        a.gy = (a.M + a.rpb - 1) / a.rpb;
    }
}
size_t lora_wgrad_scratch_bytes(int M, int C, int nq, int elem_size) {
    LoraWgradArgs a;
    a.M = M; a.C = C;
    const int VE = 16 / elem_size;
    lora_wgrad_geom(a, VE);
    return (size_t)a.gy * a.gx * a.cxb * VE * nq * sizeof(float);
}
template <typename T, int NQ>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(const LoraWgradArgs a) {
    constexpr int VE = BVec<T>::N;
    typedef typename BVec<T>::type vec_t;
    extern __shared__ float red[];  // [cxb * VE * NQ] when RL > 1
    const int ch = threadIdx.x % a.cxb, rs = threadIdx.x / a.cxb;
    const int c0 = (blockIdx.x * a.cxb + ch) * VE;
    const bool live = rs < a.RL && c0 < a.C;
    const int qb = a.mode == 0 ? (live ? (c0 / a.secN) * a.r : 0) : a.qbase;
    const int tile = a.cxb * VE * NQ;
    if (a.RL > 1)
        for (int i = threadIdx.x; i < tile; i += 256) red[i] = 0.f;
    float acc[VE][NQ];
#pragma unroll
    for (int e = 0; e < VE; ++e)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[e][q] = 0.f;
    const int m_beg = blockIdx.y * a.rpb, m_end = min(a.M, m_beg + a.rpb);
    if (live) {
        const T* p = reinterpret_cast<const T*>(a.P) + c0;
        const float* qp = a.Q + qb;
        auto fma_row = [&](const vec_t& pv, const float* qr) {
            float qv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(qr + q);
                qv[q] = t[0]; qv[q + 1] = t[1]; qv[q + 2] = t[2]; qv[q + 3] = t[3];
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float x = (float)pv[e];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[e][q] += x * qv[q];
            }
        };
        int m = m_beg + rs;
        const int step = a.RL;
        for (; m + 3 * step < m_end; m += 4 * step) {
            const vec_t p0 = *reinterpret_cast<const vec_t*>(p + (size_t)m * a.ldp);
            const vec_t p1 = *reinterpret_cast<const vec_t*>(p + (size_t)(m + step) * a.ldp);
            const vec_t p2 = *reinterpret_cast<const vec_t*>(p + (size_t)(m + 2 * step) * a.ldp);
            const vec_t p3 = *reinterpret_cast<const vec_t*>(p + (size_t)(m + 3 * step) * a.ldp);
            fma_row(p0, qp + (size_t)m * a.ldq);
            fma_row(p1, qp + (size_t)(m + step) * a.ldq);
            fma_row(p2, qp + (size_t)(m + 2 * step) * a.ldq);
            fma_row(p3, qp + (size_t)(m + 3 * step) * a.ldq);
        }
        for (; m < m_end; m += step) fma_row(*reinterpret_cast<const vec_t*>(p + (size_t)m * a.ldp), qp + (size_t)m * a.ldq);
    }
    float* dst = a.partial + ((size_t)blockIdx.y * a.gx + blockIdx.x) * tile;
    if (a.RL == 1) {
        if (threadIdx.x < a.cxb) {
#pragma unroll
            for (int e = 0; e < VE; ++e)
#pragma unroll
                for (int q = 0; q < NQ; q += 4)
                    *reinterpret_cast<f32x4*>(dst + (ch * VE + e) * NQ + q) = f32x4{acc[e][q], acc[e][q + 1], acc[e][q + 2], acc[e][q + 3]};
        }
        return;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int e = 0; e < VE; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) atomicAdd(&red[(ch * VE + e) * NQ + q], acc[e][q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tile; i += 256) dst[i] = red[i];
}
// sums the partial tiles and adds them into the gradient tensors: 16 outputs x 16 row-slab groups per block (the
// partial rows are ~100 dependent-latency loads apart if one thread walks them alone)
__global__ __launch_bounds__(256) void lora_wgrad_reduce_kernel(const LoraWgradArgs a, int VE, int NQ) {
    __shared__ float red[16][17];
    const int tile = a.cxb * VE * NQ;
    const int total = a.gx * tile;
    const int ol = threadIdx.x & 15, yg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + ol;
    float s = 0.f;
    if (i < total) {
        const float* p = a.partial + i;
        const size_t pitch = (size_t)a.gx * tile;
        int y = yg;
        for (; y + 48 < a.gy; y += 64) s += (p[y * pitch] + p[(y + 16) * pitch]) + (p[(y + 32) * pitch] + p[(y + 48) * pitch]);
        for (; y < a.gy; y += 16) s += p[y * pitch];
    }
    red[yg][ol] = s;
    __syncthreads();
    if (threadIdx.x >= 16 || i >= total) return;
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][ol];
    const int bx = i / tile, il = i - bx * tile;
    const int q = il % NQ, cl = il / NQ;
    const int c = bx * a.cxb * VE + cl;
    if (c >= a.C) return;
    s *= a.scale;
    if (a.mode == 0) {
        const int j = c / a.secN;
        if (a.out[j]) a.out[j][(size_t)(c - j * a.secN) * a.r + q] += s;
    } else {
        const int qq = a.qbase + q, j = qq / a.r;
        if (a.out[j]) a.out[j][(size_t)(qq - j * a.r) * a.C + c] += s;
    }
}
// mode 0: dB of the fused modules of one linear (P = dY [M][nmod*secN], Q = z [M][nmod*r]);
// mode 1: dA (P = x [M][k], Q = dz [M][nmod*r]).  out[j] == nullptr: module j has no adapter.
// scratch: lora_wgrad_scratch_bytes(M, C, nmod*r, sizeof(T)) bytes.
template <typename T>
int launch_lora_wgrad(const void* P, int ldp, const float* Q, int ldq, int M, int C, int mode, int r, int nmod, int secN,
                      float* const out[3], float scale, float* scratch, hipStream_t st) {
    constexpr int VE = BVec<T>::N;
    MRISR_REQUIRE(r % 4 == 0 && r >= 4 && r <= 16 && nmod >= 1 && nmod <= 3, "LoRA wgrad: rank 4/8/12/16, <= 3 fused modules");
    MRISR_REQUIRE(C % VE == 0 && ldp % VE == 0 && ldq % 4 == 0 && (mode == 1 || secN % VE == 0) && scratch, "LoRA wgrad alignment");
    LoraWgradArgs a;
    a.P = P; a.ldp = ldp; a.Q = Q; a.ldq = ldq; a.M = M; a.C = C; a.mode = mode; a.r = r; a.nmod = nmod; a.secN = secN;
    for (int j = 0; j < 3; ++j) a.out[j] = j < nmod ? out[j] : nullptr;
    a.scale = scale;
    a.partial = scratch;
    lora_wgrad_geom(a, VE);
    const dim3 grid(a.gx, a.gy);
    const int R = nmod * r;
    ProfScope ps("lora_wgrad", 2.0 * M * (double)C * (mode ? R : r), (double)M * C * sizeof(T), st);
    auto go = [&](int nq, int qbase) -> int {
        a.qbase = qbase;
        const size_t smem = a.RL > 1 ? (size_t)a.cxb * VE * nq * sizeof(float) : 0;
        MRISR_REQUIRE(smem <= 65536, "LoRA wgrad: LDS tile");
        switch (nq) {
            case 4: hipLaunchKernelGGL((lora_wgrad_kernel<T, 4>), grid, dim3(256), smem, st, a); break;
            case 8: hipLaunchKernelGGL((lora_wgrad_kernel<T, 8>), grid, dim3(256), smem, st, a); break;
            case 12: hipLaunchKernelGGL((lora_wgrad_kernel<T, 12>), grid, dim3(256), smem, st, a); break;
            default: hipLaunchKernelGGL((lora_wgrad_kernel<T, 16>), grid, dim3(256), smem, st, a); break;
        }
        const int total = a.gx * a.cxb * VE * nq;
        hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, st, a, VE, nq);
        return 0;
    };
    if (mode == 0) TRY_(go(r, 0));
    else if (R <= 16) TRY_(go(R, 0));
    else for (int j = 0; j < nmod; ++j) TRY_(go(r, j * r));  // wide adapters: one pass per module
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 2x2 sum pool (backward of nearest x2 upsampling): dst[b][y][x][c] (+)= sum src[b][2y+dy][2x+dx][c]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void sumpool2_kernel(const T* __restrict__ src, T* dst, int B, int H, int W, int C, int accumulate) {
    const long long total = (long long)B * H * W * C;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long p = i / C;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
        const size_t base = (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        float v = to_f32(src[base]) + to_f32(src[base + C]) + to_f32(src[base + (size_t)2 * W * C]) + to_f32(src[base + (size_t)2 * W * C + C]);
        if (accumulate) v += to_f32(dst[i]);
        dst[i] = from_f32<T>(v);
    }
}
template <typename T>
int launch_sumpool2(const void* src, void* dst, int B, int H, int W, int C, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(sumpool2_kernel<T>, dim3(bw_blocks((long long)B * H * W * C)), dim3(256), 0, st, reinterpret_cast<const T*>(src),
                       reinterpret_cast<T*>(dst), B, H, W, C, accumulate);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// loss: MSE(eps_hat, eps) over all elements.  pred NHWC T, target NCHW f32.  Writes dpred (NHWC T) = 2 (pred - tgt)/n and
// accumulates the loss into loss[0] (f32, zeroed by the caller).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mse_grad_kernel(const T* __restrict__ pred, const float* __restrict__ tgt, T* __restrict__ dpred,
                                                       float* loss, int B, int C, int H, int W, float inv_n) {
    __shared__ float red[4];
    const long long total = (long long)B * C * H * W;
    float acc = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long p = i / C;
        const long long hw = (long long)H * W;
        const long long b = p / hw, r = p - b * hw;
        const float d = to_f32(pred[i]) - tgt[(size_t)((b * C + c) * hw + r)];
        acc += d * d;
        dpred[i] = from_f32<T>(2.0f * d * inv_n);
    }
    acc = bw_wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}
template <typename T>
int launch_mse_grad(const void* pred, const float* tgt, void* dpred, float* loss, int B, int C, int H, int W, hipStream_t st) {
    const long long total = (long long)B * C * H * W;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mse_grad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const T*>(pred), tgt,
                       reinterpret_cast<T*>(dpred), loss, B, C, H, W, 1.0f / (float)total);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// optimiser on flat f32 buffers: global grad-norm (for clip 1.0, nb ResDif c11:34) and AdamW (c11:29-33)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += g[i] * g[i];
    acc = bw_wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
int launch_sumsq(const float* g, long long n, float* out, hipStream_t st) {
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, n, out);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// sumsq: device scalar holding the (already all-reduced) sum of squared grads; clip scale = min(1, max_norm/(norm+1e-6))
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, const float* sumsq, float grad_scale, float max_norm, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2) {
    float clip = 1.0f;
    if (max_norm > 0.f) {
        const float norm = sqrtf(sumsq[0]) * grad_scale;
        clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float gs = grad_scale * clip;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float pi = p[i] * (1.f - lr * wd);  // decoupled weight decay (torch.optim.AdamW)
        pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = pi;
    }
}
int launch_adamw(float* p, const float* g, float* m, float* v, long long n, const float* sumsq, float grad_scale, float max_norm,
                 float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(bw_blocks(n)), dim3(256), 0, st, p, g, m, v, n, sumsq, grad_scale, max_norm, lr, b1, b2, eps,
                       wd, bc1, bc2);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// convolution weight gradients (T2I-Adapter training).  dW[co][ci][tap] = sum_m dY[m][co] * X[pix(m, tap)][ci] is run as one
// GEMM per tap with the pixel index m as the contraction: both operands are needed "pixel-contiguous", i.e. transposed.
//   im2col_tap_T: Xt[ci][m] = X[b][oy*stride + ky - pad][ox*stride + kx - pad][ci] (0 outside the image, 0 for m >= M)
//   (dY^T comes from launch_transpose);  the GEMM writes f32 [Cout][Cin]; wgrad_accum folds it into the flat gradient in
//   PyTorch's [Cout][Cin][kh][kw] layout; colsum gives the bias gradient.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_tap_T_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C, int Ho,
                                                           int Wo, int stride, int pad, int ky, int kx, int M, int Mpad) {
    __shared__ T tile[32][33];
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, c = c0 + tx;
        T v = from_f32<T>(0.f);
        if (m < M && c < C) {
            const int hw = Ho * Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / Wo, ox = r - oy * Wo;
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * C + c];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, m = m0 + tx;
        if (c < C && m < Mpad) out[(size_t)c * Mpad + m] = tile[tx][i];
    }
}
// every tap at once (blockIdx.z = tap): out[(tap * C + c)][m] - the K-major operand of ONE pixel-contraction GEMM per conv
template <typename T>
__global__ __launch_bounds__(256) void im2col_all_T_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C, int Ho,
                                                           int Wo, int stride, int pad, int ks, int M, int Mpad) {
    __shared__ T tile[32][33];
    const int tap = blockIdx.z, ky = tap / ks, kx = tap - ky * ks;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, c = c0 + tx;
        T v = from_f32<T>(0.f);
        if (m < M && c < C) {
            const int hw = Ho * Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / Wo, ox = r - oy * Wo;
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * C + c];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, m = m0 + tx;
        if (c < C && m < Mpad) out[((size_t)tap * C + c) * Mpad + m] = tile[tx][i];
    }
}
template <typename T>
int launch_im2col_all_T(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, int ks, int Mpad,
                        hipStream_t st) {
    const int M = B * Ho * Wo;
    MRISR_REQUIRE(Mpad >= M, "im2col: padded pixel count");
    ProfScope ps("im2col_all_T", 0.0, (1.0 + ks * ks) * M * (double)C * sizeof(T), st);
    hipLaunchKernelGGL(im2col_all_T_kernel<T>, dim3((Mpad + 31) / 32, (C + 31) / 32, ks * ks), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                       reinterpret_cast<T*>(out), B, H, W, C, Ho, Wo, stride, pad, ks, M, Mpad);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
template <typename T>
int launch_im2col_tap_T(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, int ky, int kx,
                        int Mpad, hipStream_t st) {
    const int M = B * Ho * Wo;
    MRISR_REQUIRE(Mpad >= M, "im2col: padded pixel count");
    ProfScope ps("im2col_tap_T", 0.0, 2.0 * M * (double)C * sizeof(T), st);
    hipLaunchKernelGGL(im2col_tap_T_kernel<T>, dim3((Mpad + 31) / 32, (C + 31) / 32), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                       reinterpret_cast<T*>(out), B, H, W, C, Ho, Wo, stride, pad, ky, kx, M, Mpad);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// out = h > 0 ? dy : 0   (backward of ReLU; h is the ReLU's OUTPUT; out may alias dy)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h, T* out, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = to_f32(h[i]) > 0.f ? dy[i] : from_f32<T>(0.f);
}
template <typename T>
int launch_relu_bwd(const void* dy, const void* h, void* out, long long n, hipStream_t st) {
    hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(bw_blocks(n)), dim3(256), 0, st, reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(h),
                       reinterpret_cast<T*>(out), n);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// out[c] += sum_m dy[m][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* out, int M, int C, int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int m_beg = blockIdx.y * rows_per_block, m_end = min(M, m_beg + rows_per_block);
    float acc = 0.f;
    if (c < C)
        for (int m = m_beg + sub; m < m_end; m += 4) acc += to_f32(dy[(size_t)m * C + c]);
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <typename T>
int launch_colsum(const void* dy, float* out, int M, int C, hipStream_t st) {
    const int rpb = 512;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3((C + 63) / 64, (M + rpb - 1) / rpb), dim3(256), 0, st, reinterpret_cast<const T*>(dy), out, M, C, rpb);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// gw[i * taps + tap] += tmp[i]   (tmp: this tap's [Cout][Cin] product)
__global__ void wgrad_accum_kernel(const float* __restrict__ tmp, float* gw, long long n, int taps, int tap) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) gw[i * taps + tap] += tmp[i];
}
// gw[(co * Cin + ci) * taps + tap] += tmp[co][tap * Cin + ci]   (tmp: the all-taps product [Cout][taps * Cin])
__global__ void wgrad_accum_all_kernel(const float* __restrict__ tmp, float* gw, int Cout, int Cin, int taps) {
    const long long n = (long long)Cout * Cin * taps;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {  // i walks gw (coalesced writes)
        const int tap = (int)(i % taps);
        const long long cc = i / taps;
        const int ci = (int)(cc % Cin);
        const long long co = cc / Cin;
        gw[i] += tmp[(co * taps + tap) * Cin + ci];
    }
}
int launch_wgrad_accum_all(const float* tmp, float* gw, int Cout, int Cin, int taps, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_accum_all_kernel, dim3(bw_blocks((long long)Cout * Cin * taps)), dim3(256), 0, st, tmp, gw, Cout, Cin, taps);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// ---- full-parameter training (ControlNet): the same two reductions with a source pitch, a source channel pitch (zero-padded layers)
// and the GEGLU row interleave of `ff.net.0.proj` (packed row p of raw row r = g * half + j:  p = (j >> 4) * 32 + (j & 15) + 16 g)
__device__ __forceinline__ int geglu_src(int r, int half) {
    if (half <= 0) return r;
    const int g = r >= half, j = g ? r - half : r;
    return (j >> 4) * 32 + (j & 15) + (g ? 16 : 0);
}
// gw[(co * Cin + ci) * taps + tap] += tmp[src(co) * ld_tmp + tap * cin_src + ci]
__global__ void wgrad_accum_gen_kernel(const float* __restrict__ tmp, int ld_tmp, int cin_src, float* gw, int Cout, int Cin, int taps, int half) {
    const long long n = (long long)Cout * Cin * taps;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int tap = (int)(i % taps);
        const long long cc = i / taps;
        const int ci = (int)(cc % Cin);
        const int co = (int)(cc / Cin);
        gw[i] += tmp[(size_t)geglu_src(co, half) * ld_tmp + (size_t)tap * cin_src + ci];
    }
}
int launch_wgrad_accum_gen(const float* tmp, int ld_tmp, int cin_src, float* gw, int Cout, int Cin, int taps, int geglu_half, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_accum_gen_kernel, dim3(bw_blocks((long long)Cout * Cin * taps)), dim3(256), 0, st, tmp, ld_tmp, cin_src, gw, Cout, Cin, taps, geglu_half);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// out[c] += sum_m dy[m * ld + col0 + src(c)]
template <typename T>
__global__ __launch_bounds__(256) void colsum_gen_kernel(const T* __restrict__ dy, int ld, int col0, float* out, int M, int C, int rows_per_block, int half) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int m_beg = blockIdx.y * rows_per_block, m_end = min(M, m_beg + rows_per_block);
    float acc = 0.f;
    if (c < C) {
        const int sc = col0 + geglu_src(c, half);
        for (int m = m_beg + sub; m < m_end; m += 4) acc += to_f32(dy[(size_t)m * ld + sc]);
    }
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <typename T>
int launch_colsum_gen(const void* dy, int ld, int col0, float* out, int M, int C, int geglu_half, hipStream_t st) {
    const int rpb = 512;
    hipLaunchKernelGGL(colsum_gen_kernel<T>, dim3((C + 63) / 64, (M + rpb - 1) / rpb), dim3(256), 0, st, reinterpret_cast<const T*>(dy), ld, col0, out, M, C, rpb, geglu_half);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
template int launch_colsum_gen<float>(const void*, int, int, float*, int, int, int, hipStream_t);
template int launch_colsum_gen<bf16>(const void*, int, int, float*, int, int, int, hipStream_t);
// ---- affine parameters of the norms, time-embedding path (full-parameter training) ------------------------------------------------
// GroupNorm(+SiLU): g_gamma[c] += sum_{b,hw} dpre xhat, g_beta[c] += sum dpre, dpre = dy (* silu'(gamma xhat + beta)); mean / rstd from the
// forward's partial sums.  grid (B, ceil(C / 64)); 64 channels x 4 row lanes per block; one atomic per channel and block.
template <typename T>
__global__ __launch_bounds__(256) void gn_affine_grad_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ fwd_partial, int nsplit,
                                                             int groups, int HW, int C, float eps, int silu, float* g_gamma, float* g_beta) {
    __shared__ float red[2][4][64];
    __shared__ float mean_s[64], rstd_s[64];
    const int b = blockIdx.x, c0 = blockIdx.y * 64, lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int Cg = C / groups, c = c0 + lane;
    const int g0 = c0 / Cg, g1 = min(groups - 1, (c0 + 63) / Cg);
    if ((int)threadIdx.x <= g1 - g0) {
        const int gq = g0 + threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < nsplit; ++q) {
            const float* p = fwd_partial + (((size_t)b * nsplit + q) * groups + gq) * 2;
            s1 += (double)p[0];
            s2 += (double)p[1];
        }
        const double n = (double)HW * Cg, mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[threadIdx.x] = (float)mean;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    float ag = 0.f, ab = 0.f;
    if (c < C) {
        const int gl = c / Cg - g0;
        const float mu = mean_s[gl], rs = rstd_s[gl], ga = gamma[c], be = beta[c];
        for (int r = sub; r < HW; r += 4) {
            const size_t o = ((size_t)b * HW + r) * C + c;
            const float xh = (to_f32(x[o]) - mu) * rs;
            float d = to_f32(dy[o]);
            if (silu) d *= silu_grad(xh * ga + be);
            ag += d * xh;
            ab += d;
        }
    }
    red[0][sub][lane] = ag;
    red[1][sub][lane] = ab;
    __syncthreads();
    if (sub == 0 && c < C) {
        atomicAdd(g_gamma + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
        atomicAdd(g_beta + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
    }
}
template <typename T>
int launch_gn_affine_grad(const void* x, const void* dy, const float* gamma, const float* beta, const float* fwd_partial, int nsplit, int groups,
                          int B, int HW, int C, float eps, int silu, float* g_gamma, float* g_beta, hipStream_t st) {
    MRISR_REQUIRE(C % groups == 0 && C / groups <= 64 * 64, "GroupNorm affine gradient: group size");
    hipLaunchKernelGGL(gn_affine_grad_kernel<T>, dim3(B, (C + 63) / 64), dim3(256), 0, st, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy),
                       gamma, beta, fwd_partial, nsplit, groups, HW, C, eps, silu, g_gamma, g_beta);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// LayerNorm: g_gamma[c] += sum_rows dy xhat, g_beta[c] += sum_rows dy.  A wave owns a run of rows (statistics by a shuffle tree), a lane the
// channels lane, lane + 64, ... (C <= 64 * 24); one atomic per channel and wave at the end.
template <typename T>
__global__ __launch_bounds__(256) void ln_affine_grad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int M, int C, float eps, int rows_per_wave,
                                                             float* g_gamma, float* g_beta) {
    constexpr int MAXC = 24;
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = wave * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
    float ag[MAXC], ab[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) ag[i] = ab[i] = 0.f;
    const int nc = (C + 63) / 64;
    for (int r = r0; r < r1; ++r) {
        float v[MAXC];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = (i < nc && c < C) ? to_f32(x[(size_t)r * C + c]) : 0.f;
            s1 += v[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
        const float mu = s1 / C;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (i < nc && c < C) { const float d = v[i] - mu; s2 += d * d; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        const float rs = rsqrtf(s2 / C + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (i < nc && c < C) {
                const float d = to_f32(dy[(size_t)r * C + c]);
                ag[i] += d * (v[i] - mu) * rs;
                ab[i] += d;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (i < nc && c < C && r0 < r1) { atomicAdd(g_gamma + c, ag[i]); atomicAdd(g_beta + c, ab[i]); }
    }
}
template <typename T>
int launch_ln_affine_grad(const void* x, const void* dy, int M, int C, float eps, float* g_gamma, float* g_beta, hipStream_t st) {
    MRISR_REQUIRE(C <= 64 * 24, "LayerNorm affine gradient: row width");
    const int rpw = 32, waves = (M + rpw - 1) / rpw;
    hipLaunchKernelGGL(ln_affine_grad_kernel<T>, dim3((waves + 3) / 4), dim3(256), 0, st, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), M, C, eps,
                       rpw, g_gamma, g_beta);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// out[b][off + c] += sum_hw dh[b][hw][c]  (the time-embedding projection enters a ResnetBlock as a per-image row vector); div = HW for
// per-sample timesteps, one row for a scalar timestep (rows = 1: every image adds into row 0)
template <typename T>
__global__ __launch_bounds__(256) void rowvec_grad_kernel(const T* __restrict__ dh, float* out, int ld_out, int off, int HW, int C, int scalar_t) {
    __shared__ float red[4][64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int r = sub; r < HW; r += 4) acc += to_f32(dh[((size_t)b * HW + r) * C + c]);
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && c < C) atomicAdd(out + (size_t)(scalar_t ? 0 : b) * ld_out + off + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <typename T>
int launch_rowvec_grad(const void* dh, float* out, int ld_out, int off, int B, int HW, int C, int scalar_t, hipStream_t st) {
    hipLaunchKernelGGL(rowvec_grad_kernel<T>, dim3(B, (C + 63) / 64), dim3(256), 0, st, reinterpret_cast<const T*>(dh), out, ld_out, off, HW, C, scalar_t);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// tiny dense layers of the time-embedding MLP (rows <= 64), f32 activations, weights of type T:
//   small_wgrad:  gW[n][k] += sum_r dY[r][n] act(X[r][k]),  gB[n] += sum_r dY[r][n]          (act = SiLU when silu_in)
//   small_dgrad:  dX[r][k]  = (sum_n dY[r][n] W[n][k]) * (pre ? silu'(pre[r][k]) : 1)
__global__ void small_wgrad_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, int rows, int N, int K, int silu_in,
                                   float* gW, float* gB) {
    const long long total = (long long)N * K;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long long)n * K);
        float acc = 0.f, accb = 0.f;
        for (int r = 0; r < rows; ++r) {
            float xv = X[(size_t)r * ldx + k];
            if (silu_in) xv = silu_f(xv);
            const float d = dY[(size_t)r * ldy + n];
            acc += d * xv;
            accb += d;
        }
        gW[i] += acc;
        if (k == 0 && gB) gB[n] += accb;
    }
}
template <typename T>
__global__ void small_dgrad_kernel(const float* __restrict__ dY, int ldy, const T* __restrict__ W, int rows, int N, int K, const float* __restrict__ pre,
                                   int ldpre, float* dX, int ldx) {
    const long long total = (long long)rows * K;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += dY[(size_t)r * ldy + n] * to_f32(W[(size_t)n * K + k]);
        if (pre) acc *= silu_grad(pre[(size_t)r * ldpre + k]);
        dX[(size_t)r * ldx + k] = acc;
    }
}
int launch_small_wgrad(const float* dY, int ldy, const float* X, int ldx, int rows, int N, int K, int silu_in, float* gW, float* gB, hipStream_t st) {
    hipLaunchKernelGGL(small_wgrad_kernel, dim3(bw_blocks((long long)N * K)), dim3(256), 0, st, dY, ldy, X, ldx, rows, N, K, silu_in, gW, gB);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
template <typename T>
int launch_small_dgrad(const float* dY, int ldy, const void* W, int rows, int N, int K, const float* pre, int ldpre, float* dX, int ldx, hipStream_t st) {
    hipLaunchKernelGGL(small_dgrad_kernel<T>, dim3(bw_blocks((long long)rows * K)), dim3(256), 0, st, dY, ldy, reinterpret_cast<const T*>(W), rows, N, K, pre,
                       ldpre, dX, ldx);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
#define AFF_INST(T)                                                                                                                                              \
    template int launch_gn_affine_grad<T>(const void*, const void*, const float*, const float*, const float*, int, int, int, int, int, float, int, float*, float*, \
                                          hipStream_t);                                                                                                          \
    template int launch_ln_affine_grad<T>(const void*, const void*, int, int, float, float*, float*, hipStream_t);                                               \
    template int launch_rowvec_grad<T>(const void*, float*, int, int, int, int, int, int, hipStream_t);                                                          \
    template int launch_small_dgrad<T>(const float*, int, const void*, int, int, int, const float*, int, float*, int, hipStream_t);
AFF_INST(float)
AFF_INST(bf16)
#undef AFF_INST
// SiLU as its own pass (the ControlNet condition embedding in training keeps the pre-activations), and its backward
template <typename T>
__global__ void silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = from_f32<T>(silu_f(to_f32(x[i])));
}
template <typename T>
__global__ void silu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dx[i] = from_f32<T>(to_f32(dy[i]) * silu_grad(to_f32(pre[i])));
}
template <typename T>
int launch_silu_fwd(const void* x, void* y, long long n, hipStream_t st) {
    hipLaunchKernelGGL(silu_fwd_kernel<T>, dim3(bw_blocks(n)), dim3(256), 0, st, reinterpret_cast<const T*>(x), reinterpret_cast<T*>(y), n);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
template <typename T>
int launch_silu_bwd(const void* dy, const void* pre, void* dx, long long n, hipStream_t st) {
    hipLaunchKernelGGL(silu_bwd_kernel<T>, dim3(bw_blocks(n)), dim3(256), 0, st, reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(pre), reinterpret_cast<T*>(dx), n);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// dgrad bank of a zero-padded 3x3 conv: wd[ci][ky][kx][co] over the PADDED channel counts, zero outside the raw tensor
template <typename T>
__global__ void pack_conv_dgrad_padded_kernel(const float* __restrict__ w, T* __restrict__ wd, int Cout, int Cin, int Cout_p, int Cin_p) {
    const long long total = (long long)Cout_p * Cin_p * 9;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout_p);
        const int t = (int)((i / Cout_p) % 9);
        const int ci = (int)(i / (9ll * Cout_p));
        const int ky = t / 3, kx = t - ky * 3;
        wd[i] = from_f32<T>(co < Cout && ci < Cin ? w[(((size_t)co * Cin + ci) * 3 + (2 - ky)) * 3 + (2 - kx)] : 0.f);
    }
}
template <typename T>
int launch_pack_conv_dgrad_padded(const float* w, void* wd, int Cout, int Cin, int Cout_p, int Cin_p, hipStream_t st) {
    hipLaunchKernelGGL(pack_conv_dgrad_padded_kernel<T>, dim3(bw_blocks((long long)Cout_p * Cin_p * 9)), dim3(256), 0, st, w, reinterpret_cast<T*>(wd), Cout, Cin, Cout_p, Cin_p);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
#define SILU_INST(T)                                                                    \
    template int launch_silu_fwd<T>(const void*, void*, long long, hipStream_t);        \
    template int launch_silu_bwd<T>(const void*, const void*, void*, long long, hipStream_t); \
    template int launch_pack_conv_dgrad_padded<T>(const float*, void*, int, int, int, int, hipStream_t);
SILU_INST(float)
SILU_INST(bf16)
#undef SILU_INST
int launch_wgrad_accum(const float* tmp, float* gw, long long n, int taps, int tap, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_accum_kernel, dim3(bw_blocks(n)), dim3(256), 0, st, tmp, gw, n, taps, tap);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// dgrad filter bank of a 3x3 conv: wd[ci][ky][kx][co] = w[co][ci][2-ky][2-kx]  (w: PyTorch f32 [Cout][Cin][3][3])
template <typename T>
__global__ void pack_conv_dgrad2_kernel(const float* __restrict__ w, T* __restrict__ wd, int Cout, int Cin) {
    const long long total = (long long)Cout * Cin * 9;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const int t = (int)((i / Cout) % 9);
        const int ci = (int)(i / (9ll * Cout));
        const int ky = t / 3, kx = t - ky * 3;
        wd[i] = from_f32<T>(w[(((size_t)co * Cin + ci) * 3 + (2 - ky)) * 3 + (2 - kx)]);
    }
}
template <typename T>
int launch_pack_conv_dgrad(const float* w, void* wd, int Cout, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(pack_conv_dgrad2_kernel<T>, dim3(bw_blocks((long long)Cout * Cin * 9)), dim3(256), 0, st, w, reinterpret_cast<T*>(wd), Cout, Cin);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// exponential moving average of the trainable vector (diffusers EMAModel.step): ema = decay * ema + (1 - decay) * theta
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ theta, long long n, float decay) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ema[i] = decay * ema[i] + (1.f - decay) * theta[i];
}
int launch_ema(float* ema, const float* theta, long long n, float decay, hipStream_t st) {
    hipLaunchKernelGGL(ema_kernel, dim3(bw_blocks(n)), dim3(256), 0, st, ema, theta, n, decay);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

#define BWD_INST(T)                                                                                                         \
    template int launch_groupnorm_bwd<T>(const GroupNormBwdArgs&, hipStream_t);                                             \
    template int launch_layernorm_bwd<T>(const void*, const void*, void*, const float*, int, int, float, int, hipStream_t); \
    template int launch_geglu_bwd<T>(const void*, const void*, void*, long long, int, hipStream_t);                         \
    template int launch_geglu_fwd<T>(const void*, void*, long long, int, hipStream_t);                                      \
    template int launch_rows_to_heads<T>(const void*, int, int, void*, int, int, int, int, int, int, hipStream_t);         \
    template int launch_softmax_bwd<T>(const void*, const float*, void*, int, long long, int, float, hipStream_t);          \
    template int launch_transpose<T>(const void*, void*, int, int, int, int, long long, long long, int, int, hipStream_t);  \
    template int launch_lora_wgrad<T>(const void*, int, const float*, int, int, int, int, int, int, int, float* const[3],   \
                                      float, float*, hipStream_t);                                                          \
    template int launch_sumpool2<T>(const void*, void*, int, int, int, int, int, hipStream_t);                              \
    template int launch_mse_grad<T>(const void*, const float*, void*, float*, int, int, int, int, hipStream_t);                  \
    template int launch_im2col_tap_T<T>(const void*, void*, int, int, int, int, int, int, int, int, int, int, int, hipStream_t); \
    template int launch_im2col_all_T<T>(const void*, void*, int, int, int, int, int, int, int, int, int, int, hipStream_t);                    \
    template int launch_relu_bwd<T>(const void*, const void*, void*, long long, hipStream_t);                                   \
    template int launch_colsum<T>(const void*, float*, int, int, hipStream_t);                                                   \
    template int launch_pack_conv_dgrad<T>(const float*, void*, int, int, hipStream_t);
BWD_INST(float)
BWD_INST(bf16)

}  // namespace mrisr
