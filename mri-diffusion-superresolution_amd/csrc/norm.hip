// GroupNorm(+SiLU) over NHWC activations (with the skip-concat of two sources folded into the read),
// LayerNorm over token rows, and the row softmax of the f32 parity attention.  All HBM-bound: 16-byte
// vector accesses, f32 statistics, deterministic fixed-order reductions (no float atomics).
#include "common.h"
#include "prof.h"

namespace mrisr {

template <typename T> struct Vec;  // 16-byte vector of T
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    typedef bf16x8 type;
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4 type;
};

// ------------------------------------------------------------------------------------------------
// GroupNorm stage 1: per (sample, row-split) partial sums per group.
//   block = (slots = C/VE/VPT) x RL threads; a thread owns VPT 16-byte channel vectors of every
//   RL-th row of its split, so its accumulators are per-channel; the block then folds rows and
//   channels-of-a-group through LDS in a fixed order.
// ------------------------------------------------------------------------------------------------
template <typename T, int VPT>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GroupNormArgs a, int slots, int RL) {
    constexpr int VE = Vec<T>::N;
    typedef typename Vec<T>::type vec_t;
    extern __shared__ float sm[];  // [2][RL][C]
    const int C = a.c0 + a.c1;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int tid = threadIdx.x;
    const int slot = tid % slots, rl = tid / slots;
    const int rows_per = (a.HW + a.nsplit - 1) / a.nsplit;
    const int r_beg = sp * rows_per, r_end = min(a.HW, r_beg + rows_per);
    float s1[VPT][VE], s2[VPT][VE];
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int e = 0; e < VE; ++e) s1[v][e] = s2[v][e] = 0.f;
    if (rl < RL) {
        constexpr int UR = 4;  // rows in flight per thread (the pass is otherwise latency-, not bandwidth-bound)
        for (int r0 = r_beg + rl; r0 < r_end; r0 += RL * UR) {
            vec_t xv[UR][VPT];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * RL;
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int ch = (slot * VPT + v) * VE;
                    const T* src = ch < a.c0
                                       ? reinterpret_cast<const T*>(a.x0) + ((size_t)b * a.HW + r) * a.c0 + ch
                                       : reinterpret_cast<const T*>(a.x1) + ((size_t)b * a.HW + r) * a.c1 + (ch - a.c0);
                    if (r < r_end) xv[u][v] = *reinterpret_cast<const vec_t*>(src);
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (r0 + u * RL >= r_end) break;
#pragma unroll
                for (int v = 0; v < VPT; ++v)
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const float f = (float)xv[u][v][e];
                        s1[v][e] += f;
                        s2[v][e] += f * f;
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
    const int Cg = C / a.groups;
    if (tid < 2 * a.groups) {
        const int gq = tid % a.groups, which = tid / a.groups;
        double acc = 0.0;
        for (int r = 0; r < RL; ++r) {
            const float* row = sm + (size_t)(which * RL + r) * C + gq * Cg;
            float t = 0.f;
            for (int c = 0; c < Cg; ++c) t += row[c];
            acc += (double)t;
        }
        a.partial[(((size_t)b * a.nsplit + sp) * a.groups + gq) * 2 + which] = (float)acc;
    }
}

// GroupNorm stage 2: y = x * scale[c] + shift[c]  [SiLU];  concat folded into the read.
//   Same thread geometry as stage 1: a thread owns VPT fixed 16-byte channel vectors, so
//   scale = rstd*gamma and shift = beta - mean*rstd*gamma live in registers and the row loop is
//   load - 8 FMA (+SiLU) - store with no integer division.
template <typename T, int VPT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GroupNormArgs a, int slots, int RL, int rows_per_block) {
    constexpr int VE = Vec<T>::N;
    typedef typename Vec<T>::type vec_t;
    __shared__ float mean_s[64], rstd_s[64];
    const int C = a.c0 + a.c1;
    const int Cg = C / a.groups;
    const int b = blockIdx.y;
    if (threadIdx.x < a.groups) {
        double s1 = 0.0, s2 = 0.0;
        for (int sp = 0; sp < a.nsplit; ++sp) {
            const float* p = a.partial + (((size_t)b * a.nsplit + sp) * a.groups + threadIdx.x) * 2;
            s1 += (double)p[0];
            s2 += (double)p[1];
        }
        const double n = (double)a.HW * Cg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[threadIdx.x] = (float)mean;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const int slot = tid % slots, rl = tid / slots;
    if (rl >= RL) return;
    float sc[VPT][VE], sh[VPT][VE];
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int c = (slot * VPT + v) * VE + e;
            const int gq = c / Cg;
            const float s = rstd_s[gq] * a.gamma[c];
            sc[v][e] = s;
            sh[v][e] = a.beta[c] - mean_s[gq] * s;
        }
    const int r_beg = blockIdx.x * rows_per_block;
    const int r_end = min(a.HW, r_beg + rows_per_block);
    T* y = reinterpret_cast<T*>(a.y) + (size_t)b * a.HW * C;
    // UR rows per iteration: all their loads are issued before the first use (a thread otherwise has one 16-byte load in
    // flight at a time and the pass runs at latency, not bandwidth)
    constexpr int UR = 4;
    for (int r0 = r_beg + rl; r0 < r_end; r0 += RL * UR) {
        vec_t xv[UR][VPT];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int r = r0 + u * RL;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int ch = (slot * VPT + v) * VE;
                const T* src = ch < a.c0 ? reinterpret_cast<const T*>(a.x0) + ((size_t)b * a.HW + r) * a.c0 + ch
                                         : reinterpret_cast<const T*>(a.x1) + ((size_t)b * a.HW + r) * a.c1 + (ch - a.c0);
                if (r < r_end) xv[u][v] = *reinterpret_cast<const vec_t*>(src);
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int r = r0 + u * RL;
            if (r >= r_end) break;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int ch = (slot * VPT + v) * VE;
                vec_t o;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    float f = (float)xv[u][v][e] * sc[v][e] + sh[v][e];
                    if (a.silu) f = silu_f(f);
                    o[e] = from_f32<T>(f);
                }
                *reinterpret_cast<vec_t*>(y + (size_t)r * C + ch) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm in ONE pass (bf16): a workgroup owns one image and a slab of whole groups (slab = a multiple of lcm(8, C/groups)
// channels) and keeps its part of the image in registers - NV 16-byte vectors per thread, all loaded before the first use -
// so the activation is read once instead of twice and the statistics never leave the CU.  Per-channel sums are folded
// through LDS in a fixed order (deterministic), groups are formed from channel sums (a 16-byte vector may straddle two
// groups: C/groups = 10, 20, 60 ...).  The (sum, sum of squares) pairs also go to the workspace in the two-kernel layout
// (split 0; the other splits zero) for the backward pass.  All slabs of an image run on one XCD (4 MB L2 absorbs the
// partial cache lines of the 80...240-byte slab rows); falls back to the two-kernel path when a slab does not fit.
// ------------------------------------------------------------------------------------------------
template <int NVM, int VE, bool SLAB = false, int NT = 256>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const GroupNormArgs a, int slab, int slots, int RL, int nslab, int xcd_map, const GnSlabSrc ss = GnSlabSrc()) {
    typedef bf16 T;
    typedef __attribute__((ext_vector_type(VE))) __bf16 vec_t;  // 16-byte (VE = 8) or 8-byte (VE = 4) channel vectors
    __shared__ float part[NT * 16];   // [tid][s1[VE] at 0 | s2[VE] at 8]
    __shared__ float seg[2 * 1280];    // stage A of the channel fold: [segment][2 * slab]
    __shared__ double gsum[2 * 80];    // [which][group of the slab]
    __shared__ float mean_s[80], rstd_s[80];
    const int C = a.c0 + a.c1;
    const int Cg = C / a.groups;
    int b, sl;
    if (xcd_map) {  // linear id L: XCD = L % 8 = image % 8  ->  every slab of an image lands on the same XCD
        const int L = blockIdx.x;
        b = (L & 7) + 8 * (L / (8 * nslab));
        sl = (L >> 3) % nslab;
    } else {
        b = blockIdx.x / nslab;
        sl = blockIdx.x - b * nslab;
    }
    const int ch0 = sl * slab;
    const int tid = threadIdx.x;
    const int slot = tid % slots, rl = tid / slots;
    const bool active = rl < RL;
    const int ch = ch0 + slot * VE;
    const T* src;
    int ld;
    if (ch < a.c0) { src = reinterpret_cast<const T*>(a.x0) + (size_t)b * a.HW * a.c0 + ch; ld = a.c0; }
    else { src = reinterpret_cast<const T*>(a.x1) + (size_t)b * a.HW * a.c1 + (ch - a.c0); ld = a.c1; }
    vec_t xv[NVM];
    float s1[VE], s2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
        if constexpr (SLAB) {
            // the input does not exist yet: it is the split-K GEMM's output, summed here from the f32 slabs with splitk_reduce_kernel's
            // operations in its order (sum over splits, * alpha + bias, + time-embedding row, + residual), rounded to bf16 as it would have been
            typedef __attribute__((ext_vector_type(VE))) float fvec_t;
            const size_t MN = (size_t)a.B * a.HW * C;
            fvec_t bv;
#pragma unroll
            for (int e = 0; e < VE; ++e) bv[e] = 0.f;
            if (ss.bias) bv = *reinterpret_cast<const fvec_t*>(ss.bias + ch);
#pragma unroll
            for (int i = 0; i < NVM; ++i) {
                const int r = rl + i * RL;
                if (r < a.HW) {
                    const size_t mrow = (size_t)b * a.HW + r;
                    const float* pp = ss.partial + mrow * C + ch;
                    fvec_t v;
#pragma unroll
                    for (int e = 0; e < VE; ++e) v[e] = 0.f;
                    for (int sp = 0; sp < ss.splitk; ++sp) {
                        const fvec_t t = *reinterpret_cast<const fvec_t*>(pp + (size_t)sp * MN);
#pragma unroll
                        for (int e = 0; e < VE; ++e) v[e] += t[e];
                    }
#pragma unroll
                    for (int e = 0; e < VE; ++e) v[e] = v[e] * ss.alpha + bv[e];
                    if (ss.rowvec) {
                        const fvec_t t = *reinterpret_cast<const fvec_t*>(ss.rowvec + (mrow / ss.rowvec_div) * ss.rowvec_ld + ch);
#pragma unroll
                        for (int e = 0; e < VE; ++e) v[e] += t[e];
                    }
                    if (ss.resid) {
                        const vec_t t = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(ss.resid) + mrow * ss.ldr + ch);
#pragma unroll
                        for (int e = 0; e < VE; ++e) v[e] += (float)t[e];
                    }
#pragma unroll
                    for (int e = 0; e < VE; ++e) xv[i][e] = (bf16)v[e];
                    if (ss.raw_out) *reinterpret_cast<vec_t*>(reinterpret_cast<T*>(ss.raw_out) + mrow * C + ch) = xv[i];
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < NVM; ++i) {
            const int r = rl + i * RL;
            if (r < a.HW) xv[i] = *reinterpret_cast<const vec_t*>(src + (size_t)r * ld);
        }
        }
#pragma unroll
        for (int i = 0; i < NVM; ++i) {
            if (rl + i * RL < a.HW) {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float f = (float)xv[i][e];
                    s1[e] += f;
                    s2[e] += f * f;
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
    // channel fold, stage A: output o = which * slab + c, rows split into nseg segments so that all threads work
    const int nout = 2 * slab;
    int nseg = NT / nout;
    if (nseg < 1) nseg = 1;
    if (nseg > RL) nseg = RL;
    const int seg_len = (RL + nseg - 1) / nseg;
    for (int t = tid; t < nout * nseg; t += NT) {
        const int o = t % nout, sg = t / nout;
        const int which = o / slab, c = o - which * slab;
        const int off = (c / VE) * 16 + which * 8 + (c % VE);
        float acc = 0.f;
        const int r1 = min(RL, (sg + 1) * seg_len);
        for (int r = sg * seg_len; r < r1; ++r) acc += part[(r * slots) * 16 + off];
        seg[sg * nout + o] = acc;
    }
    __syncthreads();
    // stage B: a wave per (group, which) pair - lanes take the group's channels x segments, then a shuffle tree
    const int ng = slab / Cg;
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int p = wave; p < 2 * ng; p += NT / 64) {
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
        const double a1 = gsum[tid], a2 = gsum[80 + tid];
        const double n = (double)a.HW * Cg;
        const double mean = a1 / n;
        double var = a2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        const int gq = ch0 / Cg + tid;
        for (int sp = 0; sp < a.nsplit; ++sp) {
            float* p = a.partial + (((size_t)b * a.nsplit + sp) * a.groups + gq) * 2;
            p[0] = sp == 0 ? (float)a1 : 0.f;
            p[1] = sp == 0 ? (float)a2 : 0.f;
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int gl = (slot * VE + e) / Cg;
        const float sv = rstd_s[gl] * a.gamma[ch + e];
        sc[e] = sv;
        sh[e] = a.beta[ch + e] - mean_s[gl] * sv;
    }
    T* y = reinterpret_cast<T*>(a.y) + (size_t)b * a.HW * C + ch;
#pragma unroll
    for (int i = 0; i < NVM; ++i) {
        const int r = rl + i * RL;
        if (r < a.HW) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float f = (float)xv[i][e] * sc[e] + sh[e];
                if (a.silu) f = silu_f(f);
                o[e] = (bf16)f;
            }
            *reinterpret_cast<vec_t*>(y + (size_t)r * C) = o;
        }
    }
}

static int g_gn_fused = [] { const char* e = getenv("MRISR_GN_FUSED"); return e ? atoi(e) : 1; }();
extern "C" void mrisr_debug_gn_fused(int on) { g_gn_fused = on; }

// slab / thread geometry of the one-pass kernels for this shape; false: use the two-kernel path
bool gn_fused_geometry(int c0, int c1, int groups, int HW, int* slab_out, int* slots_out, int* rl_out, int* nv_out, int ve) {
    if (!g_gn_fused) return false;
    struct { int c0, c1, groups, HW; } a = {c0, c1, groups, HW};
    const int C = a.c0 + a.c1, Cg = C / a.groups;
    int base = Cg;
    while (base % ve) base += Cg;  // lcm(ve, Cg)
    if (C % base) return false;
    int best = 0, best_active = 0;
    for (int f = 1; f * base <= 640 && f * base <= C; ++f) {
        const int slab = f * base;
        if (C % slab || a.c0 % ve || a.c1 % ve) continue;
        const int slots = slab / ve;
        if (slots > 256 || slab / Cg > 80) continue;
        const int RL = min(256 / slots, a.HW);
        const int nv = (a.HW + RL - 1) / RL;
        if (nv > 24) continue;
        const int active = slots * RL;
        if (active > best_active) { best = slab; best_active = active; }
        if (active >= 200) break;  // the smallest slab that fills the workgroup: most workgroups
    }
    if (!best) return false;
    *slab_out = best;
    *slots_out = best / ve;
    *rl_out = min(256 / *slots_out, a.HW);
    *nv_out = (a.HW + *rl_out - 1) / *rl_out;
    return true;
}

int groupnorm_nsplit(int B, int HW) {
    int ns = 512 / (B > 0 ? B : 1);
    if (ns < 1) ns = 1;
    if (ns > HW / 8) ns = HW / 8;
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    return ns;
}

template <typename T>
int launch_groupnorm(const GroupNormArgs& a, hipStream_t st) {
    constexpr int VE = Vec<T>::N;
    const int C = a.c0 + a.c1;
    MRISR_REQUIRE(C % a.groups == 0 && a.groups <= 64, "GroupNorm groups");
    MRISR_REQUIRE(a.c0 % VE == 0 && a.c1 % VE == 0, "GroupNorm channel alignment");
    MRISR_REQUIRE(a.partial != nullptr && a.nsplit >= 1, "GroupNorm workspace");
    const int nvec = C / VE;
    int vpt = 1;
    while (nvec / vpt > 256 || (nvec % vpt) != 0) ++vpt;
    MRISR_REQUIRE(vpt <= 4, "GroupNorm: too many channels for the stats kernel");
    const int slots = nvec / vpt;
    int RL = 256 / slots;
    if (RL < 1) RL = 1;
    const size_t smem = (size_t)2 * RL * C * sizeof(float);
    dim3 grid(a.nsplit, a.B);
    const double act_bytes = (double)a.B * a.HW * C * sizeof(T);
    std::string n1 = "groupnorm_stats", n2 = "groupnorm_apply";
    if (prof_enabled() && prof_shapes()) {
        char buf[96];
        snprintf(buf, sizeof(buf), " B=%d HW=%d C=%d", a.B, a.HW, C);
        n1 += buf;
        n2 += buf;
    }
    if constexpr (sizeof(T) == 2) {
        // probe (MRISR_GN_WIDE=<slab channels>): 512-thread workgroups holding a WIDER slab (16-byte vectors, e.g. 80 channels = 160 contiguous
        // bytes per row instead of 40) for the 32 x 32 maps - fewer, fatter workgroups with better-coalesced rows
        static const int wide = [] { const char* e = getenv("MRISR_GN_WIDE"); return e ? atoi(e) : 0; }();
        if (wide > 0 && g_gn_fused && a.c1 == 0 && C % wide == 0 && wide % 8 == 0 && wide % (C / a.groups) == 0 && wide / 8 <= 512 && wide / (C / a.groups) <= 80 && a.HW >= 512) {
            const int wslots = wide / 8, wRL = std::min(512 / wslots, a.HW), wnv = (a.HW + wRL - 1) / wRL;
            if (wnv <= 24 && 2 * wide <= 1280) {
                const int nslab = C / wide, xmap = (a.B % 8) == 0 ? 1 : 0;
                ProfScope ps("groupnorm_fused_wide", 0.0, 2.0 * act_bytes, st);
                hipLaunchKernelGGL((gn_fused_kernel<24, 8, false, 512>), dim3(a.B * nslab), dim3(512), 0, st, a, wide, wslots, wRL, nslab, xmap, GnSlabSrc());
                MRISR_CHECK_HIP(hipGetLastError());
                return 0;
            }
        }
        int slab = 0, fslots = 0, fRL = 0, nv = 0;
        if (gn_fused_geometry(a.c0, a.c1, a.groups, a.HW, &slab, &fslots, &fRL, &nv, 8)) {
            // fewer than two workgroups per CU: their load / fold / store phases cannot overlap - try slabs of 8-byte vectors
            // (half the channels per slab, twice the workgroups)
            int ve = 8, s4 = 0, sl4 = 0, rl4 = 0, nv4 = 0;
            static const int ve4_ok = [] { const char* e = getenv("MRISR_GN_VE4"); return e ? atoi(e) : 1; }();
            static const int ve4_blocks = [] { const char* e = getenv("MRISR_GN_VE4_BLOCKS"); return e ? atoi(e) : 512; }();
            if (ve4_ok && a.B * (C / slab) < ve4_blocks && gn_fused_geometry(a.c0, a.c1, a.groups, a.HW, &s4, &sl4, &rl4, &nv4, 4) &&
                a.B * (C / s4) > a.B * (C / slab)) {
                ve = 4; slab = s4; fslots = sl4; fRL = rl4; nv = nv4;
            }
            const int nslab = C / slab;
            const int xmap = (a.B % 8) == 0 ? 1 : 0;
            std::string nf = "groupnorm_fused";
            if (prof_enabled() && prof_shapes()) {
                char buf[96];
                snprintf(buf, sizeof(buf), " B=%d HW=%d C=%d", a.B, a.HW, C);
                nf += buf;
            }
            ProfScope ps(prof_intern(nf), 0.0, 2.0 * act_bytes, st);
            const dim3 fg(a.B * nslab);
#define GN_GO(NV, VEV) hipLaunchKernelGGL((gn_fused_kernel<NV, VEV>), fg, dim3(256), 0, st, a, slab, fslots, fRL, nslab, xmap)
            if (ve == 8) {
                if (nv <= 2) GN_GO(2, 8); else if (nv <= 4) GN_GO(4, 8); else if (nv <= 8) GN_GO(8, 8); else if (nv <= 16) GN_GO(16, 8); else GN_GO(24, 8);
            } else {
                if (nv <= 2) GN_GO(2, 4); else if (nv <= 4) GN_GO(4, 4); else if (nv <= 8) GN_GO(8, 4); else if (nv <= 16) GN_GO(16, 4); else GN_GO(24, 4);
            }
#undef GN_GO
            MRISR_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    {
    ProfScope ps(prof_intern(n1), 0.0, act_bytes, st);
    switch (vpt) {
        case 1: hipLaunchKernelGGL((gn_stats_kernel<T, 1>), grid, dim3(256), smem, st, a, slots, RL); break;
        case 2: hipLaunchKernelGGL((gn_stats_kernel<T, 2>), grid, dim3(256), smem, st, a, slots, RL); break;
        case 3: hipLaunchKernelGGL((gn_stats_kernel<T, 3>), grid, dim3(256), smem, st, a, slots, RL); break;
        default: hipLaunchKernelGGL((gn_stats_kernel<T, 4>), grid, dim3(256), smem, st, a, slots, RL); break;
    }
    }
    MRISR_CHECK_HIP(hipGetLastError());
    // ~2048 blocks over the batch; every block streams a contiguous run of rows
    int bx = 1024 / (a.B > 0 ? a.B : 1) + 1;
    int rows_per_block = (a.HW + bx - 1) / bx;
    if (rows_per_block < RL) rows_per_block = RL;
    bx = (a.HW + rows_per_block - 1) / rows_per_block;
    ProfScope ps2(prof_intern(n2), 0.0, 2.0 * act_bytes, st);
    switch (vpt) {
        case 1: hipLaunchKernelGGL((gn_apply_kernel<T, 1>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        case 2: hipLaunchKernelGGL((gn_apply_kernel<T, 2>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        case 3: hipLaunchKernelGGL((gn_apply_kernel<T, 3>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
        default: hipLaunchKernelGGL((gn_apply_kernel<T, 4>), dim3(bx, a.B), dim3(256), 0, st, a, slots, RL, rows_per_block); break;
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

static int g_gn_slabs = -1;  // test hook: -1 = MRISR_GN_SLABS (default 1), 0 off, 1 on
extern "C" void mrisr_debug_gn_slabs(int on) { g_gn_slabs = on; ++g_plan_salt; }
static bool gn_slabs_geometry(int C, int groups, int HW, int B, int* ve, int* slab, int* slots, int* rl, int* nv) {
    if (!gn_fused_geometry(C, 0, groups, HW, slab, slots, rl, nv, 8)) return false;
    *ve = 8;
    int s4 = 0, sl4 = 0, rl4 = 0, nv4 = 0;
    if (B * (C / *slab) < 512 && gn_fused_geometry(C, 0, groups, HW, &s4, &sl4, &rl4, &nv4, 4) && C / s4 > C / *slab) {
        *ve = 4; *slab = s4; *slots = sl4; *rl = rl4; *nv = nv4;
    }
    return *nv <= 8;  // the slabs are summed into registers: few vectors per thread (the deep levels: 8 x 8 and 4 x 4 maps, 16 x 16 at most)
}
bool groupnorm_slabs_ok(int C, int groups, int HW) {
    static const int env = [] { const char* e = getenv("MRISR_GN_SLABS"); return e ? atoi(e) : 1; }();
    if (g_gn_slabs < 0 ? !env : !g_gn_slabs) return false;
    if (C % groups || groups > 64 || C % 8) return false;
    int ve, slab, slots, rl, nv;
    return gn_slabs_geometry(C, groups, HW, 32, &ve, &slab, &slots, &rl, &nv);
}
int launch_groupnorm_slabs(const GroupNormArgs& a, const GnSlabSrc& s, hipStream_t st) {
    const int C = a.c0;
    MRISR_REQUIRE(a.c1 == 0 && a.x1 == nullptr && C % a.groups == 0 && a.groups <= 64 && C % 8 == 0, "GroupNorm from slabs: one source, whole groups");
    MRISR_REQUIRE(s.partial && s.splitk >= 1 && a.partial != nullptr && a.nsplit >= 1 && (!s.rowvec || s.rowvec_div >= 1) && (!s.resid || s.ldr % 8 == 0), "GroupNorm from slabs: operands");
    int ve, slab, slots, rl, nv;
    MRISR_REQUIRE(gn_slabs_geometry(C, a.groups, a.HW, a.B, &ve, &slab, &slots, &rl, &nv), "GroupNorm from slabs: geometry (groupnorm_slabs_ok)");
    const int nslab = C / slab;
    const int xmap = (a.B % 8) == 0 ? 1 : 0;
    std::string nf = "groupnorm_from_slabs";
    if (prof_enabled() && prof_shapes()) {
        char buf[96];
        snprintf(buf, sizeof(buf), " B=%d HW=%d C=%d s=%d", a.B, a.HW, C, s.splitk);
        nf += buf;
    }
    const double act_bytes = (double)a.B * a.HW * C * 2.0;
    ProfScope ps(prof_intern(nf), 0.0, act_bytes * (2.0 * s.splitk + 1.0 + (s.raw_out ? 1.0 : 0.0) + (s.resid ? 1.0 : 0.0)), st);
    const dim3 fg(a.B * nslab);
#define GNS_GO(NV, VEV) hipLaunchKernelGGL((gn_fused_kernel<NV, VEV, true>), fg, dim3(256), 0, st, a, slab, slots, rl, nslab, xmap, s)
    if (ve == 8) { if (nv <= 2) GNS_GO(2, 8); else if (nv <= 4) GNS_GO(4, 8); else GNS_GO(8, 8); }
    else { if (nv <= 2) GNS_GO(2, 4); else if (nv <= 4) GNS_GO(4, 4); else GNS_GO(8, 4); }
#undef GNS_GO
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per token row, the row lives in registers (<= MAXV 16-byte vectors per lane).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int C, float eps) {
    constexpr int VE = Vec<T>::N;
    typedef typename Vec<T>::type vec_t;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C / VE;
    float v[MAXV][VE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const vec_t t = *reinterpret_cast<const vec_t*>(x + (size_t)row * C + vi * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                v[i][e] = (float)t[e];
                s += v[i][e];
            }
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const int c = vi * VE + e;
                o[e] = from_f32<T>((v[i][e] - mean) * rstd * gamma[c] + beta[c]);
            }
            *reinterpret_cast<vec_t*>(y + (size_t)row * C + vi * VE) = o;
        }
    }
}

// Short rows (bf16, C = 320 / 640): LPR = 16 / 32 lanes per row, three 16-byte vectors per lane, so a wave normalises 4 / 2
// rows at once with 83 % of its lanes busy (one wave per row leaves 37 % idle at C = 320) and three loads in flight per lane.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_short_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int M, int C, float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int nvec = C / 8;
    const bool live = row < M;
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int vi = sub + i * LPR;
        if (live && vi < nvec) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (size_t)row * C + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)t[e];
                s += v[i][e];
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (live && sub + i * LPR < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int vi = sub + i * LPR;
        if (live && vi < nvec) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + vi * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + vi * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + vi * 8), b1 = *reinterpret_cast<const f32x4*>(beta + vi * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (bf16)((v[i][e] - mean) * rstd * g0[e] + b0[e]);
                o[4 + e] = (bf16)((v[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<bf16x8*>(y + (size_t)row * C + vi * 8) = o;
        }
    }
}

template <typename T>
int launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, int M, int C, float eps,
                     hipStream_t st) {
    constexpr int VE = Vec<T>::N;
    MRISR_REQUIRE(C % VE == 0, "LayerNorm channel alignment");
    const int nvec = C / VE;
    const int need = (nvec + 63) / 64;
    const dim3 grid((M + 3) / 4);
    const T* xi = reinterpret_cast<const T*>(x);
    T* yo = reinterpret_cast<T*>(y);
    ProfScope ps("layernorm", 0.0, 2.0 * M * (double)C * sizeof(T), st);
    if constexpr (sizeof(T) == 2) {
        static const int short_rows = [] { const char* e = getenv("MRISR_LN_SHORT"); return e ? atoi(e) : 1; }();
        if (short_rows && nvec <= 96) {
            const bf16* xb = reinterpret_cast<const bf16*>(x);
            bf16* yb = reinterpret_cast<bf16*>(y);
            if (nvec <= 48) hipLaunchKernelGGL((layernorm_short_kernel<16>), dim3((M + 15) / 16), dim3(256), 0, st, xb, yb, gamma, beta, M, C, eps);
            else hipLaunchKernelGGL((layernorm_short_kernel<32>), dim3((M + 7) / 8), dim3(256), 0, st, xb, yb, gamma, beta, M, C, eps);
            MRISR_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    if (need <= 1) hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, dim3(256), 0, st, xi, yo, gamma, beta, M, C, eps);
    else if (need <= 2) hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, dim3(256), 0, st, xi, yo, gamma, beta, M, C, eps);
    else if (need <= 3) hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, dim3(256), 0, st, xi, yo, gamma, beta, M, C, eps);
    else if (need <= 5) hipLaunchKernelGGL((layernorm_kernel<T, 5>), grid, dim3(256), 0, st, xi, yo, gamma, beta, M, C, eps);
    else if (need <= 10) hipLaunchKernelGGL((layernorm_kernel<T, 10>), grid, dim3(256), 0, st, xi, yo, gamma, beta, M, C, eps);
    else MRISR_REQUIRE(false, "LayerNorm: row too long");
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// row softmax for the materialised (f32 parity) attention: one wave per row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int ld, T* __restrict__ p,
                                                           int ldp, long long rows, int nk) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < nk; c += 64) mx = fmaxf(mx, sr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < nk; c += 64) sum += expf(sr[c] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* pr = p + row * ldp;
    for (int c = lane; c < ldp; c += 64) pr[c] = from_f32<T>(c < nk ? expf(sr[c] - mx) * inv : 0.f);
}

// vectorised variant: the whole row lives in registers (one global read, one exp per element), 16-byte accesses.
// Needs ld == ldp, ld % 4 == 0, ld <= 256 * MAXV.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_vec_kernel(const float* __restrict__ s, int ld, T* __restrict__ p, long long rows,
                                                               int nk) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * ld;
    f32x4 v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < ld) {
            v[i] = *reinterpret_cast<const f32x4*>(sr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c + e >= nk) v[i][e] = -INFINITY;
                mx = fmaxf(mx, v[i][e]);
            }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < ld) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = expf(v[i][e] - mx);  // exp(-inf) = 0 for the masked tail
                sum += v[i][e];
            }
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* pr = p + row * ld;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < ld) {
            if constexpr (sizeof(T) == 2) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[i][e] * inv);
                *reinterpret_cast<bf16x4*>(pr + c) = o;
            } else {
                *reinterpret_cast<f32x4*>(pr + c) = v[i] * inv;
            }
        }
    }
}

template <typename T>
int launch_softmax_rows(const float* s, int ld, void* p, int ldp, long long rows, int nk, hipStream_t st) {
    // NOTE: p may alias s only when sizeof(T) == 4 and ldp == ld (each lane rewrites what it alone read)
    const dim3 grid((unsigned)((rows + 3) / 4));
    ProfScope ps("softmax_rows", 0.0, (double)rows * ld * (4.0 + sizeof(T)), st);
    T* pp = reinterpret_cast<T*>(p);
    if (ld == ldp && ld % 4 == 0 && ld <= 4096 && nk >= 1) {
        const int need = (ld / 4 + 63) / 64;
        if (need <= 1) hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 1>), grid, dim3(256), 0, st, s, ld, pp, rows, nk);
        else if (need <= 2) hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 2>), grid, dim3(256), 0, st, s, ld, pp, rows, nk);
        else if (need <= 4) hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 4>), grid, dim3(256), 0, st, s, ld, pp, rows, nk);
        else if (need <= 8) hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 8>), grid, dim3(256), 0, st, s, ld, pp, rows, nk);
        else hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 16>), grid, dim3(256), 0, st, s, ld, pp, rows, nk);
    } else {
        hipLaunchKernelGGL(softmax_rows_kernel<T>, grid, dim3(256), 0, st, s, ld, pp, ldp, rows, nk);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

template int launch_groupnorm<float>(const GroupNormArgs&, hipStream_t);
template int launch_groupnorm<bf16>(const GroupNormArgs&, hipStream_t);
template int launch_layernorm<float>(const void*, void*, const float*, const float*, int, int, float, hipStream_t);
template int launch_layernorm<bf16>(const void*, void*, const float*, const float*, int, int, float, hipStream_t);
template int launch_softmax_rows<float>(const float*, int, void*, int, long long, int, hipStream_t);
template int launch_softmax_rows<bf16>(const float*, int, void*, int, long long, int, hipStream_t);

}  // namespace mrisr
