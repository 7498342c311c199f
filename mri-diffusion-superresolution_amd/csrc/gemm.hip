// Implicit GEMM on the CDNA4 matrix cores:  D[m][n] = sum_k A[m][k] * W[n][k]
//
//  * one kernel serves every dense contraction of the denoiser: 3x3 convolutions (im2col gathered on the
//    fly from NHWC activations: stride 1/2, optional nearest-x2 up-sampling, zero padding, skip-concat of
//    two sources), 1x1 convolutions / linears (plain rows, optional second K-source = the LoRA rank tail
//    or a concat), and the batched Q.K^T / P.V products of the f32 parity attention;
//  * both operands are K-contiguous, staged global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4)
//    into 128-byte rows whose 16-byte chunks are XOR-swizzled on the SOURCE side (chunk ^= row & 7) so that
//    the ds_read_b128 fragment reads are bank-conflict free; rows that fall outside the matrix or in the
//    conv's zero padding read a device zero page instead;
//  * MFMA is issued with the weight rows as the first operand, so a lane ends up with 4 consecutive
//    output channels n of one output row m: epilogue math (bias, time-embedding row vector, activation,
//    GEGLU gate, residual) is per-lane and the store is one 8/16-byte word per (m, 4n);
//  * bf16 path: v_mfma_f32_16x16x32_bf16;  f32 parity path: v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).
#include "common.h"
#include "prof.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace mrisr {

static void* g_zero_page = nullptr;
static constexpr size_t kZeroPageBytes = 128 * 1024;  // also the all-zero bias vector of bias-free row-panel launches (N <= 32,768)
const void* zero_page() { return g_zero_page; }
int init_zero_page() {
    if (g_zero_page) return 0;
    MRISR_CHECK_HIP(hipMalloc(&g_zero_page, kZeroPageBytes));
    MRISR_CHECK_HIP(hipMemset(g_zero_page, 0, kZeroPageBytes));
    return 0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---- epilogue for 4 consecutive output channels n0..n0+3 of output row m (shared with split-K reduce) ----
template <typename T>
__device__ __forceinline__ void store4(T* p, const float* v);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, const float* v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <typename T>
__device__ __forceinline__ void load4(const T* p, float* v);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float* v) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load4<bf16>(const bf16* p, float* v) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

// V^T [b][h][d][token]: a lane of the MFMA result holds 4 d-values of ONE token, so a direct store is 4 x 2 bytes at a token
// stride.  Exchange with the 3 neighbouring token lanes (fr ^ 1, fr ^ 2) so that every lane ends up with 4 consecutive
// tokens (w0 = tokens 4k, 4k+1; w1 = 4k+2, 4k+3, bf16 pairs) of ONE d row, whose index (0..3) is returned: a single 8-byte
// store.  All four lanes of an exchange must be active together (tile rows and M are multiples of 4).
__device__ __forceinline__ int exchange_tokens4(const float* v, unsigned& w0, unsigned& w1) {
    const int fr = threadIdx.x & 15;
    const bool odd = fr & 1;
    const float s0 = __shfl_xor(odd ? v[0] : v[2], 1);
    const float s1 = __shfl_xor(odd ? v[1] : v[3], 1);
    // even lane: rows d0,d1 for tokens (t, t+1); odd lane: rows d2,d3 for tokens (t-1, t)
    const float a0 = odd ? s0 : v[0], a1 = odd ? v[2] : s0;   // row A: (token lo, token hi)
    const float c0 = odd ? s1 : v[1], c1 = odd ? v[3] : s1;   // row C
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 pa = {(bf16)a0, (bf16)a1}, pc = {(bf16)c0, (bf16)c1};
    const unsigned ua = __builtin_bit_cast(unsigned, pa), uc = __builtin_bit_cast(unsigned, pc);
    const bool hi2 = fr & 2;
    // lanes (fr&2)==0 keep row A and hand row C to fr^2; lanes (fr&2)!=0 keep row C and hand row A over
    const unsigned recv = __shfl_xor(hi2 ? ua : uc, 2);
    w0 = hi2 ? recv : ua;
    w1 = hi2 ? uc : recv;
    return (odd ? 2 : 0) + (hi2 ? 1 : 0);  // even/odd picks (d0|d1) vs (d2|d3); hi2 picks the second of the pair
}

// v: accumulated values (already summed over K) for columns n0..n0+3 (GEGLU: u values; gate in vg).
template <typename T>
__device__ __forceinline__ void epilogue4(const GemmArgs& g, int z, int m, int n0, float* v, const float* vg, const float* zl = nullptr,
                                          T* lds_dst = nullptr, bool resid_later = false, bool mfma_lanes = true, T* lds_t = nullptr,
                                          int ml = 0, int nl = 0, int tpitch = 0, const float* pb = nullptr, const float* pbg = nullptr) {
    // pb / pbg: the column operands of these 4 columns (bias [+ the time-embedding row vector when one row serves every output
    // row]; GEGLU: u and gate bias), preloaded by preload_cols() BEFORE the caller's fragment loop.  Loading them here - inside
    // the caller's per-fragment `if (m < M && n < N)` - put one global-load round trip in front of every fragment's math: 8-20
    // SERIAL L2 latencies per workgroup, which was most of a short-K projection's run time (tools/probes/rp_probe2.sh).
    const float alpha = g.alpha;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.act == ACT_GEGLU) {
        // weight rows are interleaved in blocks of 16: [u0..u15 | g0..g15 | u16.. ]; n0 indexes the u block
        // in the interleaved space, the gate block sits 16 columns later; output column = compacted index.
        float bu[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0};
        if (pb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { bu[r] = pb[r]; bg[r] = pbg[r]; }
        } else if (g.bias) {
            load4<float>(g.bias + n0, bu);
            load4<float>(g.bias + n0 + 16, bg);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] * alpha + bu[r]) * gelu_erf_t<T>(vg[r] * alpha + bg[r]);
        const int nc = (n0 >> 5) * 16 + (n0 & 15);
        T* o = reinterpret_cast<T*>(g.out) + (size_t)m * g.ldo + nc;
        store4<T>(o, v);
        return;
    }
    if (pb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = pb[r];
    } else if (g.bias) {
        load4<float>(g.bias + n0, b4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] * alpha + b4[r];
    if (g.lora_z || zl) {
        // fused LoRA: the rank-r update of these 4 columns (r is tiny: 4 FMAs per output for rank 4).  zl: this row's
        // down-projection computed by the same workgroup (LDS); otherwise z comes from launch_lora_down (global)
        const float* zr = (zl ? zl : g.lora_z + (size_t)m * g.lora_zld) + (n0 / g.lora_secN) * g.lora_r;
        const float* lb = g.lora_b + (size_t)n0 * g.lora_r;
        if (g.lora_r == 4) {  // the common rank: five 16-byte loads instead of twenty scalar ones
            float z4[4];
            load4<float>(zr, z4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float l4[4];
                load4<float>(lb + r * 4, l4);
                v[r] += z4[0] * l4[0] + z4[1] * l4[1] + z4[2] * l4[2] + z4[3] * l4[3];
            }
        } else {
            for (int q = 0; q < g.lora_r; ++q) {
                const float zq = zr[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += zq * lb[r * g.lora_r + q];
            }
        }
    }
    if (g.rowvec && !(pb && g.rowvec_div >= g.M)) {  // (folded into pb when row 0 serves every output row: scalar timestep)
        float t4[4];
        load4<float>(g.rowvec + (size_t)(m / g.rowvec_div) * g.rowvec_ld + n0, t4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t4[r];
    }
    if (g.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (g.act == ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    }
    const size_t zoff = (size_t)(z / g.heads) * g.o_bs + (size_t)(z % g.heads) * g.o_hs;
    if (g.resid && !resid_later) {
        float r4[4];
        load4<T>(reinterpret_cast<const T*>(g.resid) + zoff + (size_t)m * g.ldr + n0, r4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += r4[r];
    }
    if (g.out_mode == OUT_NONE) {
        if (v[0] == 1.2345e30f) store4<T>(reinterpret_cast<T*>(g.out), v);  // benchmark-only: keep the math, drop the store
    } else if (g.out_mode == OUT_ROWS) {
        if (lds_dst) store4<T>(lds_dst, v);  // staged: the workgroup writes the tile out as whole rows afterwards
        else store4<T>(reinterpret_cast<T*>(g.out) + zoff + (size_t)m * g.ldo + n0, v);
    } else if (g.out_mode == OUT_F32) {
        store4<float>(reinterpret_cast<float*>(g.out) + zoff + (size_t)m * g.ldo + n0, v);
    } else {  // OUT_HEADS
        if (lds_dst) {  // staged, [token][channel] section: the tile leaves through LDS as whole (token, head) rows
            store4<T>(lds_dst, v);
            return;
        }
        if (sizeof(T) == 2 && lds_t) {  // staged, transposed section: LDS tile is [channel][token]
            unsigned w0, w1;
            const int drow = exchange_tokens4(v, w0, w1);
            *reinterpret_cast<uint2*>(lds_t + (nl + drow) * tpitch + (ml & ~3)) = make_uint2(w0, w1);
            return;
        }
        const int s = n0 / g.secC;
        const int c = n0 - s * g.secC;
        const int h = c / g.hd;
        const int dd = c - h * g.hd;
        const int b = m / g.ntok;
        const int tok = m - b * g.ntok;
        T* base = reinterpret_cast<T*>(s == 0 ? g.sec_ptr[0] : (s == 1 ? g.sec_ptr[1] : g.sec_ptr[2]));
        const int tr = (s == 0 ? g.sec_tr[0] : (s == 1 ? g.sec_tr[1] : g.sec_tr[2]));
        const size_t bh = (size_t)b * g.nheads + h;
        if (!tr) {
            store4<T>(base + (bh * g.npad + tok) * g.dpad + dd, v);
        } else if (sizeof(T) == 2 && mfma_lanes && (g.ntok & 3) == 0 && (g.M & 3) == 0) {
            // (only from the MFMA kernels, where lanes fr ^ 1, fr ^ 2 hold the neighbouring tokens - NOT from the split-K
            // reduce kernel, whose threads walk the output linearly)
            unsigned w0, w1;
            const int drow = exchange_tokens4(v, w0, w1);
            T* o = base + (bh * g.dpad + dd + drow) * g.npad + (tok & ~3);
            *reinterpret_cast<uint2*>(o) = make_uint2(w0, w1);
        } else {
            T* o = base + (bh * g.dpad + dd) * g.npad + tok;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * g.npad] = from_f32<T>(v[r]);
        }
    }
}

// Column operands of a lane's NFR column fragments (fragment i covers columns nbase + 16 i .. + 3), all loads issued back to
// back ahead of the epilogue loop (clamped addresses instead of a per-lane branch: columns past N are never stored).
template <int NFR>
__device__ __forceinline__ void preload_cols(const GemmArgs& g, int nbase, float (&pb)[NFR][4]) {
    const bool fold = g.rowvec && g.rowvec_div >= g.M;
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        const int n = min(nbase + i * 16, g.N - 4);
        float b[4] = {0.f, 0.f, 0.f, 0.f}, t[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) load4<float>(g.bias + n, b);
        if (fold) load4<float>(g.rowvec + n, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[i][r] = b[r] + t[r];
    }
}

// ---- MFMA wrappers -------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& w, const bf16x8& a) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const f32x4& w, const f32x4& a) {
    // lane group g supplies k = 4g+j at step j: a permutation of the 16 k's shared by both operands
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[j], acc, 0, 0, 0);
}

// Test hook (mrisr_debug_gemm_flags & 2048): fill the workgroup's WHOLE LDS allocation with 0xFF bytes (NaN as bf16, as f32 and as
// e4m3) before the kernel's first LDS-DMA.  A freshly scheduled workgroup otherwise inherits the LDS image the previous workgroup
// of the same launch left behind - in these kernels the same kind of data at the same offsets, often the very same weight tile -,
// so a `ds_read` that is not ordered behind the DMA it depends on (the issuing wave's covering vmcnt + a barrier the reader has
// passed: nothing else orders it, MI355X_MICROARCH.md item 7) returns plausible, frequently CORRECT stale bytes and passes every
// reference check and repeatability screen.  Poisoned, such a read returns NaN and the parity tests see it.
__device__ __forceinline__ void lds_poison(char* smem_base, int nthreads) {
    // hsa_kernel_dispatch_packet_t::group_segment_size (byte 28): static + dynamic LDS of this launch
    const unsigned bytes = ((const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_dispatch_ptr())[7];
    for (unsigned o = threadIdx.x * 16u; o + 16u <= bytes; o += (unsigned)nthreads * 16u)
        *reinterpret_cast<uint4*>(smem_base + o) = make_uint4(~0u, ~0u, ~0u, ~0u);
    __syncthreads();
}
static int gemm_flags_now();

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g, const char* __restrict__ zero) {
    constexpr int EB = sizeof(T);
    constexpr int CH = 16 / EB;   // elements per 16-byte chunk
    constexpr int BK = 8 * CH;    // elements per 128-byte LDS row
    constexpr int A_IT = BM / 32, W_IT = BN / 32;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MF = WTM / 16, NF = WTN / 16;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.z;
    const int split = blockIdx.y;
    if (g.dbg & 2048) lds_poison(smem, (int)blockDim.x);

    // ---- tile decode with an XCD-contiguous remap (blocks b, b+8, ... share an XCD/L2) ----
    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = (g.M + BM - 1) / BM;
    const int nblk = ntn * ntm;
    int logical;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // grouped order: within a group of `group_m` M tiles walk M first, so the workgroups that run together on an XCD share
    // their weight tiles (group_m ways) as well as their activation tiles through that XCD's L2 (N-first order re-streamed
    // the whole weight matrix once per M tile: 3-6x the algorithmic bytes on the wide-N and deep-K layers)
    int tn, tm;
    {
        const int GM = g.group_m > 1 ? g.group_m : 1;
        const int per_group = GM * ntn;
        const int grp = logical / per_group;
        const int first_m = grp * GM;
        const int gsz = min(GM, ntm - first_m);
        const int within = logical - grp * per_group;
        tm = first_m + within % gsz;
        tn = within / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const T* a0 = reinterpret_cast<const T*>(g.a0) + (size_t)z * g.a_bs;
    const T* a1 = reinterpret_cast<const T*>(g.a1);
    const T* w = reinterpret_cast<const T*>(g.w) + (size_t)z * g.w_bs;

    // ---- per-thread loader geometry ----
    const int lrow = tid >> 3, pch = tid & 7;
    const T* wsrc[W_IT];
    bool wok[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int row = it * 32 + lrow;
        const int n = n0 + row;
        const int c = pch ^ (row & 7);
        wok[it] = n < g.N;
        wsrc[it] = w + (size_t)n * g.K + c * CH;
    }
    // plain mode: row pointers; conv mode: output pixel coordinates
    const T* asrc0[A_IT];
    const T* asrc1[A_IT];
    int ab[A_IT], ay[A_IT], ax[A_IT];
    bool aok[A_IT];
    int acoff[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = it * 32 + lrow;
        const int m = m0 + row;
        const int c = pch ^ (row & 7);
        aok[it] = m < g.M;
        acoff[it] = c * CH;
        if (!g.conv) {
            asrc0[it] = a0 + (size_t)m * g.lda0 + c * CH;
            asrc1[it] = a1 ? a1 + (size_t)m * g.lda1 + c * CH : nullptr;
            ab[it] = ay[it] = ax[it] = 0;
        } else {
            const int hw = g.Hout * g.Wout;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / g.Wout;
            ab[it] = b;
            ay[it] = oy * g.stride - g.pad;
            ax[it] = (rem - oy * g.Wout) * g.stride - g.pad;
            asrc0[it] = asrc1[it] = nullptr;
        }
    }

    // ---- K range of this split ----
    const int nkt = g.K / BK;
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int kt_beg = split * per;
    const int kt_end = min(nkt, kt_beg + per);
    const int Ct = g.c0 + g.c1;
    // conv k-walk state (uniform)
    int tap = 0, cc = 0;
    if (g.conv) {
        const int k0 = kt_beg * BK;
        tap = k0 / Ct;
        cc = k0 - tap * Ct;
    }
    const int Hc = g.Hin << g.ups, Wc = g.Win << g.ups;

    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        const int k0 = kt * BK;
        // W tile
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const void* src = wok[it] ? (const void*)(wsrc[it] + k0) : (const void*)zero;
            glds16(src, sb + BM * 128 + (it * 256 + wave * 64) * 16);
        }
        if (!g.conv) {
            const bool second = k0 >= g.c0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const void* src = zero;
                if (aok[it]) src = second ? (const void*)(asrc1[it] + (k0 - g.c0)) : (const void*)(asrc0[it] + k0);
                glds16(src, sb + (it * 256 + wave * 64) * 16);
            }
        } else {
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool second = cc >= g.c0;
            const T* base = second ? a1 : a0;
            const int ld = second ? g.lda1 : g.lda0;
            const int ch = second ? cc - g.c0 : cc;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int iy = ay[it] + ky, ix = ax[it] + kx;
                const void* src = zero;
                if (aok[it] && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc && (!g.zstuff || ((iy | ix) & 1) == 0)) {
                    const size_t pix = ((size_t)ab[it] * g.Hin + (iy >> g.ups)) * g.Win + (ix >> g.ups);
                    src = base + pix * ld + ch + acoff[it];
                }
                glds16(src, sb + (it * 256 + wave * 64) * 16);
            }
            cc += BK;
            if (cc >= Ct) { cc = 0; ++tap; }
        }
    };

    // ---- accumulators ----
    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave / WGN) * WTM, wn0 = (wave % WGN) * WTN;
    const int fr = lane & 15, fg = lane >> 4;

    typedef typename Frag<T>::type frag_t;
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE;
        const char* sw = sa + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int phys = ((kk * 4 + fg) ^ (fr & 7)) * 16;
            frag_t wf[NF], af[MF];
#pragma unroll
            for (int i = 0; i < NF; ++i)
                wf[i] = *reinterpret_cast<const frag_t*>(sw + (wn0 + i * 16 + fr) * 128 + phys);
#pragma unroll
            for (int j = 0; j < MF; ++j)
                af[j] = *reinterpret_cast<const frag_t*>(sa + (wm0 + j * 16 + fr) * 128 + phys);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) mma(acc[i][j], wf[i], af[j]);
        }
    };

    if (kt_beg < kt_end) {
        stage(kt_beg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            if (kt + 1 < kt_end) stage(kt + 1, cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue ----
    if (g.splitk > 1) {
        float* part = g.partial + ((size_t)z * g.splitk + split) * (size_t)g.M * g.N;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wm0 + j * 16 + fr;
                const int n = n0 + wn0 + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4<float>(part + (size_t)m * g.N + n, v);
                }
            }
        return;
    }
    float pb[NF][4];
    preload_cols<NF>(g, n0 + wn0 + fg * 4, pb);
    if (g.act == ACT_GEGLU) {
#pragma unroll
        for (int i = 0; i < NF; i += 2)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wm0 + j * 16 + fr;
                const int n = n0 + wn0 + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float vg[4] = {acc[i + 1][j][0], acc[i + 1][j][1], acc[i + 1][j][2], acc[i + 1][j][3]};
                    epilogue4<T>(g, z, m, n, v, vg, nullptr, nullptr, false, true, nullptr, 0, 0, 0, pb[i], pb[i + 1 < NF ? i + 1 : i]);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm0 + j * 16 + fr;
            const int n = n0 + wn0 + i * 16 + fg * 4;
            if (m < g.M && n < g.N) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue4<T>(g, z, m, n, v, nullptr, nullptr, nullptr, false, true, nullptr, 0, 0, 0, pb[i]);
            }
        }
}

// =================================================================================================
// bf16 throughput kernel, buffer-addressed ("bl"): same tiling / LDS image / MFMA orientation as gemm_kernel,
// but the LDS-DMA goes through buffer descriptors (`buffer_load_dwordx4 ... offen lds`):
//   address = SGPR base (descriptor) + per-lane VGPR byte offset + SGPR soffset.
// The per-lane offsets are computed ONCE per (3x3 tap, concat source) segment - for plain rows once per
// kernel - and the walk along K is a scalar soffset add, so the steady-state loop issues no vector ALU for
// addressing at all (the 64-bit per-lane address arithmetic of gemm_kernel cost ~3.6 VALU per MFMA and made the
// loop VALU-issue bound: profiles/r01_pmc_*).  Rows outside the matrix / in the zero padding get an offset
// beyond num_records: the hardware range check returns zeros, no zero page and no select.
// =================================================================================================
// staged epilogue: the tile sits in LDS as [BM][pitch] T; write it out row by row, 16 bytes per lane
template <int BM, int BN, int NT = 256>
__device__ __forceinline__ void copy_out_tile(const GemmArgs& g, const bf16* otile, int pitch, int m0, int n0, int z, bool add_resid) {
    __syncthreads();
    constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
    bf16* out = reinterpret_cast<bf16*>(g.out) + (size_t)z * g.o_bs;
    const bf16* res = add_resid ? reinterpret_cast<const bf16*>(g.resid) + (size_t)z * g.o_bs : nullptr;
    // Residual pieces of a BATCH of rows are loaded first, then added and stored: the one-piece-at-a-time loop (load, add, store; the store may alias
    // the next load as far as the compiler knows) was a chain of dependent L2 round trips - 5 to 10 per workgroup.  A lane only ever reads the addresses
    // it writes itself, so reading ahead of its own stores is safe when the residual aliases the output.
    constexpr int TOTAL = BM * CPR, IT = (TOTAL + NT - 1) / NT, BATCH = IT < 5 ? IT : 5;
    for (int it0 = 0; it0 < IT; it0 += BATCH) {
        bf16x8 rv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = (int)threadIdx.x + (it0 + u) * NT;
            const int row = idx / CPR, ch = idx - row * CPR;
            const int m = m0 + row, n = n0 + ch * 8;
            if (res && idx < TOTAL && m < g.M && n + 8 <= g.N) rv[u] = *reinterpret_cast<const bf16x8*>(res + (size_t)m * g.ldr + n);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = (int)threadIdx.x + (it0 + u) * NT;
            const int row = idx / CPR, ch = idx - row * CPR;
            const int m = m0 + row, n = n0 + ch * 8;
            if (idx >= TOTAL || m >= g.M || n >= g.N) continue;
            const bf16* src = otile + row * pitch + ch * 8;
            bf16* dst = out + (size_t)m * g.ldo + n;
            if (n + 8 <= g.N) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
                if (res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rv[u][e]);
                }
                *reinterpret_cast<bf16x8*>(dst) = v;
            } else {
                bf16x4 v = *reinterpret_cast<const bf16x4*>(src);  // N % 8 == 4 tail
                if (res) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(res + (size_t)m * g.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16)((float)v[e] + (float)r[e]);
                }
                *reinterpret_cast<bf16x4*>(dst) = v;
            }
        }
    }
}

// head-major outputs (OUT_HEADS) of a tile that lies inside ONE section: [token][channel] sections are staged like row outputs
// and leave as 16-byte pieces of the (token, head) rows - consecutive tokens are adjacent in [b][h][token][d], so a tile
// writes long runs; transposed sections (V^T [b][h][d][token]) are staged as [channel][token] and leave as 16 bytes = 8
// tokens of one d row (BM tokens = one contiguous run per row).  Before: 8-byte stores from the accumulator layout.
template <int BM, int BN>
__device__ __forceinline__ void copy_out_heads(const GemmArgs& g, const bf16* otile, int m0, int n0, bool tr) {
    __syncthreads();
    const int s = n0 / g.secC;
    bf16* base = reinterpret_cast<bf16*>(s == 0 ? g.sec_ptr[0] : (s == 1 ? g.sec_ptr[1] : g.sec_ptr[2]));
    if (!tr) {
        constexpr int CPR = BN / 8, P = BN + 8;
        for (int idx = threadIdx.x; idx < BM * CPR; idx += 256) {
            const int row = idx / CPR, ch = idx - row * CPR;
            const int m = m0 + row, n = n0 + ch * 8;
            if (m >= g.M || n >= g.N) continue;
            const int c = n - s * g.secC, h = c / g.hd, dd = c - h * g.hd;
            const int b = m / g.ntok, tok = m - b * g.ntok;
            *reinterpret_cast<bf16x8*>(base + (((size_t)b * g.nheads + h) * g.npad + tok) * g.dpad + dd) =
                *reinterpret_cast<const bf16x8*>(otile + row * P + ch * 8);
        }
    } else {
        constexpr int TPC = BM / 8, P = BM + 8;
        for (int idx = threadIdx.x; idx < BN * TPC; idx += 256) {
            const int col = idx / TPC, tc = idx - col * TPC;
            const int n = n0 + col, m = m0 + tc * 8;
            if (m >= g.M || n >= g.N) continue;
            const int c = n - s * g.secC, h = c / g.hd, dd = c - h * g.hd;
            const int b = m / g.ntok, tok = m - b * g.ntok;
            *reinterpret_cast<bf16x8*>(base + (((size_t)b * g.nheads + h) * g.dpad + dd) * g.npad + tok) =
                *reinterpret_cast<const bf16x8*>(otile + col * P + tc * 8);
        }
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void bl16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
#define BL_OOB 0x80000000u


// ---- weight pre-touch -----------------------------------------------------------------------------------------------------------
// In-model the weights of a layer are cold (HBM) when its GEMM starts, and the latency-bound GEMMs then pay an HBM round trip in every K step
// in which some workgroup touches a weight tile first: with the weights pre-touched the GEMMs of one denoising step take 0.69 ms less
// (profiles/r02y_weight_prefetch.log).  A side stream costs more than that in graph dependencies, so the kernel does it itself: right at
// its start every workgroup DMAs ITS share of the whole weight matrix (1-KiB pieces, no use of the data) into a stage buffer that the first
// real stage overwrites later.  All shares together request the matrix from HBM once, at full bandwidth, and the K loop's own fetches then
// find it in the memory-side cache / L2.  The pieces are older than the first stage's DMA, so the first `vmcnt(0)` covers them.
__device__ __forceinline__ void pretouch_weights(const __amdgpu_buffer_rsrc_t& rw, long long w_bytes, char* scratch, int wave, int lane, int cap) {
    // (a per-XCD form - the workgroups of each XCD touch the whole matrix, so that it sits in all eight L2s - measured 0.5-1.5 % slower)
    const int n_wg = (int)(gridDim.x * gridDim.y);
    const int wg = (int)(blockIdx.x + blockIdx.y * gridDim.x);
    const long long pieces = (w_bytes + 1023) >> 10;
    const int per_wg = (int)((pieces + n_wg - 1) / n_wg);
    const int per_wave = min((per_wg + 3) >> 2, cap);
    const long long first = (long long)wg * per_wg + (long long)wave * per_wave;
    for (int i = 0; i < per_wave; ++i) {
        const long long pc = first + i;
        const unsigned off = pc < pieces ? (unsigned)(pc << 10) + (unsigned)lane * 16u : BL_OOB;
        bl16(rw, scratch + ((wave * 2 + (i & 1)) << 10), off, 0u);  // 8 KB of scratch: fits the smallest stage (BN = 64)
    }
}

// NSTAGE = 2: double buffer, one drain (vmcnt(0)) + barrier per K tile; two workgroups share a CU and hide each other's
// load latency.  Deeper rings with a counted vmcnt were tried twice (8-wave kernels, and 3-stage 4-wave variants on
// in-range shapes) and lost every time (profiles/r01_gemm_sweep_incl_deep_stages.log, profiles/r01b_ws_sweep.log); note that
// a fully out-of-range LDS-DMA instruction retires out of order, so a counted wait is only safe without padding rows.
// ---- split-K, second half inside the GEMM kernel ------------------------------------------------------------------------------
// Every split stores its f32 tile to its slab, releases it (agent-scope fence: the slab leaves this XCD's L2) and takes a ticket on
// the tile's counter; the LAST split to arrive acquires, re-reads all slabs in split order (its own included: the sum is the same
// bits whichever split arrives last) into its accumulators, resets the counter for the next launch and falls through to the
// kernel's normal epilogue.  Returns true in the workgroup that goes on.  `ticket_lds`: 4 bytes of drained LDS.
template <int NF, int MF>
__device__ __forceinline__ bool splitk_last_arriver(const GemmArgs& g, f32x4 (&acc)[NF][MF], unsigned tile_idx, const float* slab0, int m_base, int n_base,
                                                    int fr, int fg, char* ticket_lds) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0)
        *reinterpret_cast<volatile unsigned*>(ticket_lds) = __hip_atomic_fetch_add(g.sk_counters + tile_idx, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<volatile unsigned*>(ticket_lds);
    if (ticket != (unsigned)g.splitk - 1) return false;
    __syncthreads();  // the ticket word may be overwritten by the epilogue's staging from here on
    if (threadIdx.x == 0) __hip_atomic_store(g.sk_counters + tile_idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const size_t slab = (size_t)g.M * g.N;
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.splitk; ++s) {
        const float* p = slab0 + (size_t)s * slab;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m_base + j * 16 + fr;
                const int n = n_base + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)m * g.N + n));
                    acc[i][j] += t;
                }
            }
    }
    return true;
}

template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool LORA>
__global__ __launch_bounds__(256, 2) void gemm_bl_kernel(const GemmArgs g) {  // 2 waves / SIMD: two workgroups per CU hide each other's loads
    // NSTAGE 3 / 4 (round 2): a ring with a COUNTED vmcnt - NSTAGE - 1 K tiles in flight per workgroup.  For the shapes whose
    // whole grid is resident at once (level-2 linears: M = 2,048, 640 workgroups of 64 x 64 = 2.5 per CU) the run time is the
    // length of ONE workgroup's chain of K-step latencies (20 steps x ~1 us); nothing else can overlap it, so the chain itself
    // has to be pipelined.  The launcher admits these configurations only for plain GEMMs that tile exactly (every DMA
    // instruction in range: a fully out-of-range LDS-DMA instruction retires out of order and would satisfy the count early).
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2 (drain per tile), 3 or 4 (counted ring)");
    typedef bf16 T;
    constexpr int BK = 64;
    constexpr int A_IT = BM / 32, W_IT = BN / 32;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MF = WTM / 16, NF = WTN / 16;
    constexpr int STAGE = (BM + BN) * 128 + (LORA ? 16 * 128 : 0);
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.z;
    const int split = blockIdx.y;
    if (g.dbg & 2048) lds_poison(smem, (int)blockDim.x);

    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = (g.M + BM - 1) / BM;
    const int nblk = ntn * ntm;
    int logical;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // grouped order: within a group of `group_m` M tiles walk M first, so the workgroups that run together on an XCD share
    // their weight tiles (group_m ways) as well as their activation tiles through that XCD's L2 (N-first order re-streamed
    // the whole weight matrix once per M tile: 3-6x the algorithmic bytes on the wide-N and deep-K layers)
    int tn, tm;
    {
        const int GM = g.group_m > 1 ? g.group_m : 1;
        const int per_group = GM * ntn;
        const int grp = logical / per_group;
        const int first_m = grp * GM;
        const int gsz = min(GM, ntm - first_m);
        const int within = logical - grp * per_group;
        tm = first_m + within % gsz;
        tn = within / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const T* a0p = reinterpret_cast<const T*>(g.a0) + (size_t)z * g.a_bs;
    const T* a1p = reinterpret_cast<const T*>(g.a1);
    const T* wp = reinterpret_cast<const T*>(g.w) + (size_t)z * g.w_bs;
    const long long a_rows = g.conv ? (long long)g.B * g.Hin * g.Win : (long long)g.M;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)min((long long)g.N * g.K * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(a0p, (unsigned)min(a_rows * g.lda0 * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(a1p ? (const void*)a1p : (const void*)a0p,
                                                 a1p ? (unsigned)min(a_rows * g.lda1 * 2, 0x7FFFFFFFll) : 0u);

    const int lrow = tid >> 3, pch = tid & 7;
    unsigned wvo[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int row = it * 32 + lrow;
        const int n = n0 + row;
        const int c = pch ^ (row & 7);
        wvo[it] = n < g.N ? (unsigned)(((size_t)n * g.K + c * 8) * 2) : BL_OOB;
    }
    // A rows: byte offsets into the current source for the current segment (plain: fixed for each source)
    unsigned avo[A_IT], avo1[A_IT];
    int ab[A_IT], ay[A_IT], ax[A_IT];  // conv: b*Hin, oy*stride-1, ox*stride-1  (ab < 0: row outside the matrix)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = it * 32 + lrow;
        const int m = m0 + row;
        const int c = pch ^ (row & 7);
        const bool ok = m < g.M;
        avo[it] = avo1[it] = BL_OOB;
        ab[it] = -1;
        ay[it] = ax[it] = 0;
        if (!g.conv) {
            if (ok) {
                avo[it] = (unsigned)(((size_t)m * g.lda0 + c * 8) * 2);
                avo1[it] = (unsigned)(((size_t)m * g.lda1 + c * 8) * 2);
            }
        } else if (ok) {
            const int hw = g.Hout * g.Wout;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / g.Wout;
            ab[it] = b * g.Hin;
            ay[it] = oy * g.stride - g.pad + (g.subpix ? (z >> 1) : 0);  // sub-pixel up-sampling: the window of parity (py, px) starts py / px later
            ax[it] = (rem - oy * g.Wout) * g.stride - g.pad + (g.subpix ? (z & 1) : 0);
        }
    }
    const int acb = pch * 16;  // physical chunk byte offset is applied through `c` below for conv rows
    // adapter rows (LORA): waves 0 and 1 each bring 8 of the 16 rows of every K tile
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(LORA ? g.lora_a : (const void*)wp, LORA ? (unsigned)((long long)g.lora_R * g.K * 2) : 0u);
    unsigned lvo = BL_OOB;
    if (LORA && (wave < 2 || NSTAGE > 2)) {  // ring variants: waves 2, 3 re-issue the two pieces (same bytes, same LDS addresses) so
                                             // that every wave's vmcnt sees the same number of DMA instructions per stage
        const int row = (wave & 1) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        if (row < g.lora_R) lvo = (unsigned)(((size_t)row * g.K + c * 8) * 2);
        // ring variants: no out-of-range lanes (a FULLY out-of-range piece - rows 8..15 at R <= 8 - would retire out of order and
        // break the counted wait); the rows >= R then carry a copy of row R - 1, whose z columns only ever meet zero B entries
        else if (NSTAGE > 2) lvo = (unsigned)(((size_t)(g.lora_R - 1) * g.K + c * 8) * 2);
    }

    const int nkt = g.K / BK;
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int kt_beg = split * per;
    const int kt_end = min(nkt, kt_beg + per);
    const int Ct = g.c0 + g.c1;
    const int Hc = g.Hin << g.ups, Wc = g.Win << g.ups;
    // conv walk state (uniform): current tap and channel position; seg_key identifies (tap, source)
    int tap = 0, cc = 0, seg_key = -1;
    if (g.conv) {
        const int k0 = kt_beg * BK;
        tap = k0 / Ct;
        cc = k0 - tap * Ct;
    }

    // a stage's DMA in two halves (weights + adapter rows | activation rows).  Probe (g.dbg & 16384): the second half issued between the
    // two K sub-steps of the tile being computed instead of at its top - what helps the one-wave-per-SIMD row-panel kernels (the wave's
    // time in the VMEM queue overlaps MFMAs already issued) costs this kernel 2.3 % of the whole step (10.10 -> 10.33 ms, same box,
    // profiles/r03z_bl_ab.log): with two workgroups per CU the other workgroup already fills the stall, and the later issue shortens the
    // time the tile has to land.  Off.
    auto stage_w = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        const unsigned k0b = (unsigned)kt * BK * 2;
#pragma unroll
        for (int it = 0; it < W_IT; ++it) bl16(rw, sb + BM * 128 + (it * 256 + wave * 64) * 16, wvo[it], k0b);
        if (LORA && (wave < 2 || NSTAGE > 2)) bl16(rl, sb + (BM + BN) * 128 + (wave & 1) * 64 * 16, lvo, k0b);
    };
    auto stage_a = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE;
        const unsigned k0b = (unsigned)kt * BK * 2;
        if (!g.conv) {
            const bool second = kt * BK >= g.c0;
            if (!second) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) bl16(ra0, sb + (it * 256 + wave * 64) * 16, avo[it], k0b);
            } else {
                const unsigned kb = (unsigned)(kt * BK - g.c0) * 2;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) bl16(ra1, sb + (it * 256 + wave * 64) * 16, avo1[it], kb);
            }
        } else {
            const bool second = cc >= g.c0;
            const int key = tap * 2 + (second ? 1 : 0);
            if (key != seg_key) {  // new (tap, source) segment: refresh the per-lane offsets (uniform branch)
                seg_key = key;
                const int ky = g.kw == 3 ? tap / 3 : tap / g.kw, kx = tap - ky * g.kw;
                const int ld = second ? g.lda1 : g.lda0;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const int row = it * 32 + lrow;
                    const int c = pch ^ (row & 7);
                    const int iy = ay[it] + ky, ix = ax[it] + kx;
                    const bool ok = ab[it] >= 0 && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc && (!g.zstuff || ((iy | ix) & 1) == 0);
                    const unsigned pix = (unsigned)((ab[it] + (iy >> g.ups)) * g.Win + (ix >> g.ups));
                    avo[it] = ok ? (pix * (unsigned)ld + (unsigned)c * 8u) * 2u : BL_OOB;
                }
            }
            const unsigned chb = (unsigned)(second ? cc - g.c0 : cc) * 2;
            if (!second) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) bl16(ra0, sb + (it * 256 + wave * 64) * 16, avo[it], chb);
            } else {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) bl16(ra1, sb + (it * 256 + wave * 64) * 16, avo[it], chb);
            }
            cc += BK;
            if (cc >= Ct) { cc = 0; ++tap; }
        }
    };
    auto stage = [&](int kt, int buf) { stage_w(kt, buf); stage_a(kt, buf); };
    (void)acb;

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave / WGN) * WTM, wn0 = (wave % WGN) * WTN;
    const int fr = lane & 15, fg = lane >> 4;
    const bool zwave = LORA && (wave % WGN) == 0;  // the wave column that also accumulates z for its rows
    // rank-4 adapters: this lane's slab of s*B for the up-projection MFMA after the loop (row n, its section's 4 columns,
    // everything else zero) - fetched now so that the loop hides the latency
    const bool lora_mma = LORA && g.lora_r == 4 && g.alpha == 1.0f && !(g.dbg & 8);
    bf16x8 lbf[NF];
    if (LORA) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            lbf[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            const int n = n0 + wn0 + i * 16 + fr;
            if (lora_mma && n < g.N) {
                const int sec = n / g.lora_secN;  // this column's adapter slot: z columns [4 sec, 4 sec + 4)
                if ((sec >> 1) == fg) {
                    const f32x4 l = *reinterpret_cast<const f32x4*>(g.lora_b + (size_t)n * 4);
                    if (sec & 1) { lbf[i][4] = (bf16)l[0]; lbf[i][5] = (bf16)l[1]; lbf[i][6] = (bf16)l[2]; lbf[i][7] = (bf16)l[3]; }
                    else { lbf[i][0] = (bf16)l[0]; lbf[i][1] = (bf16)l[1]; lbf[i][2] = (bf16)l[2]; lbf[i][3] = (bf16)l[3]; }
                }
            }
        }
    }
    f32x4 zacc[MF];
#pragma unroll
    for (int j = 0; j < MF; ++j) zacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf, auto&& mid) {
        const char* sa = smem + buf * STAGE;
        const char* sw = sa + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1) mid();
            const int phys = ((kk * 4 + fg) ^ (fr & 7)) * 16;
            bf16x8 wf[NF], af[MF];
#pragma unroll
            for (int i = 0; i < NF; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sw + (wn0 + i * 16 + fr) * 128 + phys);
#pragma unroll
            for (int j = 0; j < MF; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sa + (wm0 + j * 16 + fr) * 128 + phys);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            if (LORA && zwave) {
                const bf16x8 lf = *reinterpret_cast<const bf16x8*>(sw + (BN + fr) * 128 + phys);
#pragma unroll
                for (int j = 0; j < MF; ++j) zacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, af[j], zacc[j], 0, 0, 0);
            }
        }
    };

    if constexpr (NSTAGE == 2) {
        if (kt_beg < kt_end) {
            if (g.pretouch) pretouch_weights(rw, (long long)g.N * g.K * 2, smem + STAGE, wave, lane, g.pretouch);
            stage(kt_beg, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int cur = 0;
            const bool halves = (g.dbg & 16384) != 0;
            for (int kt = kt_beg; kt < kt_end; ++kt) {
                const bool more = kt + 1 < kt_end;
                if (more) { if (halves) stage_w(kt + 1, cur ^ 1); else stage(kt + 1, cur ^ 1); }
                compute(cur, [&]() {
                    if (halves) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) stage_a(kt + 1, cur ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
        }
    } else {
        constexpr int PER = A_IT + W_IT + (LORA ? 1 : 0);  // DMA instructions per lane and stage (uniform over the waves)
        static_assert(PER * (NSTAGE - 2) <= 63, "vmcnt range");
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (kt_beg + s < kt_end) stage(kt_beg + s, s);
        int slot = 0;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            // tile kt must have landed; up to NSTAGE - 2 younger tiles stay in flight
            const int younger = min(NSTAGE - 2, kt_end - 1 - kt);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * 2) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // RAW barrier, never __syncthreads() here: an LDS-DMA in flight is a pending LDS write on the VM counter, so the fence
            // inside __syncthreads() makes hipcc emit `s_waitcnt vmcnt(0)` in front of the s_barrier - it drained the whole ring on
            // every K step (rounds 1-2 measured "3 or 4 stages = 2 stages" with exactly that: the counted wait two lines up was
            // followed by a vmcnt(0) in the ISA).  Tile kt is ordered for every wave's ds_read by each issuing wave's counted
            // vmcnt + this barrier; the slot refilled below was last READ in compute(kt - 1), whose ds_reads have all returned
            // (their MFMAs consumed them; lgkmcnt(0) makes it explicit) before any wave passes this barrier.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NSTAGE - 1 < kt_end) stage(kt + NSTAGE - 1, (slot + NSTAGE - 1) % NSTAGE);
            compute(slot, []() {});
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        __syncthreads();  // the epilogue reuses the stage buffers
    }

    float* zlds = reinterpret_cast<float*>(smem);  // [BM][16] f32: z rows of this tile (the stages are drained)
    if (LORA) {
        if (zwave) {
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int ml = wm0 + j * 16 + fr;
                *reinterpret_cast<f32x4*>(zlds + ml * 16 + fg * 4) = zacc[j];  // z[q = 4fg + r][m = fr]
                if (g.lora_zout && tn == 0 && m0 + ml < g.M && fg * 4 < g.lora_R)
                    *reinterpret_cast<f32x4*>(g.lora_zout + (size_t)(m0 + ml) * g.lora_R + fg * 4) = zacc[j];
            }
        }
        __syncthreads();
    }
    // rank-4 adapters: the up-projection out += z (s B)^T is one more K step on the matrix cores - z (bf16) as the row
    // operand, the 16 x BN slab of s*B (this lane's row, its section's 4 columns, everything else zero) as the weight
    // operand - instead of 16 FMAs and four 16-byte loads per 4 outputs in the epilogue (which cost 0.35 ms per step).
    if (LORA && lora_mma) {
        bf16x8 zf[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            zf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (fg < 2) {
                const float* zp = zlds + (wm0 + j * 16 + fr) * 16 + fg * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(zp), b = *reinterpret_cast<const f32x4*>(zp + 4);
                zf[j] = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbf[i], zf[j], acc[i][j], 0, 0, 0);
    }
    if (g.splitk > 1) {
        float* part = g.partial + ((size_t)z * g.splitk + split) * (size_t)g.M * g.N;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wm0 + j * 16 + fr;
                const int n = n0 + wn0 + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4<float>(part + (size_t)m * g.N + n, v);
                }
            }
        // (the widest tiles keep the separate reduce kernel: the slab loop on top of 24+ accumulator fragments spills)
        if constexpr (NF * MF > 20) return;
        else {
            if (!g.sk_counters) return;
            if (!splitk_last_arriver<NF, MF>(g, acc, (unsigned)(z * gridDim.x + blockIdx.x), g.partial + (size_t)z * g.splitk * (size_t)g.M * g.N,
                                             m0 + wm0, n0 + wn0, fr, fg, smem))
                return;
        }
    }
    float pb[NF][4];
    preload_cols<NF>(g, n0 + wn0 + fg * 4, pb);
    if (g.act == ACT_GEGLU) {
        if constexpr (NF % 2 == 0) {
#pragma unroll
            for (int i = 0; i < NF; i += 2)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = m0 + wm0 + j * 16 + fr;
                    const int n = n0 + wn0 + i * 16 + fg * 4;
                    if (m < g.M && n < g.N) {
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        float vg[4] = {acc[i + 1][j][0], acc[i + 1][j][1], acc[i + 1][j][2], acc[i + 1][j][3]};
                        epilogue4<T>(g, z, m, n, v, vg, nullptr, nullptr, false, true, nullptr, 0, 0, 0, pb[i], pb[i + 1]);
                    }
                }
        }
        return;
    }
    // plain row outputs leave through LDS: the accumulator layout gives every lane 8 bytes of a row (32-byte runs per
    // store instruction, lines shared between waves - TCC counters showed 1.5x the output bytes going to HBM); staged, the
    // workgroup writes whole rows with 16-byte stores.  The stage buffers are free here (after the LORA z rows).
    constexpr int OPITCH = BN + 8;  // elements; 16-byte aligned rows, bank-spread
    T* otile = reinterpret_cast<T*>(smem + (LORA ? BM * 16 * 4 : 0));
    const bool staged = g.out_mode == OUT_ROWS && (g.ldo & 7) == 0 && g.heads == 1 && g.stage_out;
    // the residual is then added in the copy-out pass (whole-row reads; the staged value is already rounded to bf16)
    const bool resid_later = staged && g.resid != nullptr && (g.ldr & 7) == 0 && g.stage_out < 3;
    // head-major outputs: staged when the tile lies inside one section (Q, K or V) and everything is 8-aligned
    bool hstaged = false, htr = false;
    if (g.out_mode == OUT_HEADS && g.stage_out && !(g.dbg & 16)) {
        const int sec = n0 / g.secC;
        hstaged = (min(n0 + BN, g.N) - 1) / g.secC == sec && ((g.N | g.M | g.hd | g.secC | g.dpad | g.npad | g.ntok) & 7) == 0 && g.heads == 1 &&
                  g.batch == 1;
        htr = (sec == 0 ? g.sec_tr[0] : (sec == 1 ? g.sec_tr[1] : g.sec_tr[2])) != 0;
    }
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm0 + j * 16 + fr;
            const int n = n0 + wn0 + i * 16 + fg * 4;
            if (m < g.M && n < g.N) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue4<T>(g, z, m, n, v, nullptr, LORA && !lora_mma ? zlds + (m - m0) * 16 : nullptr,
                             staged || (hstaged && !htr) ? otile + (m - m0) * OPITCH + (n - n0) : nullptr, resid_later, true,
                             hstaged && htr ? otile : nullptr, m - m0, n - n0, BM + 8, pb[i]);
            }
        }
    if (staged) copy_out_tile<BM, BN>(g, otile, OPITCH, m0, n0, z, resid_later);
    else if (hstaged) copy_out_heads<BM, BN>(g, otile, m0, n0, htr);
}

// =================================================================================================
// 3x3 convolution with an LDS-staged halo ("halo"): stride 1, no up-sampling, image width a multiple of 16 and a
// whole number of M tiles per image.  An M tile is BM/W consecutive image rows; for each 64-channel chunk of the input
// the (BM/W + 2) x (W + 2) pixel neighbourhood is brought into LDS ONCE (zero padding = out-of-range DMA) and all nine
// taps read their shifted A fragments from it, so the LDS fill per (tap, chunk) phase is only the weight tile:
// BN x 128 B instead of (BM + BN) x 128 B.  The plain kernel above is capped by that fill bandwidth.
// Weights stay double buffered per tap; the K order (ky, kx, cin) of the packed filter bank is unchanged.
// =================================================================================================
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void gemm_halo_kernel(const GemmArgs g) {
    typedef bf16 T;
    constexpr int BK = 64;
    constexpr int W_IT = BN / 32;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MF = WTM / 16, NF = WTN / 16;
    constexpr int WSTAGE = BN * 128;
    constexpr int PIT_MAX = (BM >= 256 ? 13 : (BM >= 128 ? 9 : 6));  // 16-byte transfers per thread for the largest supported patch
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.y;
    if (g.dbg & 2048) lds_poison(smem, (int)blockDim.x);
    const int Wd = g.Win, Hd = g.Hin;
    const int TH = BM / Wd, PW = Wd + 2;
    const int npix = (TH + 2) * PW;
    const int pit = (npix * 8 + 255) / 256;
    char* patch = smem;
    char* wbase = smem + pit * 4096;  // whole DMA rounds: lanes past the last patch pixel still write (zeros)

    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = g.M / BM;
    const int nblk = ntn * ntm;
    int logical;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // grouped order: within a group of `group_m` M tiles walk M first, so the workgroups that run together on an XCD share
    // their weight tiles (group_m ways) as well as their activation tiles through that XCD's L2 (N-first order re-streamed
    // the whole weight matrix once per M tile: 3-6x the algorithmic bytes on the wide-N and deep-K layers)
    int tn, tm;
    {
        const int GM = g.group_m > 1 ? g.group_m : 1;
        const int per_group = GM * ntn;
        const int grp = logical / per_group;
        const int first_m = grp * GM;
        const int gsz = min(GM, ntm - first_m);
        const int within = logical - grp * per_group;
        tm = first_m + within % gsz;
        tn = within / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int img = m0 / (Hd * Wd);
    const int y0 = (m0 - img * Hd * Wd) / Wd;  // first output row of the tile inside its image

    const T* a0p = reinterpret_cast<const T*>(g.a0);
    const T* a1p = reinterpret_cast<const T*>(g.a1);
    const T* wp = reinterpret_cast<const T*>(g.w);
    const long long a_rows = (long long)g.B * Hd * Wd;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)min((long long)g.N * g.K * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(a0p, (unsigned)min(a_rows * g.lda0 * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(a1p ? (const void*)a1p : (const void*)a0p,
                                                 a1p ? (unsigned)min(a_rows * g.lda1 * 2, 0x7FFFFFFFll) : 0u);

    // weight rows: as in the plain kernel
    const int lrow = tid >> 3, pch = tid & 7;
    unsigned wvo[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int row = it * 32 + lrow;
        const int n = n0 + row;
        const int c = pch ^ (row & 7);
        wvo[it] = n < g.N ? (unsigned)(((size_t)n * g.K + c * 8) * 2) : BL_OOB;
    }
    // patch transfers: transfer t = it*256 + tid carries chunk (t & 7) of patch pixel (t >> 3)
    unsigned pvo0[PIT_MAX], pvo1[PIT_MAX];
#pragma unroll
    for (int it = 0; it < PIT_MAX; ++it) {
        const int t = it * 256 + tid;
        const int pp = t >> 3;
        const int c = (t & 7) ^ (pp & 7);
        const int hy = pp / PW, hx = pp - hy * PW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = pp < npix && iy >= 0 && iy < Hd && ix >= 0 && ix < Wd;
        const unsigned pix = (unsigned)((img * Hd + iy) * Wd + ix);
        pvo0[it] = ok ? (pix * (unsigned)g.lda0 + (unsigned)c * 8u) * 2u : BL_OOB;
        pvo1[it] = ok ? (pix * (unsigned)g.lda1 + (unsigned)c * 8u) * 2u : BL_OOB;
    }

    const int Ct = g.c0 + g.c1;
    const int nch = Ct / BK;                       // 64-channel chunks over both sources
    const int per = (nch + g.splitk - 1) / g.splitk;
    const int ch_beg = split * per, ch_end = min(nch, ch_beg + per);

    auto stage_patch = [&](int ch) {
        const int cc = ch * BK;
        const bool second = cc >= g.c0;
        const unsigned chb = (unsigned)(second ? cc - g.c0 : cc) * 2;
#pragma unroll
        for (int it = 0; it < PIT_MAX; ++it) {
            if (it < pit) {
                if (!second) bl16(ra0, patch + (it * 256 + wave * 64) * 16, pvo0[it], chb);
                else bl16(ra1, patch + (it * 256 + wave * 64) * 16, pvo1[it], chb);
            }
        }
    };
    auto stage_w = [&](int ch, int tap, int buf) {
        char* sb = wbase + buf * WSTAGE;
        const unsigned kb = (unsigned)(tap * Ct + ch * BK) * 2;
#pragma unroll
        for (int it = 0; it < W_IT; ++it) bl16(rw, sb + (it * 256 + wave * 64) * 16, wvo[it], kb);
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave / WGN) * WTM, wn0 = (wave % WGN) * WTN;
    const int fr = lane & 15, fg = lane >> 4;
    int prow[MF];  // patch pixel index of this lane's row for tap (0, 0)
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int p = wm0 + j * 16 + fr;
        const int ty = p / Wd, tx = p - ty * Wd;
        prow[j] = ty * PW + tx;
    }

    auto compute = [&](int buf, int tap) {
        const char* sw = wbase + buf * WSTAGE;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = ky * PW + kx;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int physw = ((kk * 4 + fg) ^ (fr & 7)) * 16;
            bf16x8 wf[NF], af[MF];
#pragma unroll
            for (int i = 0; i < NF; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sw + (wn0 + i * 16 + fr) * 128 + physw);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int pr = prow[j] + toff;
                af[j] = *reinterpret_cast<const bf16x8*>(patch + pr * 128 + (((kk * 4 + fg) ^ (pr & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    };

    if (ch_beg < ch_end) {
        int cur = 0;
        if (g.pretouch) pretouch_weights(rw, (long long)g.N * g.K * 2, wbase + WSTAGE, wave, lane, g.pretouch);
        stage_w(ch_beg, 0, 0);
        for (int ch = ch_beg; ch < ch_end; ++ch) {
            stage_patch(ch);  // the previous chunk's last tap ended with a barrier: the patch buffer is free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (g.dbg & 1024) {
                // probe (MRISR_GEMM_FLAGS=1024; results are wrong): what GroupNorm + SiLU applied in the CONSUMER would cost - one LDS -> VALU -> LDS pass over
                // the staged patch per 64-channel chunk (scale / shift per channel from LDS-resident vectors would come on top)
                for (int t = tid; t < npix * 8; t += 256) {
                    bf16x8 v = *reinterpret_cast<bf16x8*>(patch + t * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e] * 1.0009765625f + 0.03125f;
                        v[e] = (bf16)(f / (1.0f + __expf(-f)));
                    }
                    *reinterpret_cast<bf16x8*>(patch + t * 16) = v;
                }
                __syncthreads();
            }
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) stage_w(ch, tap + 1, cur ^ 1);
                else if (ch + 1 < ch_end) stage_w(ch + 1, 0, cur ^ 1);
                compute(cur, tap);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
        }
    }

    if (g.splitk > 1) {
        float* part = g.partial + (size_t)split * (size_t)g.M * g.N;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wm0 + j * 16 + fr;
                const int n = n0 + wn0 + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4<float>(part + (size_t)m * g.N + n, v);
                }
            }
        if constexpr (NF * MF > 20) return;
        else {
            if (!g.sk_counters) return;
            if (!splitk_last_arriver<NF, MF>(g, acc, blockIdx.x, g.partial, m0 + wm0, n0 + wn0, fr, fg, smem)) return;
        }
    }
    constexpr int OPITCH = BN + 8;  // staged row output, as in gemm_bl_kernel (patch + weight stages are drained)
    T* otile = reinterpret_cast<T*>(smem);
    const bool staged = g.out_mode == OUT_ROWS && (g.ldo & 7) == 0 && (size_t)BM * OPITCH * 2 <= (size_t)pit * 4096 + 2 * WSTAGE && g.stage_out > 1;
    const bool resid_later = staged && g.resid != nullptr && (g.ldr & 7) == 0 && g.stage_out < 3;
    float pb[NF][4];
    preload_cols<NF>(g, n0 + wn0 + fg * 4, pb);
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm0 + j * 16 + fr;
            const int n = n0 + wn0 + i * 16 + fg * 4;
            if (m < g.M && n < g.N) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue4<T>(g, 0, m, n, v, nullptr, nullptr, staged ? otile + (m - m0) * OPITCH + (n - n0) : nullptr, resid_later, true, nullptr,
                             0, 0, 0, pb[i]);
            }
        }
    if (staged) copy_out_tile<BM, BN>(g, otile, OPITCH, m0, n0, 0, resid_later);
}

// =================================================================================================
// Halo conv kernel, 256-row tiles ("halo8"): eight waves (4 x 2, a wave: 64 x BN/2 as in the 128 x 160 halo kernel) share ONE
// weight tile per (tap, chunk) phase.  The 128-row kernel runs two workgroups per CU that each fetch their own copy of the
// weight tile - 87 % of its LDS fill - and keeps one 20 KB stage per workgroup in flight, which at ~1.2 us of L2 latency
// under load caps the fill near 33 GB/s per CU: 3,500 cycles per phase against 1,290 cycles of MFMA work (level-0 convs at
// 0.9 PF).  Here the weight fill per MAC is halved and the LDS the second workgroup's copies occupied becomes a ring of
// NSTAGE weight stages, filled NSTAGE - 1 phases ahead with counted waits (every wave issues the same number of DMA
// instructions per stage, none fully out of range).  The patch (TH + 2 rows of W + 2 pixels x 64 channels) is single
// buffered: its fill is exposed once per 9 phases.
// =================================================================================================
template <int BN, int NSTAGE>
__global__ __launch_bounds__(512, 2) void gemm_halo8_kernel(const GemmArgs g) {
    typedef bf16 T;
    constexpr int BM = 256, NT = 512, BK = 64, WGN = 2;
    constexpr int WTM = 64, WTN = BN / WGN;
    constexpr int MF = WTM / 16, NF = WTN / 16;
    constexpr int W_IT = (BN * 8 + NT - 1) / NT;   // DMA rounds (one instruction per wave each) per weight stage
    constexpr int WSTAGE = W_IT * NT * 16;
    constexpr int PIT_MAX = 7;                     // 16-byte transfers per thread for the largest supported patch (W = 64: 6 x 66 pixels)
    constexpr int D = NSTAGE - 1;                  // prefetch distance in phases
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 weight stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.y;
    if (g.dbg & 2048) lds_poison(smem, (int)blockDim.x);
    const int Wd = g.Win, Hd = g.Hin;
    const int TH = BM / Wd, PW = Wd + 2;
    const int npix = (TH + 2) * PW;
    const int pit = (npix * 8 + NT - 1) / NT;
    char* patch = smem;
    char* wbase = smem + pit * (NT * 16);

    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = g.M / BM;
    const int nblk = ntn * ntm;
    int logical;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tn, tm;
    {
        const int GM = g.group_m > 1 ? g.group_m : 1;
        const int per_group = GM * ntn;
        const int grp = logical / per_group;
        const int first_m = grp * GM;
        const int gsz = min(GM, ntm - first_m);
        const int within = logical - grp * per_group;
        tm = first_m + within % gsz;
        tn = within / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int img = m0 / (Hd * Wd);
    const int y0 = (m0 - img * Hd * Wd) / Wd;

    const T* a0p = reinterpret_cast<const T*>(g.a0);
    const T* a1p = reinterpret_cast<const T*>(g.a1);
    const T* wp = reinterpret_cast<const T*>(g.w);
    const long long a_rows = (long long)g.B * Hd * Wd;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)min((long long)g.N * g.K * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(a0p, (unsigned)min(a_rows * g.lda0 * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(a1p ? (const void*)a1p : (const void*)a0p,
                                                 a1p ? (unsigned)min(a_rows * g.lda1 * 2, 0x7FFFFFFFll) : 0u);

    // weight rows: transfer t = it * 512 + tid carries chunk (t & 7) of tile row (t >> 3); rows past the tile / past N are
    // clamped (a copy lands in the stage's unused tail or in columns the epilogue never stores): no out-of-range instruction
    const int lrow = tid >> 3, pch = tid & 7;
    unsigned wvo[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int row = it * (NT / 8) + lrow;
        const int n = min(n0 + min(row, BN - 1), g.N - 1);
        const int c = pch ^ (row & 7);
        wvo[it] = (unsigned)(((size_t)n * g.K + c * 8) * 2);
    }
    unsigned pvo0[PIT_MAX], pvo1[PIT_MAX];
#pragma unroll
    for (int it = 0; it < PIT_MAX; ++it) {
        const int t = it * NT + tid;
        const int pp = t >> 3;
        const int c = (t & 7) ^ (pp & 7);
        const int hy = pp / PW, hx = pp - hy * PW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = pp < npix && iy >= 0 && iy < Hd && ix >= 0 && ix < Wd;
        const unsigned pix = (unsigned)((img * Hd + iy) * Wd + ix);
        pvo0[it] = ok ? (pix * (unsigned)g.lda0 + (unsigned)c * 8u) * 2u : BL_OOB;
        pvo1[it] = ok ? (pix * (unsigned)g.lda1 + (unsigned)c * 8u) * 2u : BL_OOB;
    }

    const int Ct = g.c0 + g.c1;
    const int nch = Ct / BK;
    const int per = (nch + g.splitk - 1) / g.splitk;
    const int ch_beg = split * per, ch_end = min(nch, ch_beg + per);
    const int P = (ch_end - ch_beg) * 9;           // phases of this workgroup: (chunk, tap) pairs

    auto stage_patch = [&](int ch) {
        const int cc = ch * BK;
        const bool second = cc >= g.c0;
        const unsigned chb = (unsigned)(second ? cc - g.c0 : cc) * 2;
#pragma unroll
        for (int it = 0; it < PIT_MAX; ++it) {
            if (it < pit) {
                if (!second) bl16(ra0, patch + (it * NT + wave * 64) * 16, pvo0[it], chb);
                else bl16(ra1, patch + (it * NT + wave * 64) * 16, pvo1[it], chb);
            }
        }
    };
    auto stage_w = [&](int p) {  // phase p -> ring slot p % NSTAGE
        const int ch = ch_beg + p / 9, tap = p - (p / 9) * 9;
        char* sb = wbase + (p % NSTAGE) * WSTAGE;
        const unsigned kb = (unsigned)(tap * Ct + ch * BK) * 2;
#pragma unroll
        for (int it = 0; it < W_IT; ++it) bl16(rw, sb + (it * NT + wave * 64) * 16, wvo[it], kb);
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave / WGN) * WTM, wn0 = (wave % WGN) * WTN;
    const int fr = lane & 15, fg = lane >> 4;
    int prow[MF];
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int p = wm0 + j * 16 + fr;
        const int ty = p / Wd, tx = p - ty * Wd;
        prow[j] = ty * PW + tx;
    }

    auto compute = [&](int p) {
        const char* sw = wbase + (p % NSTAGE) * WSTAGE;
        const int tap = p - (p / 9) * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = ky * PW + kx;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int physw = ((kk * 4 + fg) ^ (fr & 7)) * 16;
            bf16x8 wf[NF], af[MF];
#pragma unroll
            for (int i = 0; i < NF; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sw + (wn0 + i * 16 + fr) * 128 + physw);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int pr = prow[j] + toff;
                af[j] = *reinterpret_cast<const bf16x8*>(patch + pr * 128 + (((kk * 4 + fg) ^ (pr & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    };

    if (P > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < P) stage_w(d);
        stage_patch(ch_beg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int p = 0; p < P; ++p) {
            if (p + D < P) stage_w(p + D);  // into the slot phase p - 1 read: free since the barrier that ended it
            compute(p);
            const int tap = p - (p / 9) * 9;
            // RAW barriers in this loop (see gemm_bl_kernel's ring): __syncthreads() would put a vmcnt(0) in front of every s_barrier
            // and drain the weight ring each phase - which is what rounds 1-2 measured as "ring depth makes no difference"
            if (tap == 8 && p + 1 < P) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // every wave is done with this chunk's patch
                stage_patch(ch_beg + (p + 1) / 9);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patch pieces are the youngest: everything has landed
            } else {
                // phase p + 1's weights must have landed; the stages issued after them (phases p + 2 .. p + D) may stay in flight
                const int young = min(D - 1, P - 2 - p);
                if (young >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W_IT) : "memory");
                else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_IT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __syncthreads();  // (drained) the epilogue reuses the stage buffers
    }

    if (g.splitk > 1) {
        float* part = g.partial + (size_t)split * (size_t)g.M * g.N;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wm0 + j * 16 + fr;
                const int n = n0 + wn0 + i * 16 + fg * 4;
                if (m < g.M && n < g.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4<float>(part + (size_t)m * g.N + n, v);
                }
            }
        return;
    }
    constexpr int OPITCH = BN + 8;
    T* otile = reinterpret_cast<T*>(smem);
    const bool staged = g.out_mode == OUT_ROWS && (g.ldo & 7) == 0 && (size_t)BM * OPITCH * 2 <= (size_t)pit * (NT * 16) + NSTAGE * WSTAGE && g.stage_out > 1;
    const bool resid_later = staged && g.resid != nullptr && (g.ldr & 7) == 0 && g.stage_out < 3;
    float pb[NF][4];
    preload_cols<NF>(g, n0 + wn0 + fg * 4, pb);
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm0 + j * 16 + fr;
            const int n = n0 + wn0 + i * 16 + fg * 4;
            if (m < g.M && n < g.N) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue4<T>(g, 0, m, n, v, nullptr, nullptr, staged ? otile + (m - m0) * OPITCH + (n - n0) : nullptr, resid_later, true, nullptr,
                             0, 0, 0, pb[i]);
            }
        }
    if (staged) copy_out_tile<BM, BN, NT>(g, otile, OPITCH, m0, n0, 0, resid_later);
}

// =================================================================================================
// Weight-stationary kernel for the short-K projections ("ws"): K = 32*KS fits a whole [16*NB rows][K] weight slab in
// LDS (K = 320, BN = 160: 100 KB), so a workgroup loads its slab ONCE and then streams row blocks of the activation
// through it.  The A fragments never touch LDS: every wave loads its own 32 rows straight from global memory into MFMA
// operand layout (16 B per lane), a whole row block (all KS k-steps) at a time and one block ahead of the MFMAs, so
// the only exposed latencies are the first block's and HBM streams continuously - the plain kernel above runs these
// shapes as ONE lock-step wave of workgroups (five K tiles each: prologue, first-load latency and epilogue are not
// amortised).  One workgroup per CU (the slab fills the LDS), grid = slabs x row groups.
// Requirements: plain rows, no split, M % 128 == 0, N % (16*NB) == 0, K == 32*KS.  LORA as in gemm_bl_kernel.
// =================================================================================================
template <int KS, int NB, bool LORA>
__global__ __launch_bounds__(256) void gemm_ws_kernel(const GemmArgs g, int row_groups) {
    typedef bf16 T;
    constexpr int K = KS * 32;
    constexpr int BN = NB * 16;
    constexpr int WP = K * 2 + 16;  // padded slab row pitch (bytes): conflict-free 16-byte fragment reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                    // [BN (+16 adapter rows)][WP]
    float* zl = reinterpret_cast<float*>(smem + (BN + (LORA ? 16 : 0)) * WP);  // per wave [16][16] f32

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int slab = blockIdx.x / row_groups, rg = blockIdx.x - slab * row_groups;
    const int n0 = slab * BN;
    const T* ap = reinterpret_cast<const T*>(g.a0);
    const T* wp = reinterpret_cast<const T*>(g.w);

    // ---- the slab: global -> registers -> LDS, once ----
    {
        constexpr int CPR = K / 8;  // 16-byte chunks per row
        const int rows = BN + (LORA ? 16 : 0);
        for (int c = tid; c < rows * CPR; c += 256) {
            const int row = c / CPR, ch = c - row * CPR;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (row < BN) v = *reinterpret_cast<const bf16x8*>(wp + (size_t)(n0 + row) * K + ch * 8);
            else if (LORA && row - BN < g.lora_R) v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(g.lora_a) + (size_t)(row - BN) * K + ch * 8);
            *reinterpret_cast<bf16x8*>(wl + row * WP + ch * 16) = v;
        }
    }
    // ---- row blocks of 64 (a wave: one 16-row fragment), strided over the row groups; one block prefetched ahead ----
    const int nblk = g.M / 64;
    bf16x8 a_cur[KS], a_nxt[KS];
    auto load_a = [&](bf16x8 (&dst)[KS], int blk) {
        const T* base = ap + (size_t)(blk * 64 + wave * 16 + fr) * g.lda0 + fg * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) dst[kk] = *reinterpret_cast<const bf16x8*>(base + kk * 32);
    };
    int blk = rg;
    if (blk < nblk) load_a(a_cur, blk);
    __syncthreads();  // slab visible
    float* zw = zl + wave * 16 * 16;
    for (; blk < nblk; blk += row_groups) {
        const int nb_next = blk + row_groups;
        if (nb_next < nblk) load_a(a_nxt, nb_next);
        f32x4 acc[NB];
        f32x4 zacc;
        // slab fragments software-pipelined one k-step ahead; the scheduling barriers keep the compiler from hoisting
        // all KS*NB LDS reads to the top (it would need 4x the register file)
        constexpr int NW = NB + (LORA ? 1 : 0);
        bf16x8 wf[2][NW];
        auto load_w = [&](bf16x8 (&dst)[NW], int kk) {
#pragma unroll
            for (int i = 0; i < NW; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(wl + (i * 16 + fr) * WP + (kk * 32 + fg * 8) * 2);
        };
        load_w(wf[0], 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) load_w(wf[(kk + 1) & 1], kk + 1);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const f32x4 c = kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][i], a_cur[kk], c, 0, 0, 0);
            }
            if (LORA) {  // the 16 adapter rows sit right behind the slab: fragment index NB
                const f32x4 c = kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : zacc;
                zacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][NW - 1], a_cur[kk], c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int m = blk * 64 + wave * 16 + fr;
        if (LORA) {  // z[q = 4fg + r][m = fr] -> this wave's LDS rows; every lane then reads its own row
            *reinterpret_cast<f32x4*>(zw + fr * 16 + fg * 4) = zacc;
            if (g.lora_zout && slab == 0 && fg * 4 < g.lora_R) *reinterpret_cast<f32x4*>(g.lora_zout + (size_t)m * g.lora_R + fg * 4) = zacc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes have landed (wave-private rows)
        }
        if (g.act == ACT_GEGLU) {
            if constexpr (NB % 2 == 0) {
#pragma unroll
                for (int i = 0; i < NB; i += 2) {
                    float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    float vg[4] = {acc[i + 1][0], acc[i + 1][1], acc[i + 1][2], acc[i + 1][3]};
                    epilogue4<T>(g, 0, m, n0 + i * 16 + fg * 4, v, vg);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                epilogue4<T>(g, 0, m, n0 + i * 16 + fg * 4, v, nullptr, LORA ? zw + fr * 16 : nullptr);
            }
        }
        if (nb_next < nblk) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a_cur[kk] = a_nxt[kk];
        }
    }
}

// =================================================================================================
// Row-panel kernel for the short-K row GEMMs of the transformer blocks ("rp"): K = 32*KS <= 640, M large.
//
// The tiled kernels above run these shapes far below BOTH rooflines (K = 320, M = 32,768: 6.7 GFLOP + 42 MB in ~30 us):
// five to ten K tiles per workgroup, each a full load -> drain -> barrier -> 16..32 MFMA round trip, every workgroup of
// the grid in lock-step, and the activation tile re-fetched into LDS by every N tile.  Here the roles are split by
// operand instead:
//   * a workgroup owns a PANEL of 128 rows (a wave: 32 rows = two 16-row MFMA fragments) and keeps the WHOLE K extent of
//     its rows in REGISTERS in MFMA operand layout (2 x KS x 4 VGPRs per lane), loaded once, straight from global memory
//     with buffer loads (out-of-range rows read zeros) - the activations never pass through LDS;
//   * the weights stream through LDS in chunks of BN = 16*NF output columns x the whole K (<= 40 KB), double buffered,
//     by LDS-DMA; a chunk is ONE barrier interval of KS*2*NF MFMAs per wave (80 at K = 320), so the per-K-tile
//     drain/barrier of the tiled kernels disappears and the next chunk's load overlaps a long compute phase;
//   * two fragments of rows per wave halve the LDS fragment reads per MFMA (the earlier weight-stationary kernel, one
//     fragment per wave, was LDS-read bound: profiles/r01b_ws_sweep.log);
//   * optional PROLOGUE on the register-resident rows: LayerNorm (statistics over the row with two cross-lane adds,
//     affine from global) - the separate LayerNorm launch and its HBM round trip of the row tensor disappear;
//   * grid = panels x ysplit: the N range is split over `ysplit` workgroups so that ~2 workgroups per CU exist.
// Epilogues as in gemm_bl_kernel (bias, residual, GEGLU, head-major Q/K/V^T, staged row stores, rank-4 LoRA with the
// down-projection computed in-kernel from the resident rows and the up-projection as one more MFMA K step).
// =================================================================================================
typedef __attribute__((ext_vector_type(4))) unsigned rp_u4;
__device__ __forceinline__ bf16x8 rp_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// FP8 (BASELINE configs[4]: fp8 projection GEMMs): the same kernel with OCP e4m3 operands on v_mfma_f32_16x16x32_fp8_fp8 -
//   * weights arrive pre-quantised, one f32 scale per output channel (g.w8 / g.w_scale, packed at finalize);
//   * the resident rows are loaded (and LayerNorm'ed) in bf16 as before, then quantised IN REGISTERS with one scale per row
//     (amax over the row by two cross-lane maxima): 8 fp8 per lane and k-step = 2 VGPRs instead of 4;
//   * weight chunks are half the bytes (BN x K bytes: 20 KB at K = 320), fragment reads are ds_read_b64;
//   * the accumulators are rescaled by row scale x channel scale after the K loop, before the (bf16) LoRA up-projection step
//     and the bias; everything after that is the bf16 kernel's epilogue.
typedef long rp_f8x8;  // 8 fp8 (one MFMA operand)
__device__ __forceinline__ rp_f8x8 rp_quant8(const bf16x8& x, float inv_scale) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[0] * inv_scale, (float)x[1] * inv_scale, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[2] * inv_scale, (float)x[3] * inv_scale, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[4] * inv_scale, (float)x[5] * inv_scale, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[6] * inv_scale, (float)x[7] * inv_scale, hi, true);
    return (rp_f8x8)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

// NW: waves per workgroup = 32-row groups per panel.  4 (128-row panels, 2 workgroups = 8 waves per CU) or 2 (64-row panels: the
// two co-resident workgroups put ONE wave on each SIMD, which may then use the whole 512-register file - K = 640 rows (160
// registers of fragments per lane) fit without spilling, and a 64 x 640 panel is 80 KB = the weight buffers, so it can come in
// through LDS as whole pieces too).
#ifndef RP_READS_FIRST
#define RP_READS_FIRST 1
#endif
#ifndef RP_INTERLEAVE
#define RP_INTERLEAVE 1  // the next K step's fragment reads between this step's MFMA pairs instead of in front of them: 10.182 -> 10.167 ms per step (profiles/r04c_rp_il_ab.log)
#endif
template <int KS, int NF, bool LORA, int PRO, bool FP8, int NW>
__global__ __launch_bounds__(64 * NW, NW == 2 ? 1 : 2) void gemm_rp_kernel(const GemmArgs g) {
    typedef bf16 T;
    constexpr int K = KS * 32, BN = NF * 16, BM = 32 * NW, MF = 2;
    constexpr int EW = FP8 ? 1 : 2;             // bytes per weight element
    constexpr int CPR = KS * 4 / (FP8 ? 2 : 1); // 16-byte chunks per weight row
    constexpr int CHUNK = BN * K * EW;          // bytes of one weight chunk in LDS
    constexpr int SWZ = FP8 ? 3 : 7;            // 16-byte chunk swizzle: c ^ (f(row) & SWZ); fp8 rows are 320 / 640 B: f = row >> 2
    constexpr int PIECES = BN * CPR / 64;       // 1-KiB LDS-DMA pieces per chunk
    constexpr int PPW = PIECES / NW;            // pieces per wave
    static_assert((BN * CPR) % 64 == 0, "whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nchunks = (g.N + BN - 1) / BN;
    const int per = (nchunks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c_beg = blockIdx.y * per, c_end = min(nchunks, c_beg + per);
    if (c_beg >= c_end) return;
    if (g.dbg & 2048) lds_poison(smem, 64 * NW);

    const T* ap = reinterpret_cast<const T*>(g.a0);
    const void* wp = FP8 ? g.w8 : g.w;
    const void* lp = FP8 ? g.lora_a8 : g.lora_a;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(ap, (unsigned)min((long long)g.M * g.lda0 * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)min((long long)g.N * K * EW, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(LORA ? lp : wp, LORA ? (unsigned)((long long)g.lora_R * K * EW) : 0u);
    auto swz = [](int row) { return FP8 ? ((row >> 2) & 3) : (row & 7); };

    // ---- weight chunk DMA geometry: LDS position L (16-byte units) = row * CPR + (c ^ swz(row)) ----
    static_assert(PIECES % NW == 0, "every wave issues the same number of DMA pieces (no control flow around them: the compiler's\n"
                                   "wait-count model turns conditional VMEM issue into vmcnt(0) drains)");
    // All workgroups of the grid stream the same weight chunks at about the same time; walked in the same order by the 32 CUs of an XCD
    // the requests pile up on one L2 channel after the other (clock stamps in xtail.hip: a wave's ten DMA instructions took 2,100 cycles
    // to issue, 1,500 with the rotation).  g.dbg & 4096 turns the rotation off (MRISR_RP_ROT=0).
    unsigned wvo[PPW];
    int ldo[PPW];
    const bool rot = !(g.dbg & 4096);
    const int rot_w = rot ? (int)(blockIdx.x % NW) : 0, rot_q = rot ? (int)((blockIdx.x / NW + blockIdx.y) % PPW) : 0;
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        int q = p + rot_q;
        if (q >= PPW) q -= PPW;
        const int piece = q * NW + (wave + rot_w) % NW;
        const int L = piece * 64 + lane;
        const int row = L / CPR, cs = L - row * CPR;
        const int c = cs ^ swz(row);
        wvo[p] = (unsigned)(row * K * EW + c * 16);
        ldo[p] = __builtin_amdgcn_readfirstlane(piece * 1024);
    }
    // `live` false: the chunk does not exist - every lane's offset lies beyond num_records, the DMA writes zeros (harmless)
    auto stage_w = [&](int chunk, int buf, bool live) {
        char* sb = smem + buf * CHUNK;
        const unsigned base = live ? (unsigned)chunk * (unsigned)CHUNK : 0xC0000000u;  // rows past N also lie beyond num_records
#pragma unroll
        for (int p = 0; p < PPW; ++p) bl16(rw, sb + ldo[p], wvo[p], base);
    };
    // the same, the pieces of K step kk only (default; g.dbg & 8192 = all at the top of the chunk as before): the next chunk's DMA spread over this chunk's K loop - the wave waits in the VMEM queue behind running MFMAs (bench step 9.80 -> 9.72 ms)
    const bool spread = !(g.dbg & 8192);
    auto stage_step = [&](int chunk, int buf, bool live, int kk) {
        char* sb = smem + buf * CHUNK;
        const unsigned base = live ? (unsigned)chunk * (unsigned)CHUNK : 0xC0000000u;
#pragma unroll
        for (int p = kk * PPW / KS; p < (kk + 1) * PPW / KS; ++p) bl16(rw, sb + ldo[p], wvo[p], base);
    };

    // ---- the panel rows: registers, MFMA second-operand layout (row = fr of fragment j, k = 32 kk + 8 fg ..) ----
    // A_VIA_LDS (K = 320): the wave's 32 rows (20 KB) come in as whole 1-KiB LDS-DMA pieces into a wave-private region of the
    // (still empty) weight buffers and are then read out in fragment layout.  Loading the fragments straight from global
    // memory fetches every 128-byte line of a row twice as two 64-byte halves (k-steps 2t and 2t + 1): the load phase ran at
    // 3.8 TB/s of L2 traffic for 21 MB of rows (tools/probes/rp_probe.sh, flags 448).
    constexpr bool A_VIA_LDS = NW * 32 * K * 2 <= (FP8 && KS == 10 ? 81920 : 2 * CHUNK);
    // (rows straight into registers: the buffers are free from the start - pre-touch the weights now, before the row fragments are live)
    if (!A_VIA_LDS && g.pretouch > 0) pretouch_weights(rw, (long long)g.N * K * EW, smem + 2 * CHUNK - 8192, wave, lane, g.pretouch);
    bf16x8 af[MF][KS];
    if constexpr (A_VIA_LDS) {
        constexpr int ACPR = K / 8;                       // 16-byte chunks per row
        constexpr int APIECES = 32 * ACPR / 64;           // pieces per wave
        char* areg = smem + wave * (32 * K * 2);
#pragma unroll
        for (int p = 0; p < APIECES; ++p) {
            const int L = p * 64 + lane;
            const int row = L / ACPR, cs = L - row * ACPR;
            const int c = cs ^ (row & 7);
            const int m = m0 + wave * 32 + row;
            const unsigned vo = m < g.M ? (unsigned)(((size_t)m * g.lda0 + c * 8) * 2) : BL_OOB;
            bl16(ra, areg + p * 1024, vo, 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own pieces (the region is wave-private: no barrier)
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                af[j][kk] = *reinterpret_cast<const bf16x8*>(areg + (j * 16 + fr) * (K * 2) + (((kk * 4 + fg) ^ (fr & 7)) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();  // every wave has its rows in registers: the buffers may receive weights
    } else {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wave * 32 + j * 16 + fr;
            const unsigned vo = m < g.M ? (unsigned)(((size_t)m * g.lda0 + fg * 8) * 2) : BL_OOB;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) af[j][kk] = rp_load16(ra, vo, (unsigned)(kk * 64));
        }
    }
    // cold weights (pretouch_weights): the share lands in the last 8 KB of buffer 1 - behind the adapter rows, overwritten by chunk c_beg + 1
    if (A_VIA_LDS && g.pretouch > 0) pretouch_weights(rw, (long long)g.N * K * EW, smem + 2 * CHUNK - 8192, wave, lane, g.pretouch);
    stage_w(c_beg, 0, true);
    if (LORA) {  // the adapters' A rows (R <= 16) borrow the front of buffer 1 until chunk c_beg + 1 is staged
        constexpr int LPIECES = 16 * CPR / 64;
#pragma unroll
        for (int p = 0; p < (LPIECES + NW - 1) / NW; ++p) {
            const int piece = p * NW + wave;
            const int L = piece * 64 + lane;
            const int row = L / CPR, cs = L - row * CPR;
            const int c = cs ^ swz(row);
            const unsigned vo = (piece < LPIECES && row < g.lora_R) ? (unsigned)(row * K * EW + c * 16) : BL_OOB;
            if (piece < LPIECES) bl16(rl, smem + CHUNK + piece * 1024, vo, 0u);
        }
    }
    // LayerNorm: gamma | beta (f32) into LDS by DMA, between the adapter rows and the pre-touch scratch of buffer 1.  The apply loop used to load
    // them from global memory k-step by k-step - KS serialized round trips, 7-14 us per launch, most of the prologue's cost.
    constexpr int LNP = (K * 4 + 1023) / 1024;         // 1-KiB pieces per vector
    constexpr int GB_OFF = 2 * CHUNK - 8192 - 6144;    // 6 KB: [gamma LNP KB][beta LNP KB]
    static_assert(2 * LNP <= 6 && GB_OFF >= CHUNK + 16 * K * EW, "LayerNorm vectors fit between the adapter rows and the scratch");
    if (PRO == 1) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(g.ln_gamma, (unsigned)(K * 4));
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.ln_beta, (unsigned)(K * 4));
#pragma unroll
        for (int p = 0; p < (2 * LNP + NW - 1) / NW; ++p) {
            const int piece = p * NW + wave;
            if (piece < LNP) bl16(rg, smem + GB_OFF + piece * 1024, (unsigned)(piece * 1024 + lane * 16), 0u);
            else if (piece < 2 * LNP) bl16(rb, smem + GB_OFF + piece * 1024, (unsigned)((piece - LNP) * 1024 + lane * 16), 0u);
        }
    }

    if (PRO == 1) {
        // LayerNorm over the K extent of each resident row: a row's K values live in the 4 lanes {fr + 16 fg'}.  ONE statistics pass
        // with sums shifted by the row's first element c (sum d, sum d^2 with d = x - c: no cancellation for rows with a large common
        // offset; the inputs carry 8 significant bits), then one apply pass.  `opaque` keeps the compiler from carrying the 160
        // unpacked f32 values of the first pass over to the second (it spilled hundreds of registers doing so).
        float mean[MF], rstd[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const float c0 = __shfl((float)af[j][0][0], fr);  // lane fr (fg = 0) holds the row's element 0
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)af[j][kk][e] - c0; s1 += d; s2 = fmaf(d, d, s2); }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float md = s1 * (1.0f / K);
            mean[j] = c0 + md;
            rstd[j] = rsqrtf(fmaxf(s2 * (1.0f / K) - md * md, 0.f) + g.ln_eps);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                rp_u4 w = __builtin_bit_cast(rp_u4, af[j][kk]);
                asm volatile("" : "+v"(w));
                af[j][kk] = __builtin_bit_cast(bf16x8, w);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // gamma / beta (every wave's pieces), and with them chunk c_beg: all due now anyway
        __syncthreads();
        const float* gp = reinterpret_cast<const float*>(smem + GB_OFF) + fg * 8;
        const float* bp = reinterpret_cast<const float*>(smem + GB_OFF + LNP * 1024) + fg * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {  // k outer: this k-step's gamma / beta serve both row fragments
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp + kk * 32), g1 = *reinterpret_cast<const f32x4*>(gp + kk * 32 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + kk * 32), b1 = *reinterpret_cast<const f32x4*>(bp + kk * 32 + 4);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
                    o[e] = (bf16)(((float)af[j][kk][e] - mean[j]) * rstd[j] * ga + be);
                }
                af[j][kk] = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- FP8: quantise the resident rows, one scale per row (amax / 448, the e4m3 maximum) ----
    rp_f8x8 a8[FP8 ? MF : 1][FP8 ? KS : 1];
    float sa[MF] = {1.f, 1.f};
    if (FP8) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            float am = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf((float)af[j][kk][e]));
            am = fmaxf(am, __shfl_xor(am, 16));
            am = fmaxf(am, __shfl_xor(am, 32));
            sa[j] = fmaxf(am, 1e-20f) * (1.0f / 448.0f);
            const float inv = 1.0f / sa[j];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a8[j][kk] = rp_quant8(af[j][kk], inv);
        }
    }
    // weight fragment of lane (fr, fg) for k-step kk out of an LDS image with row pitch K * EW (base: row 0 of the fragment)
    auto wfrag_off = [&](int kk) {
        if (FP8) return (((kk * 2 + (fg >> 1)) ^ swz(fr)) * 16) + (fg & 1) * 8;
        return ((kk * 4 + fg) ^ swz(fr)) * 16;
    };

    // ---- LoRA: z = x A^T from the resident rows (one 16-column "chunk"), kept in registers as the row operand of the
    // up-projection step: lane (fr, fg) holds z[m = fr][q = 4 fg + r], which it uses as k elements 0..3 of its k group
    bf16x8 zf[MF];
    if (LORA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sl = smem + CHUNK + fr * (K * EW);
        f32x4 zacc[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) zacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if constexpr (FP8) {
                const rp_f8x8 lf = *reinterpret_cast<const rp_f8x8*>(sl + wfrag_off(kk));
#pragma unroll
                for (int j = 0; j < MF; ++j) zacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lf, a8[j][kk], zacc[j], 0, 0, 0);
            } else {
                const bf16x8 lf = *reinterpret_cast<const bf16x8*>(sl + wfrag_off(kk));
#pragma unroll
                for (int j = 0; j < MF; ++j) zacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, af[j][kk], zacc[j], 0, 0, 0);
            }
        }
        if (FP8) {  // back to real units: row scale x the adapter rows' own scales (q = 4 fg + r)
            const f32x4 las = *reinterpret_cast<const f32x4*>(g.lora_a_scale + fg * 4);
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) zacc[j][r] *= sa[j] * las[r];
        }
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            zf[j] = bf16x8{(bf16)zacc[j][0], (bf16)zacc[j][1], (bf16)zacc[j][2], (bf16)zacc[j][3], (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            const int m = m0 + wave * 32 + j * 16 + fr;
            if (g.lora_zout && blockIdx.y == 0 && m < g.M && fg * 4 < g.lora_R)
                *reinterpret_cast<f32x4*>(g.lora_zout + (size_t)m * g.lora_R + fg * 4) = zacc[j];
        }
        __syncthreads();  // every wave is done with the adapter rows: buffer 1 may receive a weight chunk
    }

    // Outputs (rp_ok admits only these forms): rows [M][ldo] (+ residual rows), GEGLU rows, or head-major Q / K / V^T sections.
    // Every form leaves through a wave-private staging tile as 16-byte pieces.  The epilogue runs once per CHUNK (up to 20
    // times per workgroup), so it is straight-line code: the bias is the accumulators' initial value, the operand-mode
    // branching is per chunk, never per fragment (the generic epilogue4 - ~30 uniform branches and a global-load round trip
    // per fragment - made the first version of this kernel spend 2/3 of its time between K loops: tools/probes/).
    const bool has_resid = g.resid != nullptr;
    constexpr int OPITCH = BN + 8;      // wave-private row tile [32][OPITCH]
    constexpr int TPITCH = 32 + 8;      // wave-private transposed tile [BN][TPITCH] (V^T)
    constexpr int GP = BN / 2 + 8;      // GEGLU tile [32][GP]
    constexpr int WREG = CHUNK / NW;     // bytes of a wave's staging region inside the chunk's own (consumed) weight buffer
    static_assert(32 * OPITCH * 2 <= WREG && BN * TPITCH * 2 <= WREG && NW * WREG <= CHUNK, "staging regions");
    const int wm0 = wave * 32;
    const bool full_m = m0 + BM <= g.M;
    const bool heads = g.out_mode == OUT_HEADS;
    const bool geglu = g.act == ACT_GEGLU;
    // stores of the previous chunk's epilogue that may still be in flight at the top of the loop (per lane; -1 = unknown).  The
    // wait at the top must cover chunk c's LDS-DMA (older than those stores: vmcnt retires in order) but NOT the stores:
    // waiting for them would serialise every chunk on an HBM write round trip.
    int pending = 0;

    for (int c = c_beg; c < c_end; ++c) {
        const int buf = (c - c_beg) & 1;
        const int n0 = c * BN;
        // the chunk's bias (GEGLU: interleaved like the weight rows) = initial value of the accumulators; loaded first so that
        // the wait + barrier below cover its latency.  Columns past N (partial last chunk) are clamped: never stored.
        float pb[NF][4];
#pragma unroll
        for (int i = 0; i < NF; ++i) load4<float>(g.bias + min(n0 + i * 16 + fg * 4, g.N - 4), pb[i]);  // (launch_rp: never null)
        float sw[FP8 ? NF : 1][4];
        if (FP8) {
#pragma unroll
            for (int i = 0; i < NF; ++i) load4<float>(g.w_scale + min(n0 + i * 16 + fg * 4, g.N - 4), sw[i]);
        }
        // counted wait only when every DMA instruction of chunk c is in range (a fully out-of-range LDS-DMA instruction retires
        // out of order and would satisfy the count early) and the store count is known; the NF bias (+ NF scale) loads above are younger
        const int young = pending > 0 && n0 + BN <= g.N ? pending + (FP8 ? 2 * NF : NF) : 0;
        if (young == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (young == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (young == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (young == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (young == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (young == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk c has landed; every wave has left chunk c - 1 (its buffer, incl. the staging regions, is free)
        if (!spread) stage_w(c + 1, buf ^ 1, c + 1 < c_end);  // unconditional issue (see stage_w): the K loop's bias wait stays a counted one
        bf16x8 lbf[NF];
        if (LORA) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                lbf[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                const int n = n0 + i * 16 + fr;
                if (n < g.N && n / g.lora_secN == fg) {
                    const f32x4 l = *reinterpret_cast<const f32x4*>(g.lora_b + (size_t)n * 4);
                    lbf[i][0] = (bf16)l[0]; lbf[i][1] = (bf16)l[1]; lbf[i][2] = (bf16)l[2]; lbf[i][3] = (bf16)l[3];
                }
            }
        }
        f32x4 acc[NF][MF];
        const char* sw_l = smem + buf * CHUNK + fr * (K * EW);
        // weight fragments software-pipelined one k-step ahead; the scheduling barriers keep the compiler from hoisting all
        // KS*NF LDS reads above the MFMAs (it would need the whole register file)
        if constexpr (FP8) {
            rp_f8x8 wf[2][NF];
            auto load_w = [&](rp_f8x8 (&dst)[NF], int kk) {
                const int off = wfrag_off(kk);
#pragma unroll
                for (int i = 0; i < NF; ++i) dst[i] = *reinterpret_cast<const rp_f8x8*>(sw_l + i * 16 * K + off);
            };
            load_w(wf[0], 0);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (spread) { stage_step(c + 1, buf ^ 1, c + 1 < c_end, kk); __builtin_amdgcn_sched_barrier(0); }
                if (kk + 1 < KS) load_w(wf[(kk + 1) & 1], kk + 1);
                if (RP_READS_FIRST) __builtin_amdgcn_sched_barrier(0);  // else the scheduler sinks the prefetch to the end of the step
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        const f32x4 cz = kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[kk & 1][i], a8[j][kk], cz, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // real units: row scale x channel scale, then the bias (the bf16 kernel starts its accumulators at the bias instead)
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(acc[i][j][r], sa[j] * sw[i][r], pb[i][r]);
        } else {
            bf16x8 wf[2][NF];
            auto load_w = [&](bf16x8 (&dst)[NF], int kk) {
                const int off = wfrag_off(kk);
#pragma unroll
                for (int i = 0; i < NF; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(sw_l + i * 16 * (K * 2) + off);
            };
            load_w(wf[0], 0);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (spread) { stage_step(c + 1, buf ^ 1, c + 1 < c_end, kk); __builtin_amdgcn_sched_barrier(0); }
                if (kk + 1 < KS) load_w(wf[(kk + 1) & 1], kk + 1);
                if (RP_READS_FIRST && !RP_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);  // else the scheduler sinks the prefetch to the end of the step
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        const f32x4 cz = kk == 0 ? f32x4{pb[i][0], pb[i][1], pb[i][2], pb[i][3]} : acc[i][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][i], af[j][kk], cz, 0, 0, 0);
                    }
                if (RP_INTERLEAVE && kk + 1 < KS) {  // the next step's fragment reads between this step's MFMA pairs
#pragma unroll
                    for (int q = 0; q < NF; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g.dbg & 128) {  // probe: no epilogue / stores
            if (acc[0][0][0] == 1.2345e30f) reinterpret_cast<float*>(g.out)[0] = acc[1][1][1] + acc[NF - 1][0][2];
            pending = 0;
            continue;
        }
        if (LORA) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbf[i], zf[j], acc[i][j], 0, 0, 0);
        }
        // ---- epilogue of the chunk ----
        const bool full = full_m && n0 + BN <= g.N;
        T* wt = reinterpret_cast<T*>(smem + buf * CHUNK + wave * WREG);
        __syncthreads();  // every wave has finished reading this chunk's weights: its buffer becomes the staging area
        if (geglu) {
            if constexpr (NF % 2 == 0) {
                // u * gelu(gate) of the chunk's (u, gate) fragment pairs -> [32][BN/2] tile -> 16-byte pieces of the compacted rows
#pragma unroll
                for (int i = 0; i < NF; i += 2)
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[i][j][r] * gelu_erf_t<T>(acc[i + 1][j][r]));
                        *reinterpret_cast<bf16x4*>(wt + (j * 16 + fr) * GP + (i >> 1) * 16 + fg * 4) = o;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes (a wave's LDS accesses execute in order)
                constexpr int CPRG = BN / 16;
                T* ob = reinterpret_cast<T*>(g.out);
#pragma unroll
                for (int it = 0; it < NF / 2; ++it) {
                    const int idx = it * 64 + lane;
                    const int row = idx / CPRG, ch = idx - row * CPRG;
                    const int m = m0 + wm0 + row;
                    if (m >= g.M || n0 + ch * 16 >= g.N) continue;
                    *reinterpret_cast<bf16x8*>(ob + (size_t)m * g.ldo + (n0 >> 1) + ch * 8) = *reinterpret_cast<const bf16x8*>(wt + row * GP + ch * 8);
                }
                pending = full ? NF / 2 : -1;
            }
            continue;
        }
        bool htr = false;
        int sidx = 0;
        if (heads) {
            sidx = n0 / g.secC;
            htr = (sidx == 0 ? g.sec_tr[0] : (sidx == 1 ? g.sec_tr[1] : g.sec_tr[2])) != 0;
        }
        T* base = heads ? reinterpret_cast<T*>(sidx == 0 ? g.sec_ptr[0] : (sidx == 1 ? g.sec_ptr[1] : g.sec_ptr[2])) : reinterpret_cast<T*>(g.out);
        if (!htr) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const bf16x4 o = {(bf16)acc[i][j][0], (bf16)acc[i][j][1], (bf16)acc[i][j][2], (bf16)acc[i][j][3]};
                    *reinterpret_cast<bf16x4*>(wt + (j * 16 + fr) * OPITCH + i * 16 + fg * 4) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int CPRW = BN / 8;  // 16-byte pieces per row
#pragma unroll
            for (int it = 0; it < NF; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / CPRW, ch = idx - row * CPRW;
                const int m = m0 + wm0 + row, n = n0 + ch * 8;
                if (m >= g.M || n >= g.N) continue;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(wt + row * OPITCH + ch * 8);
                if (g.dbg & 512) { if (v[0] == (bf16)1.2345e30f) *reinterpret_cast<bf16x8*>(base) = v; continue; }  // probe: no global store
                if (!heads) {
                    if (has_resid) {  // residual read as whole rows too (may alias the output: same lane reads then writes)
                        // (batching these loads ahead of the adds, as copy_out_tile does, spills here: the row fragments are live)
                        const bf16x8 r = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(g.resid) + (size_t)m * g.ldr + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)r[e]);
                    }
                    *reinterpret_cast<bf16x8*>(base + (size_t)m * g.ldo + n) = v;
                } else {
                    const int cc = n - sidx * g.secC, h = cc / g.hd, dd = cc - h * g.hd;
                    const int b = m / g.ntok, tok = m - b * g.ntok;
                    *reinterpret_cast<bf16x8*>(base + (((size_t)b * g.nheads + h) * g.npad + tok) * g.dpad + dd) = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    unsigned w0, w1;
                    const int drow = exchange_tokens4(v, w0, w1);
                    *reinterpret_cast<uint2*>(wt + (i * 16 + fg * 4 + drow) * TPITCH + ((j * 16 + fr) & ~3)) = make_uint2(w0, w1);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < NF; ++it) {
                const int idx = it * 64 + lane;
                const int col = idx >> 2, tc = idx & 3;  // 4 pieces of 8 tokens per channel row
                const int m = m0 + wm0 + tc * 8, n = n0 + col;
                if (m >= g.M || n >= g.N) continue;
                const int cc = n - sidx * g.secC, h = cc / g.hd, dd = cc - h * g.hd;
                const int b = m / g.ntok, tok = m - b * g.ntok;
                *reinterpret_cast<bf16x8*>(base + (((size_t)b * g.nheads + h) * g.dpad + dd) * g.npad + tok) =
                    *reinterpret_cast<const bf16x8*>(wt + col * TPITCH + tc * 8);
            }
        }
        pending = full ? NF : -1;
    }
}

// =================================================================================================
// Fused feed-forward of a transformer block at C = 320 ("mlp"):  out = resid + FF2(GEGLU(FF1(LayerNorm(x)))) in ONE kernel.
// The row-panel idea carried one GEMM further: a workgroup owns 128 rows (a wave: 32), keeps LayerNorm(x) in registers as
// MFMA fragments (80 VGPRs) AND the whole 32 x 320 output tile of FF2 as accumulators (160 VGPRs) - one wave per SIMD, so a wave
// may use the whole 512-register file.  The hidden activation never exists in memory: per chunk of 32 hidden units
//   FF1:  64 interleaved (u, gate) weight rows x K = 320 from LDS  -> acc1 (starts at the bias)              80 MFMAs / wave
//   GEGLU in registers: h = u * gelu(gate); the accumulator layout (lane: 4 consecutive columns of row fr) IS a valid second
//         MFMA operand once the K order inside the 32-block is permuted - FF2's weight is packed with that permutation
//         (launch_pack_mlp_w2), so h goes from FF1's accumulators to FF2's operand without leaving the lane;
//   FF2:  one K step over those 32 hidden units against 320 weight rows from LDS -> acc2                       40 MFMAs / wave
// Weights stream through LDS double buffered (W1 chunk 40 KB + W2 chunk 20 KB per stage, LDS-DMA, one barrier per chunk of 120
// MFMAs).  Before: FF1 (row-panel, 75 us) wrote 84 MB of hidden activations that FF2 (tiled kernel, 66 us, K = 1280) read back
// from HBM; 256 workgroups = one per CU for M = 32,768.
// =================================================================================================
struct MlpDev {
    const void* x; int ldx; int M;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* w1; const float* b1;   // [2H][320] (u, gate) interleaved in blocks of 16; bias interleaved alike (never null)
    const void* w2p; const float* b2;  // [320][H], K permuted inside 32-blocks; bias [320] (never null)
    const void* resid; int ldr;
    void* out; int ldo; int H;
    int poison;  // test hook: lds_poison before the first DMA
    // optional continuation (the transformer's proj_out, C = 320 -> 320, no adapter): out2 = (out rows) Wp^T + bp + xres, the FF output never stored
    const void* wp; const float* bp;   // [320][320] K permuted to the accumulator order; bias (never null when wp is set)
    const void* xres; int ldxr;        // outer residual rows (the transformer's input), bf16
    void* out2; int ldo2;
    int rot;     // per-workgroup rotation of the piece order inside a chunk (MRISR_MLP_ROT, default 1)
    int spread;  // next chunk's DMA one piece per K step instead of all at the top of the chunk (MRISR_MLP_SPREAD, default 1)
};

template <int KS, int DBG, bool PROJ = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const MlpDev a) {
    typedef bf16 T;
    constexpr int K = KS * 32, N2 = K, MF = 2, NF1 = 4, NF2 = N2 / 16;
    constexpr int CH1 = 64 * K * 2;            // W1 chunk: 64 interleaved rows x K
    constexpr int CH2 = N2 * 64;               // W2 chunk: N2 rows x 32 hidden (64 B)
    constexpr int W2BASE = 2 * CH1;
    constexpr int CPR = KS * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const int nchunks = a.H / 32;
    if (a.poison) lds_poison(smem, 256);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)min((long long)a.M * a.ldx * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(a.w1, (unsigned)((long long)2 * a.H * K * 2));
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(a.w2p, (unsigned)((long long)N2 * a.H * 2));

    // DMA geometry.  W1 chunk: as in the row-panel kernel (row pitch K*2, 16-byte chunk c ^ (row & 7)).  W2 chunk: rows of 64 B
    // (4 chunks), chunk q ^ g((row & 15) >> 2) with g = {0, 2, 3, 1}: conflict-free ds_read_b128 of 16 rows x one chunk
    // Every workgroup streams the same chunk at about the same time: each walks the pieces of a chunk in its own rotation, so that the
    // CUs of an XCD do not pile up on one L2 channel after the other (a.rot; clock stamps in the sister kernel xtail.hip: the ten DMA
    // instructions of a wave took 2,100 cycles to issue, 1,500 rotated).
    unsigned wvo1[10], wvo2[5];
    int ldo1[10], ldo2[5];
    const int rot_w = a.rot ? (int)(blockIdx.x & 3) : 0, rot_q = a.rot ? (int)((blockIdx.x >> 2) % 10) : 0;
#pragma unroll
    for (int p = 0; p < 10; ++p) {
        int q = p + rot_q;
        if (q >= 10) q -= 10;
        const int piece = q * 4 + ((wave + rot_w) & 3);
        const int L = piece * 64 + lane;
        const int row = L / CPR, cs = L - row * CPR;
        wvo1[p] = (unsigned)((row * K + (cs ^ (row & 7)) * 8) * 2);
        ldo1[p] = __builtin_amdgcn_readfirstlane(piece * 1024);
    }
    auto gsw = [](int x) { return (0x78 >> (2 * x)) & 3; };  // {0, 2, 3, 1}
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        int q = p + (rot_q >> 1);
        if (q >= 5) q -= 5;
        const int piece = q * 4 + ((wave + rot_w) & 3);
        const int L = piece * 64 + lane;
        const int row = L >> 2, qs = L & 3;
        const int qq = qs ^ gsw((row & 15) >> 2);
        wvo2[p] = (unsigned)(((size_t)row * a.H + qq * 8) * 2);
        ldo2[p] = __builtin_amdgcn_readfirstlane(piece * 1024);
    }
    auto stage = [&](int c, int buf, bool live) {
        const unsigned b1o = live ? (unsigned)c * (unsigned)CH1 : 0xC0000000u;
        const unsigned b2o = live ? (unsigned)c * 64u : 0xC0000000u;
#pragma unroll
        for (int p = 0; p < 10; ++p) bl16(r1, smem + buf * CH1 + ldo1[p], wvo1[p], b1o);
#pragma unroll
        for (int p = 0; p < 5; ++p) bl16(r2, smem + W2BASE + buf * CH2 + ldo2[p], wvo2[p], b2o);
    };
    // one piece of chunk c at a time (a.spread): the wave waits in the VMEM queue behind a running MFMA instead of ahead of the K loop
    auto stage1 = [&](int c, int buf, bool live, int p) { bl16(r1, smem + buf * CH1 + ldo1[p], wvo1[p], live ? (unsigned)c * (unsigned)CH1 : 0xC0000000u); };
    auto stage2 = [&](int c, int buf, bool live, int p) { bl16(r2, smem + W2BASE + buf * CH2 + ldo2[p], wvo2[p], live ? (unsigned)c * 64u : 0xC0000000u); };
    constexpr bool proj = PROJ;
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(proj ? a.wp : a.w1, proj ? (unsigned)(N2 * K * 2) : 0u);
    auto stageP = [&](int q, int buf, int p) { bl16(rP, smem + buf * CH1 + ldo1[p], wvo1[p], (unsigned)q * (unsigned)CH1); };

    // ---- the panel rows through LDS (the W1 buffers are still empty), then LayerNorm in registers ----
    bf16x8 af[MF][KS];
    {
        char* areg = smem + wave * (32 * K * 2);
#pragma unroll
        for (int p = 0; p < 32 * CPR / 64; ++p) {
            const int L = p * 64 + lane;
            const int row = L / CPR, cs = L - row * CPR;
            const int m = m0 + wave * 32 + row;
            const unsigned vo = m < a.M ? (unsigned)(((size_t)m * a.ldx + (cs ^ (row & 7)) * 8) * 2) : BL_OOB;
            bl16(rx, areg + p * 1024, vo, 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                af[j][kk] = *reinterpret_cast<const bf16x8*>(areg + (j * 16 + fr) * (K * 2) + (((kk * 4 + fg) ^ (fr & 7)) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    stage(0, 0, true);
    // LayerNorm vectors into LDS (W2 buffer 1, free until the loop's first barrier): see the row-panel kernel
    constexpr int GP = (K * 4 + 1023) / 1024;
    constexpr int GB_OFF = W2BASE + CH2;
    {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.ln_gamma, (unsigned)(K * 4));
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.ln_beta, (unsigned)(K * 4));
        if (wave < GP) bl16(rg, smem + GB_OFF + wave * 1024, (unsigned)(wave * 1024 + lane * 16), 0u);
        else if (wave < 2 * GP) bl16(rb, smem + GB_OFF + wave * 1024, (unsigned)((wave - GP) * 1024 + lane * 16), 0u);
        static_assert(2 * GP <= 4, "one piece per wave");
    }
    {
        float mean[MF], rstd[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const float c0 = __shfl((float)af[j][0][0], fr);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)af[j][kk][e] - c0; s1 += d; s2 = fmaf(d, d, s2); }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float md = s1 * (1.0f / K);
            mean[j] = c0 + md;
            rstd[j] = rsqrtf(fmaxf(s2 * (1.0f / K) - md * md, 0.f) + a.ln_eps);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                rp_u4 w = __builtin_bit_cast(rp_u4, af[j][kk]);
                asm volatile("" : "+v"(w));
                af[j][kk] = __builtin_bit_cast(bf16x8, w);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* gp = reinterpret_cast<const float*>(smem + GB_OFF) + fg * 8;
        const float* bp = reinterpret_cast<const float*>(smem + GB_OFF + GP * 1024) + fg * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp + kk * 32), g1 = *reinterpret_cast<const f32x4*>(gp + kk * 32 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + kk * 32), b1 = *reinterpret_cast<const f32x4*>(bp + kk * 32 + 4);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
                    o[e] = (bf16)(((float)af[j][kk][e] - mean[j]) * rstd[j] * ga + be);
                }
                af[j][kk] = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- FF2 accumulators start at its bias ----
    f32x4 acc2[NF2][MF];
#pragma unroll
    for (int i = 0; i < NF2; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.b2 + i * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < MF; ++j) acc2[i][j] = b;
    }
    const int s2off = fr * 64 + ((fg ^ gsw(fr >> 2)) * 16);
    float pbn[NF1][4];  // FF1 bias of the NEXT chunk (loaded a chunk ahead, behind the previous DMA)
#pragma unroll
    for (int i = 0; i < NF1; ++i) load4<float>(a.b1 + i * 16 + fg * 4, pbn[i]);

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk c (weights + bias) has landed: issued one whole chunk ago
        __syncthreads();
        float pb[NF1][4];
#pragma unroll
        for (int i = 0; i < NF1; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[i][r] = pbn[i][r];
        const bool live = c + 1 < nchunks && !(DBG & 1);
        const bool spread = a.spread != 0;
        auto next_bias = [&]() {
            const int cn = min(c + 1, nchunks - 1);
#pragma unroll
            for (int i = 0; i < NF1; ++i) load4<float>(a.b1 + cn * 64 + i * 16 + fg * 4, pbn[i]);
        };
        if (!spread) {
            next_bias();
            stage(c + 1, buf ^ 1, live);
            if (proj && c + 1 == nchunks) {
#pragma unroll
                for (int p = 0; p < 10; ++p) stageP(0, buf ^ 1, p);
            }
        }
        // ---- FF1 chunk ----
        f32x4 acc1[NF1][MF];
        const char* s1 = smem + buf * CH1 + fr * (K * 2);
        constexpr int PF = 2;  // fragment reads two K steps ahead (one wave per SIMD: nothing else hides the ds_read latency)
        bf16x8 wf[PF + 1][NF1];
        auto load_w = [&](bf16x8 (&dst)[NF1], int kk) {
            const int off = ((kk * 4 + fg) ^ (fr & 7)) * 16;
#pragma unroll
            for (int i = 0; i < NF1; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(s1 + i * 16 * (K * 2) + off);
        };
#pragma unroll
        for (int kk = 0; kk < PF; ++kk) load_w(wf[kk], kk);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (spread) {
                if (proj && c + 1 == nchunks) stageP(0, buf ^ 1, kk); else stage1(c + 1, buf ^ 1, live, kk);
                if (kk == 1) next_bias();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + PF < KS) load_w(wf[(kk + PF) % (PF + 1)], kk + PF);
#pragma unroll
            for (int i = 0; i < NF1; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const f32x4 cz = kk == 0 ? f32x4{pb[i][0], pb[i][1], pb[i][2], pb[i][3]} : acc1[i][j];
                    if (DBG & 4) { acc1[i][j] = cz; acc1[i][j][0] += (float)wf[kk % (PF + 1)][i][0]; continue; }
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk % (PF + 1)][i], af[j][kk], cz, 0, 0, 0);
                }
            // the four fragment reads of step kk + PF between the MFMA pairs of this step: their issue (four waves share the LDS array)
            // hides behind a running MFMA (xtail.hip clock stamps: 196 -> 177 cycles per K step)
            if (!(DBG & 4) && kk + PF < KS) {
#pragma unroll
                for (int q = 0; q < NF1; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GEGLU in registers -> FF2's row operand: k elements 0..3 = hidden 4fg + r of block 0, 4..7 = of block 1 ----
        bf16x8 h[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {  // two at a time: packed f32 math (same operations as gelu_erf_t<bf16>, same bits)
                    const f32x2 u = {acc1[2 * p][j][r], acc1[2 * p][j][r + 1]};
                    const f32x2 gt = {acc1[2 * p + 1][j][r], acc1[2 * p + 1][j][r + 1]};
                    const f32x2 v = (DBG & 2) ? u + gt : u * gelu_erf2_bf16(gt);
                    h[j][p * 4 + r] = (bf16)v[0];
                    h[j][p * 4 + r + 1] = (bf16)v[1];
                }
        // ---- FF2: one K step over these 32 hidden units ----
        const char* s2 = smem + W2BASE + buf * CH2 + s2off;
        bf16x8 w2f[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w2f[0][u] = *reinterpret_cast<const bf16x8*>(s2 + u * 16 * 64);
#pragma unroll
        for (int i0 = 0; i0 < NF2; i0 += 4) {
            const int cur = (i0 >> 2) & 1;
            if (spread) stage2(c + 1, buf ^ 1, live, i0 >> 2);
            __builtin_amdgcn_sched_barrier(0);
            if (i0 + 4 < NF2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) w2f[cur ^ 1][u] = *reinterpret_cast<const bf16x8*>(s2 + (i0 + 4 + u) * 16 * 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    if (DBG & 8) { acc2[i0 + u][j][0] += (float)w2f[cur][u][0] + (float)h[j][0]; continue; }
                    acc2[i0 + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[cur][u], h[j], acc2[i0 + u][j], 0, 0, 0);
                }
            if (!(DBG & 8) && i0 + 4 < NF2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if constexpr (PROJ) {
        // ---- continuation: t2 = bf16(bf16(acc2) + resid) becomes the row operand (accumulator order = the K permutation Wp is packed with) of
        // one more GEMM, 5 chunks of 64 output columns through the W1 ring; its output (+ bias + outer residual) leaves chunk by chunk straight
        // from the accumulator layout.  The FF output itself is never stored: 21 MB less written and read back, one launch less.
        const T* rb = reinterpret_cast<const T*>(a.resid);
        bf16x8 t2[MF][KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = min(m0 + wave * 32 + j * 16 + fr, a.M - 1);
                bf16x4 r0 = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, r1v = r0;
                if (rb) {
                    r0 = *reinterpret_cast<const bf16x4*>(rb + (size_t)m * a.ldr + kk * 32 + fg * 4);
                    r1v = *reinterpret_cast<const bf16x4*>(rb + (size_t)m * a.ldr + kk * 32 + 16 + fg * 4);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bf16 v = (bf16)acc2[2 * kk + (e >> 2)][j][e & 3];
                    t2[j][kk][e] = (bf16)((float)v + (float)(e < 4 ? r0[e & 3] : r1v[e & 3]));
                }
            }
        const T* xr = reinterpret_cast<const T*>(a.xres);
        T* ob2 = reinterpret_cast<T*>(a.out2);
        bf16x4 xv[NF1][MF], xvn[NF1][MF];
        auto load_x = [&](int q, bf16x4 (&dst)[NF1][MF]) {
#pragma unroll
            for (int i = 0; i < NF1; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = min(m0 + wave * 32 + j * 16 + fr, a.M - 1);
                    dst[i][j] = *reinterpret_cast<const bf16x4*>(xr + (size_t)m * a.ldxr + q * 64 + i * 16 + fg * 4);
                }
        };
        float pbp[NF1][4], pbpn[NF1][4];
        auto load_b = [&](int q, float (&dst)[NF1][4]) {
#pragma unroll
            for (int i = 0; i < NF1; ++i) load4<float>(a.bp + q * 64 + i * 16 + fg * 4, dst[i]);
        };
        load_x(0, xvn);
        load_b(0, pbpn);
        constexpr int NQ = N2 / 64;
#pragma nounroll
        for (int q = 0; q < NQ; ++q) {
            const int buf = (nchunks + q) & 1;
            // the eight stores of the previous chunk are the youngest VMEM operations and may stay in flight (vmcnt retires in order)
            if (q == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NF1; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pbp[i][r] = pbpn[i][r];
#pragma unroll
                for (int j = 0; j < MF; ++j) xv[i][j] = xvn[i][j];
            }
            f32x4 acc3[NF1][MF];
            const char* s1 = smem + buf * CH1 + fr * (K * 2);
            constexpr int PF = 2;
            bf16x8 wf[PF + 1][NF1];
            auto load_w = [&](bf16x8 (&dst)[NF1], int kk) {
                const int off = ((kk * 4 + fg) ^ (fr & 7)) * 16;
#pragma unroll
                for (int i = 0; i < NF1; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(s1 + i * 16 * (K * 2) + off);
            };
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) load_w(wf[kk], kk);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (q + 1 < NQ) {
                    stageP(q + 1, buf ^ 1, kk);
                    if (kk == 1) { load_x(q + 1, xvn); load_b(q + 1, pbpn); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + PF < KS) load_w(wf[(kk + PF) % (PF + 1)], kk + PF);
#pragma unroll
                for (int i = 0; i < NF1; ++i)
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        const f32x4 cz = kk == 0 ? f32x4{pbp[i][0], pbp[i][1], pbp[i][2], pbp[i][3]} : acc3[i][j];
                        acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk % (PF + 1)][i], t2[j][kk], cz, 0, 0, 0);
                    }
                if (kk + PF < KS) {
#pragma unroll
                    for (int u = 0; u < NF1; ++u) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NF1; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = m0 + wave * 32 + j * 16 + fr;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bf16 v = (bf16)acc3[i][j][r];
                        o[r] = (bf16)((float)v + (float)xv[i][j][r]);
                    }
                    if (m < a.M) *reinterpret_cast<bf16x4*>(ob2 + (size_t)m * a.ldo2 + q * 64 + i * 16 + fg * 4) = o;
                }
        }
        return;
    }
    // ---- output: acc2 (+ residual) through a wave-private tile, whole rows ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int OP = N2 + 8;
    T* wt = reinterpret_cast<T*>(smem + wave * (32 * OP * 2));
#pragma unroll
    for (int i = 0; i < NF2; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const bf16x4 o = {(bf16)acc2[i][j][0], (bf16)acc2[i][j][1], (bf16)acc2[i][j][2], (bf16)acc2[i][j][3]};
            *reinterpret_cast<bf16x4*>(wt + (j * 16 + fr) * OP + i * 16 + fg * 4) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int CPRO = N2 / 8;
    T* ob = reinterpret_cast<T*>(a.out);
    const T* rb = reinterpret_cast<const T*>(a.resid);
    constexpr int ITO = 32 * CPRO / 64, OB = 5;  // 20 pieces per lane, residual pieces five at a time ahead of their adds (see copy_out_tile)
    static_assert(32 * CPRO % 64 == 0 && ITO % OB == 0, "whole batches");
    for (int it0 = 0; it0 < ITO; it0 += OB) {
        bf16x8 rv[OB];
#pragma unroll
        for (int u = 0; u < OB; ++u) {
            const int idx = (it0 + u) * 64 + lane;
            const int row = idx / CPRO, ch = idx - row * CPRO;
            const int m = m0 + wave * 32 + row;
            if (rb && m < a.M) rv[u] = *reinterpret_cast<const bf16x8*>(rb + (size_t)m * a.ldr + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < OB; ++u) {
            const int idx = (it0 + u) * 64 + lane;
            const int row = idx / CPRO, ch = idx - row * CPRO;
            const int m = m0 + wave * 32 + row;
            if (m >= a.M) continue;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(wt + row * OP + ch * 8);
            if (rb) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rv[u][e]);
            }
            *reinterpret_cast<bf16x8*>(ob + (size_t)m * a.ldo + ch * 8) = v;
        }
    }
}

// FF2 weight [N2][H] (bf16, natural K order) -> K permuted inside every 32-block: position fg*8 + e holds hidden
// (e < 4 ? 4 fg + e : 16 + 4 fg + e - 4), the order in which FF1's accumulator registers line up as an MFMA operand
__global__ void pack_mlp_w2_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int N2, int H) {
    const long long total = (long long)N2 * H;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / H), k = (int)(i - (long long)n * H);
        const int blk = k & ~31, pos = k & 31, fgp = pos >> 3, e = pos & 7;
        const int srck = blk + (e < 4 ? 4 * fgp + e : 16 + 4 * fgp + (e - 4));
        dst[i] = src[(size_t)n * H + srck];
    }
}
int launch_pack_mlp_w2(const void* w2_bf16, void* dst, int N2, int H, hipStream_t st) {
    MRISR_REQUIRE(H % 32 == 0, "fused feed-forward: hidden width a multiple of 32");
    hipLaunchKernelGGL(pack_mlp_w2_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<const bf16*>(w2_bf16), reinterpret_cast<bf16*>(dst), N2, H);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// probe builds (MRISR_MLP_DBG; bits: 1 no weight DMA after chunk 0, 2 no GEGLU math, 4 no FF1 MFMAs, 8 no FF2 MFMAs); 0 = the product
#define MLP_PROBES(X) X(0) X(1) X(2) X(4) X(8) X(12) X(14) X(15)
static int g_mlp_proj = -1;  // test hook: -1 = MRISR_MLP_PROJ (default 1), 0 off, 1 on
extern "C" void mrisr_debug_mlp_proj(int on) { g_mlp_proj = on; }
bool mlp_proj_enabled() {
    static const int env = [] { const char* e = getenv("MRISR_MLP_PROJ"); return e ? atoi(e) : 1; }();
    return g_mlp_proj < 0 ? env != 0 : g_mlp_proj != 0;
}
bool mlp_fused_ok(int C, int H, int N2) {
    static const int env = [] { const char* e = getenv("MRISR_MLP_FUSED"); return e ? atoi(e) : 1; }();
    return env && C == 320 && N2 == 320 && H % 32 == 0 && H >= 64;
}
constexpr int kMlpSmem = 2 * (64 * 320 * 2) + 2 * (320 * 64);
// launch attributes (dynamic LDS above 64 KB); also called from gemm_prepare() so that it never happens inside a stream capture
static int mlp_fused_prepare() {
    static bool attr = false;
    if (!attr) {
#define MLP_ATTR(D) MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<10, D>, hipFuncAttributeMaxDynamicSharedMemorySize, kMlpSmem));
        MLP_PROBES(MLP_ATTR)
#undef MLP_ATTR
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<10, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMlpSmem));
        attr = true;
    }
    return 0;
}
int launch_mlp_fused(const MlpArgs& m, hipStream_t st) {
    MRISR_REQUIRE(mlp_fused_ok(m.C, m.H, m.N2), "fused feed-forward: C = N2 = 320");
    MRISR_REQUIRE(m.x && m.w1 && m.w2p && m.out && m.ln_gamma && m.ln_beta && m.ldx % 8 == 0 && m.ldo % 8 == 0 && (!m.resid || m.ldr % 8 == 0),
                  "fused feed-forward: operands");
    MRISR_REQUIRE((size_t)std::max(2 * m.H, m.N2) * sizeof(float) <= kZeroPageBytes, "fused feed-forward: bias-free form beyond the zero page");
    constexpr int smem = kMlpSmem;
    static const int dbg = [] { const char* e = getenv("MRISR_MLP_DBG"); return e ? atoi(e) : 0; }();
    if (mlp_fused_prepare()) return 1;
    MlpDev d;
    d.x = m.x; d.ldx = m.ldx; d.M = m.M; d.ln_gamma = m.ln_gamma; d.ln_beta = m.ln_beta; d.ln_eps = m.ln_eps;
    d.w1 = m.w1; d.b1 = m.b1 ? m.b1 : static_cast<const float*>(zero_page());
    d.w2p = m.w2p; d.b2 = m.b2 ? m.b2 : static_cast<const float*>(zero_page());
    d.resid = m.resid; d.ldr = m.ldr; d.out = m.out; d.ldo = m.ldo; d.H = m.H;
    d.wp = m.wp; d.bp = m.wp ? (m.bp ? m.bp : static_cast<const float*>(zero_page())) : nullptr;
    d.xres = m.xres; d.ldxr = m.ldxr; d.out2 = m.out2; d.ldo2 = m.ldo2;
    MRISR_REQUIRE(!m.wp || (m.xres && m.out2 && m.ldxr % 4 == 0 && m.ldo2 % 4 == 0 && m.M % 128 == 0), "fused feed-forward + proj_out: operands (whole 128-row panels)");
    d.poison = (gemm_flags_now() & 2048) ? 1 : 0;
    static const int rot = [] { const char* e = getenv("MRISR_MLP_ROT"); return e ? atoi(e) : 1; }();
    static const int spread = [] { const char* e = getenv("MRISR_MLP_SPREAD"); return e ? atoi(e) : 1; }();
    d.rot = rot; d.spread = spread;
    const double fl = 2.0 * m.M * ((double)2 * m.H * m.C + (double)m.H * m.N2);
    const double by = 2.0 * ((double)m.M * m.C * (m.resid ? 3 : 2) + 3.0 * m.H * m.C);
    ProfScope ps("mlp_fused_c320", fl, by, st);
    bool launched = false;
#define MLP_GO(D) if (dbg == D && !(D == 0 && d.wp)) { hipLaunchKernelGGL((mlp_fused_kernel<10, D>), dim3((m.M + 127) / 128), dim3(256), smem, st, d); launched = true; }
    MLP_PROBES(MLP_GO)
#undef MLP_GO
    if (dbg == 0 && d.wp) { hipLaunchKernelGGL((mlp_fused_kernel<10, 0, true>), dim3((m.M + 127) / 128), dim3(256), smem, st, d); launched = true; }
    MRISR_REQUIRE(dbg == 0 || !d.wp, "MRISR_MLP_DBG probes: without the proj_out continuation (MRISR_MLP_PROJ=0)");
    MRISR_REQUIRE(launched, "MRISR_MLP_DBG: no such probe build");
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
    const int nq = g.N >> 2;
    const long long total = (long long)g.M * nq;
    const int z = blockIdx.z;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / nq);
        const int n = (int)(i - (long long)m * nq) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* p = g.partial + (size_t)z * g.splitk * (size_t)g.M * g.N + (size_t)m * g.N + n;
        // the plain row-output form (every split conv / linear of the models): bias, time-embedding row and residual are requested BEFORE the slabs are summed -
        // inside epilogue4 each of them is one more dependent round trip behind the slab loads
        const bool fast = g.out_mode == OUT_ROWS && !g.lora_z && g.heads == 1 && g.batch == 1;
        float b4[4] = {0.f, 0.f, 0.f, 0.f}, t4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
        if (fast) {
            if (g.bias) load4<float>(g.bias + n, b4);
            if (g.rowvec) load4<float>(g.rowvec + (size_t)(m / g.rowvec_div) * g.rowvec_ld + n, t4);
            if (g.resid) load4<T>(reinterpret_cast<const T*>(g.resid) + (size_t)m * g.ldr + n, r4);
        }
        for (int s = 0; s < g.splitk; ++s) {
            float t[4];
            load4<float>(p + (size_t)s * g.M * g.N, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += t[r];
        }
        if (fast) {  // the same operations in the same order as epilogue4
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * g.alpha + b4[r];
            if (g.rowvec) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += t4[r];
            }
            if (g.act == ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (g.act == ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
            }
            if (g.resid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += r4[r];
            }
            store4<T>(reinterpret_cast<T*>(g.out) + (size_t)m * g.ldo + n, v);
            continue;
        }
        epilogue4<T>(g, z, m, n, v, nullptr, nullptr, nullptr, false, /*mfma_lanes=*/false);
    }
}

// ---- host side -----------------------------------------------------------------------------------
// M tiles per group of the tile order (MRISR_GROUP_M overrides; 1 = the old N-first order).  8 measured best over the whole
// step (same box, graph replay: 53.9 slices/s vs 53.0 for N-first; a per-grid sqrt(T * BN / BM) model: 53.1 vs 53.6): profiles/r01e_group_m.log
static int auto_group_m(int ntm, int ntn, int BM, int BN) {
    (void)ntn; (void)BM; (void)BN;
    static const int env = [] { const char* e = getenv("MRISR_GROUP_M"); return e ? atoi(e) : 8; }();
    return std::max(1, std::min(env, ntm));
}

template <typename T, int BM, int BN, int WGM, int WGN>
static int prepare_cfg() {
    constexpr int smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        auto kern = gemm_kernel<T, BM, BN, WGM, WGN>;
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    return 0;
}
template <typename T>
static int prepare_all() {
    if (prepare_cfg<T, 128, 128, 2, 2>()) return 1;
    if (prepare_cfg<T, 256, 64, 4, 1>()) return 1;
    if (prepare_cfg<T, 128, 64, 2, 2>()) return 1;
    if (prepare_cfg<T, 64, 64, 2, 2>()) return 1;
    return 0;
}
template <typename T, int BM, int BN, int WGM, int WGN>
static int launch_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<T, BM, BN, WGM, WGN>;
    if (prepare_cfg<T, BM, BN, WGM, WGN>()) return 1;
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const_cast<GemmArgs&>(g).group_m = auto_group_m(ntm, ntn, BM, BN);
    dim3 grid(ntn * ntm, g.splitk, g.batch);
    static const std::string pname = std::string("gemm_") + (sizeof(T) == 2 ? "bf16_" : "f32_") + std::to_string(BM) + "x" + std::to_string(BN);
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K * g.batch;
        if (by == 0.0) {
            const double a_el = g.conv ? (double)g.B * g.Hin * g.Win * (g.c0 + g.c1) : (double)g.M * g.K;
            by = sizeof(T) * g.batch * (a_el + (double)g.N * g.K + (double)g.M * g.N);
        }
    }
    ProfScope ps(pname.c_str(), fl, by, st);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, g, (const char*)zero_page());
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

static int g_force_tile = 0;  // test hook: 0 auto, 1..4 small-kernel configs, 5..9 ring-kernel configs
extern "C" void mrisr_debug_force_tile(int t) { g_force_tile = t; }

template <int BM, int BN, int WGM, int WGN, int NSTAGE>
static int prepare_bl() {
    constexpr int smem = NSTAGE * (BM + BN) * 128;
    constexpr int smem_l = NSTAGE * ((BM + BN) * 128 + 16 * 128);
    static bool done = false;
    if (!done) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bl_kernel<BM, BN, WGM, WGN, NSTAGE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bl_kernel<BM, BN, WGM, WGN, NSTAGE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_l));
        done = true;
    }
    return 0;
}
template <int BM, int BN, int WGM, int WGN, int NSTAGE>
static int launch_bl(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = NSTAGE * (BM + BN) * 128;
    constexpr int smem_l = NSTAGE * ((BM + BN) * 128 + 16 * 128);
    if (prepare_bl<BM, BN, WGM, WGN, NSTAGE>()) return 1;
    if (NSTAGE > 2)
        MRISR_REQUIRE(!g.conv && !g.c1 && !g.a1 && g.batch == 1 && (g.splitk == 1 || (g.K / 64) / g.splitk >= NSTAGE) && g.M % BM == 0 && g.N % BN == 0,
                      "counted-ring GEMM: plain single-source GEMM that tiles exactly; split: at least NSTAGE K tiles per split");
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const_cast<GemmArgs&>(g).group_m = auto_group_m(ntm, ntn, BM, BN);
    dim3 grid(ntn * ntm, g.splitk, g.batch);
    static const std::string base_name = std::string("gemm_bf16_bl") + std::to_string(BM) + "x" + std::to_string(BN) + (NSTAGE > 2 ? "d" + std::to_string(NSTAGE) : std::string());
    std::string pname = base_name;
    if (prof_enabled() && prof_shapes()) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s %s M=%d N=%d K=%d s=%d b=%d", base_name.c_str(), g.conv ? (g.ups ? "convup" : (g.stride == 2 ? "convs2" : "conv")) : "lin",
                 g.M, g.N, g.K, g.splitk, g.batch);
        pname = buf;
    }
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K * g.batch;
        if (by == 0.0) {
            const double a_el = g.conv ? (double)g.B * g.Hin * g.Win * (g.c0 + g.c1) : (double)g.M * g.K;
            by = 2.0 * g.batch * (a_el + (double)g.N * g.K + (double)g.M * g.N);
        }
    }
    ProfScope ps(prof_intern(pname), fl, by, st);
    if (g.lora_a) {
        MRISR_REQUIRE(g.splitk == 1 && !g.conv && g.c1 == 0 && g.lora_R >= 1 && g.lora_R <= 16 && g.lora_b && g.act != ACT_GEGLU,
                      "in-kernel LoRA down-projection: plain un-split GEMM, R <= 16");
        hipLaunchKernelGGL((gemm_bl_kernel<BM, BN, WGM, WGN, NSTAGE, true>), grid, dim3(256), smem_l, st, g);
    } else {
        hipLaunchKernelGGL((gemm_bl_kernel<BM, BN, WGM, WGN, NSTAGE, false>), grid, dim3(256), smem, st, g);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
// buffer-addressed configurations: id -> <BM, BN, WGM, WGN>
#define BL_CFGS(X)             \
    X(14, 128, 128, 2, 2, 2)   \
    X(15, 256, 64, 4, 1, 2)    \
    X(16, 128, 64, 2, 2, 2)    \
    X(17, 64, 64, 2, 2, 2)     \
    X(18, 64, 128, 2, 2, 2)    \
    X(25, 128, 160, 2, 2, 2)   \
    X(26, 64, 160, 2, 2, 2)    \
    X(27, 160, 160, 2, 2, 2)   \
    X(28, 128, 192, 2, 2, 2)   \
    X(29, 192, 128, 2, 2, 2)   \
    X(30, 160, 128, 2, 2, 2)   \
    X(31, 96, 160, 2, 2, 2)    \
    X(32, 64, 64, 2, 2, 4)     \
    X(33, 64, 64, 2, 2, 3)     \
    X(34, 64, 128, 2, 2, 3)    \
    X(35, 128, 64, 2, 2, 3)    \
    X(36, 128, 128, 2, 2, 3)   \
    X(37, 128, 160, 2, 2, 4)   \
    X(38, 128, 128, 2, 2, 4)   \
    X(39, 64, 160, 2, 2, 3)

static int prepare_bls() {
#define X(id, bm, bn, wm, wn, ns) if (prepare_bl<bm, bn, wm, wn, ns>()) return 1;
    BL_CFGS(X)
#undef X
    return 0;
}
// halo-conv configurations: id -> <BM, BN, WGM, WGN>
#define HALO_CFGS(X)        \
    X(41, 128, 160, 2, 2)   \
    X(42, 128, 128, 2, 2)   \
    X(43, 64, 160, 2, 2)    \
    X(44, 128, 192, 2, 2)   \
    X(45, 256, 64, 4, 1)

static size_t halo_smem(const GemmArgs& g, int BM, int BN) {
    const int TH = BM / g.Win;
    const size_t npix = (size_t)(TH + 2) * (g.Win + 2);
    const size_t patch = (npix * 8 + 255) / 256 * 4096;  // whole 256-lane DMA rounds
    return patch + 2 * (size_t)BN * 128;
}
// eligibility of the halo kernel for a conv launch with M tile BM
static bool halo_ok(const GemmArgs& g, int BM) {
    if (!g.conv || g.stride != 1 || g.ups != 0 || g.zstuff || g.batch != 1 || g.pad != 1) return false;
    if (g.Win % 8 != 0 || g.Win > 64 || BM % g.Win != 0) return false;  // (a 16-pixel fragment may span two image rows: prow is per pixel)
    if ((g.Hin * g.Win) % BM != 0 || g.M % BM != 0 || g.Hout != g.Hin || g.Wout != g.Win) return false;
    const int npix = (BM / g.Win + 2) * (g.Win + 2);
    const int pit_max = BM >= 256 ? 13 : (BM >= 128 ? 9 : 6);
    return (npix * 8 + 255) / 256 <= pit_max && g.N % 4 == 0;
}
template <int BM, int BN, int WGM, int WGN>
static int launch_halo(const GemmArgs& g, hipStream_t st) {
    MRISR_REQUIRE(halo_ok(g, BM), "halo conv kernel: unsupported geometry");
    const size_t smem = halo_smem(g, BM, BN);
    MRISR_REQUIRE(smem <= 160 * 1024, "halo conv kernel: LDS");
    static size_t attr = 0;
    if (smem > attr) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_halo_kernel<BM, BN, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024)));
        attr = 96 * 1024;
    }
    const int ntn = (g.N + BN - 1) / BN, ntm = g.M / BM;
    const_cast<GemmArgs&>(g).group_m = auto_group_m(ntm, ntn, BM, BN);
    dim3 grid(ntn * ntm, g.splitk, 1);
    static const std::string base_name = std::string("gemm_bf16_halo") + std::to_string(BM) + "x" + std::to_string(BN);
    std::string pname = base_name;
    if (prof_enabled() && prof_shapes()) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s conv M=%d N=%d K=%d s=%d b=%d", base_name.c_str(), g.M, g.N, g.K, g.splitk, g.batch);
        pname = buf;
    }
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K;
        if (by == 0.0) by = 2.0 * ((double)g.B * g.Hin * g.Win * (g.c0 + g.c1) + (double)g.N * g.K + (double)g.M * g.N);
    }
    ProfScope ps(prof_intern(pname), fl, by, st);
    hipLaunchKernelGGL((gemm_halo_kernel<BM, BN, WGM, WGN>), grid, dim3(256), smem, st, g);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}
static int halo_bm(int tile) {
    switch (tile) {
#define X(id, bm, bn, wm, wn) case id: return bm;
        HALO_CFGS(X)
#undef X
    }
    return 0;
}

// 256-row halo kernel (eight waves): id -> <BN, weight stages>
#define HALO8_CFGS(X) \
    X(46, 160, 2)     \
    X(47, 160, 3)     \
    X(48, 160, 4)
static bool is_halo8(int tile) { return tile >= 46 && tile <= 48; }
static bool halo8_ok(const GemmArgs& g) {
    if (!g.conv || g.stride != 1 || g.ups != 0 || g.zstuff || g.batch != 1 || g.pad != 1) return false;
    if (g.Win % 8 != 0 || g.Win > 64 || 256 % g.Win != 0) return false;
    if ((g.Hin * g.Win) % 256 != 0 || g.M % 256 != 0 || g.Hout != g.Hin || g.Wout != g.Win) return false;
    const int npix = (256 / g.Win + 2) * (g.Win + 2);
    return (npix * 8 + 511) / 512 <= 7 && g.N % 4 == 0;
}
template <int BN, int NSTAGE>
static int launch_halo8(const GemmArgs& g, hipStream_t st) {
    MRISR_REQUIRE(halo8_ok(g), "256-row halo conv kernel: unsupported geometry");
    const int npix = (256 / g.Win + 2) * (g.Win + 2);
    const size_t smem = (size_t)((npix * 8 + 511) / 512) * 8192 + (size_t)NSTAGE * ((BN * 8 + 511) / 512) * 8192;
    MRISR_REQUIRE(smem <= 160 * 1024, "256-row halo conv kernel: LDS");
    static bool attr = false;
    if (!attr) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_halo8_kernel<BN, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int ntn = (g.N + BN - 1) / BN, ntm = g.M / 256;
    const_cast<GemmArgs&>(g).group_m = auto_group_m(ntm, ntn, 256, BN);
    dim3 grid(ntn * ntm, g.splitk, 1);
    static const std::string base_name = std::string("gemm_bf16_halo256x") + std::to_string(BN) + "r" + std::to_string(NSTAGE);
    std::string pname = base_name;
    if (prof_enabled() && prof_shapes()) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s conv M=%d N=%d K=%d s=%d b=%d", base_name.c_str(), g.M, g.N, g.K, g.splitk, g.batch);
        pname = buf;
    }
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K;
        if (by == 0.0) by = 2.0 * ((double)g.B * g.Hin * g.Win * (g.c0 + g.c1) + (double)g.N * g.K + (double)g.M * g.N);
    }
    ProfScope ps(prof_intern(pname), fl, by, st);
    hipLaunchKernelGGL((gemm_halo8_kernel<BN, NSTAGE>), grid, dim3(512), smem, st, g);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// weight-stationary configurations: id -> <KS, NB>   (K = 32*KS, slab of 16*NB output columns)
#define WS_CFGS(X)   \
    X(50, 10, 10)    \
    X(51, 10, 8)     \
    X(52, 20, 4)

static void ws_dims(int tile, int* ks, int* nb) {
    *ks = *nb = 0;
    switch (tile) {
#define X(id, k, n) case id: *ks = k; *nb = n; break;
        WS_CFGS(X)
#undef X
    }
}
static bool ws_ok(const GemmArgs& g, int tile) {
    int ks, nb;
    ws_dims(tile, &ks, &nb);
    if (!ks || g.conv || g.c1 || g.batch != 1 || g.a1) return false;
    if (g.K != 32 * ks || g.N % (16 * nb) != 0 || g.M % 64 != 0 || g.lda0 % 8 != 0) return false;
    if (g.act == ACT_GEGLU && (nb & 1)) return false;
    if (g.out_mode == OUT_NONE) return false;
    return true;
}
template <int KS, int NB>
static int launch_ws(const GemmArgs& g, hipStream_t st) {
    constexpr int K = KS * 32, BN = NB * 16, WP = K * 2 + 16;
    MRISR_REQUIRE(g.splitk == 1, "weight-stationary kernel: no split-K");
    MRISR_REQUIRE(!g.conv && !g.c1 && !g.a1 && g.batch == 1 && g.K == K && g.N % BN == 0 && g.M % 64 == 0 && g.lda0 % 8 == 0 &&
                      (g.act != ACT_GEGLU || NB % 2 == 0),
                  "weight-stationary kernel: plain rows, K = 32*KS, N a multiple of the slab, M a multiple of 64");
    const bool lora = g.lora_a != nullptr;
    if (lora) MRISR_REQUIRE(g.lora_R >= 1 && g.lora_R <= 16 && g.lora_b && g.act != ACT_GEGLU, "in-kernel LoRA down-projection: R <= 16");
    const size_t smem = (size_t)(BN + (lora ? 16 : 0)) * WP + (lora ? 4 * 16 * 16 * sizeof(float) : 0);
    static bool attr = false;
    if (!attr) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ws_kernel<KS, NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ws_kernel<KS, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    MRISR_REQUIRE(smem <= 150 * 1024, "weight-stationary kernel: LDS");
    const int slabs = g.N / BN, nblk = g.M / 64;
    int rgroups = 256 / slabs;
    if (rgroups < 1) rgroups = 1;
    if (rgroups > nblk) rgroups = nblk;
    static const std::string base_name = std::string("gemm_bf16_ws") + std::to_string(K) + "x" + std::to_string(BN);
    std::string pname = base_name;
    if (prof_enabled() && prof_shapes()) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s lin M=%d N=%d K=%d s=1 b=1", base_name.c_str(), g.M, g.N, g.K);
        pname = buf;
    }
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K;
        if (by == 0.0) by = 2.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
    }
    ProfScope ps(prof_intern(pname), fl, by, st);
    if (lora) hipLaunchKernelGGL((gemm_ws_kernel<KS, NB, true>), dim3(slabs * rgroups), dim3(256), smem, st, g, rgroups);
    else hipLaunchKernelGGL((gemm_ws_kernel<KS, NB, false>), dim3(slabs * rgroups), dim3(256), smem, st, g, rgroups);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// row-panel configurations: id -> <KS, NF, FP8, NW>   (K = 32*KS; weight chunks of 16*NF output columns x K: 40 KB bf16, 20 / 40 KB
// fp8; NW waves = 32*NW-row panels)
#define RP_CFGS(X)           \
    X(60, 10, 4, false, 4)   \
    X(64, 20, 2, false, 2)   \
    X(61, 20, 2, false, 4)   \
    X(62, 10, 4, true, 4)    \
    X(63, 20, 4, true, 2)    \
    X(65, 10, 4, false, 2)

static void rp_dims(int tile, int* ks, int* nf, bool* fp8, int* nw = nullptr) {
    *ks = *nf = 0;
    *fp8 = false;
    switch (tile) {
#define X(id, k, n, f, w) case id: *ks = k; *nf = n; *fp8 = f; if (nw) *nw = w; break;
        RP_CFGS(X)
#undef X
    }
}
static bool rp_ok(const GemmArgs& g, int tile) {
    int ks, nf;
    bool fp8;
    rp_dims(tile, &ks, &nf, &fp8);
    const int bn = nf * 16;
    if (!ks || g.no_rp || g.conv || g.c1 || g.a1 || g.batch != 1 || g.heads != 1 || g.splitk > 1) return false;
    if (g.K != 32 * ks || g.lda0 % 8 != 0 || g.N % 16 != 0 || g.M < 1) return false;
    if (g.alpha != 1.0f || g.rowvec || g.lora_z || (!g.bias && (size_t)g.N * sizeof(float) > kZeroPageBytes)) return false;
    if (fp8 != (g.w8 != nullptr)) return false;  // fp8 operands are a property of the launch (packed weights), never a tuner choice
    if (fp8 && (!g.w_scale || (g.lora_a && (!g.lora_a8 || !g.lora_a_scale)))) return false;
    if (g.act == ACT_GEGLU) {
        if (g.N % 32 != 0 || (nf & 1) || g.out_mode != OUT_ROWS || g.ldo % 8 != 0 || g.resid) return false;
    } else if (g.act != ACT_NONE) {
        return false;
    } else if (g.out_mode == OUT_ROWS) {
        if (g.ldo % 8 != 0 || (g.resid && g.ldr % 8 != 0)) return false;
    } else if (g.out_mode == OUT_HEADS) {
        // head-major sections: every chunk inside one section, everything 8-aligned (16-byte pieces), no residual
        if (g.secC <= 0 || g.secC % bn != 0 || ((g.N | g.M | g.hd | g.secC | g.dpad | g.npad | g.ntok) & 7) != 0 || g.resid) return false;
    } else {
        return false;
    }
    if (g.lora_a && (g.lora_r != 4 || g.lora_R > 16 || !g.lora_b || g.act == ACT_GEGLU)) return false;
    return (long long)g.M * g.lda0 * 2 < 0x7FFFFFFFll && (long long)g.N * g.K * 2 < 0x7FFFFFFFll;
}
// the row-panel configuration for this K when the caller must have one (LayerNorm prologue, fp8 operands; 0: none).  bf16,
// K = 640: the 4-wave form wins on wide outputs (N = 1920: 33 vs 36 us, N = 5120: 62 vs 83 us at M = 8192) although it spills a
// few registers, the 2-wave form on N = 640 (16.3 vs 20.0 us): tools/probes/rp_probe3.sh
int gemm_rp_tile(const GemmArgs& g) {
    static const int env = [] { const char* e = getenv("MRISR_RP"); return e ? atoi(e) : 1; }();
    if (!env) return 0;
    if (g.w8) return g.K == 320 ? (rp_ok(g, 62) ? 62 : 0) : (rp_ok(g, 63) ? 63 : 0);
    if (g.K == 320) return rp_ok(g, 60) ? 60 : 0;
    if (g.K == 640) {
        const int first = g.N > 640 ? 61 : 64, second = g.N > 640 ? 64 : 61;
        return rp_ok(g, first) ? first : (rp_ok(g, second) ? second : 0);
    }
    return 0;
}
template <int KS, int NF, bool FP8, int NW>
static int launch_rp(int tile_id, const GemmArgs& g, hipStream_t st) {
    constexpr int K = KS * 32, BN = NF * 16, BM = 32 * NW;
    constexpr int smem = (FP8 && KS == 10) ? 81920 : 2 * BN * K * (FP8 ? 1 : 2);  // (fp8, K = 320: room for the bf16 row panel on its way in)
    MRISR_REQUIRE(rp_ok(g, tile_id), "row-panel kernel: plain un-split bf16 row GEMM with K = 32*KS (fp8: packed weights + scales)");
    static bool attr = false;
    if (!attr) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rp_kernel<KS, NF, false, 0, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rp_kernel<KS, NF, false, 1, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rp_kernel<KS, NF, true, 0, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rp_kernel<KS, NF, true, 1, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int panels = (g.M + BM - 1) / BM, nchunks = (g.N + BN - 1) / BN;
    static const int ys_env = [] { const char* e = getenv("MRISR_RP_YSPLIT"); return e ? atoi(e) : 0; }();
    int ysplit = ys_env > 0 ? ys_env : (512 + panels / 2) / panels;  // ~2 workgroups per CU
    ysplit = std::max(1, std::min(ysplit, nchunks));
    static const std::string base_name = std::string(FP8 ? "gemm_fp8_rp" : "gemm_bf16_rp") + std::to_string(K) + "x" + std::to_string(BN) + (NW == 2 ? "w2" : "");
    std::string pname = base_name;
    if (prof_enabled() && prof_shapes()) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s lin%s M=%d N=%d K=%d s=1 b=1", base_name.c_str(), g.ln_gamma ? "+ln" : "", g.M, g.N, g.K);
        pname = buf;
    }
    double fl = g.alg_flops, by = g.alg_bytes;
    if (prof_enabled()) {
        if (fl == 0.0) fl = 2.0 * g.M * (double)g.N * g.K;
        if (by == 0.0) by = 2.0 * ((double)g.M * g.K + (double)g.M * g.N) + (FP8 ? 1.0 : 2.0) * (double)g.N * g.K;
    }
    ProfScope ps(prof_intern(pname), fl, by, st);
    const dim3 grid(panels, ysplit);
    static const int pt_rp = [] { const char* e = getenv("MRISR_PRETOUCH_RP"); return e ? atoi(e) : 1; }();
    if (!pt_rp) const_cast<GemmArgs&>(g).pretouch = 0;
    const bool lora = g.lora_a != nullptr;
    if (!g.bias) {  // the kernel loads its chunk's bias unconditionally (straight-line code around the wait counts): zeros
        MRISR_REQUIRE((size_t)g.N * sizeof(float) <= kZeroPageBytes, "row-panel kernel without bias: N beyond the zero page");
        const_cast<GemmArgs&>(g).bias = static_cast<const float*>(zero_page());
    }
    if (g.ln_gamma) {
        MRISR_REQUIRE(g.ln_beta, "LayerNorm prologue: gamma and beta");
        if (lora) hipLaunchKernelGGL((gemm_rp_kernel<KS, NF, true, 1, FP8, NW>), grid, dim3(64 * NW), smem, st, g);
        else hipLaunchKernelGGL((gemm_rp_kernel<KS, NF, false, 1, FP8, NW>), grid, dim3(64 * NW), smem, st, g);
    } else {
        if (lora) hipLaunchKernelGGL((gemm_rp_kernel<KS, NF, true, 0, FP8, NW>), grid, dim3(64 * NW), smem, st, g);
        else hipLaunchKernelGGL((gemm_rp_kernel<KS, NF, false, 0, FP8, NW>), grid, dim3(64 * NW), smem, st, g);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// counted-ring configurations (NSTAGE > 2): plain single-source GEMMs that tile exactly (every DMA instruction fully in range)
static bool ring_ok(const GemmArgs& g, int tile) {
    int bm = 0, bn = 0, ns = 0;
    switch (tile) {
#define X(id, m_, n_, wm, wn, st) case id: bm = m_; bn = n_; ns = st; break;
        BL_CFGS(X)
#undef X
    }
    if (ns <= 2) return true;
    if (g.conv || g.c1 || g.a1 || g.batch != 1) return false;
    return g.M % bm == 0 && g.N % bn == 0 && g.K >= 64 * ns;
}

// buffer descriptors address at most 2^31 bytes per operand
static bool bl_ok(const GemmArgs& g) {
    const long long a_rows = g.conv ? (long long)g.B * g.Hin * g.Win : (long long)g.M;
    const long long lim = 0x7FFFFFFFll;
    return a_rows * g.lda0 * 2 < lim && a_rows * (long long)g.lda1 * 2 < lim && (long long)g.N * g.K * 2 < lim;
}

// ---- tile / split-K planner ------------------------------------------------------------------
struct TileCfg { int id, bm, bn; double eff; bool bl; };
// eff: relative main-loop efficiency measured with tools/gemm_sweep.py (profiles/r01_gemm_sweep.log)
static const TileCfg kTiles[] = {
    {14, 128, 128, 1.00, true}, {15, 256, 64, 0.90, true}, {16, 128, 64, 0.88, true}, {18, 64, 128, 0.86, true},
    {17, 64, 64, 0.60, true},
    {1, 128, 128, 0.75, false}, {2, 256, 64, 0.65, false}, {3, 128, 64, 0.60, false}, {4, 64, 64, 0.45, false},
};
static double g_tile_eff[32] = {0};
extern "C" void mrisr_debug_set_tile_eff(int id, double eff) { if (id >= 0 && id < 32) g_tile_eff[id] = eff; }

// cost model: rounds of (2 workgroups x 256 CUs) x time of one workgroup (MFMA work / efficiency + fixed prologue /
// epilogue latency), plus the split-K slab traffic.  Only the ranking matters.
static void plan(const GemmArgs& g, bool is_bf16, int fixed_split, int* tile_out, int* split_out) {
    const double cu_flops = 1.0e15 / 256.0 * 1e-6;  // FLOP per microsecond per CU at the kernel's practical rate
    double best = 1e300;
    int bt = is_bf16 ? 14 : 1, bs = 1;
    const int nkt = g.K / (is_bf16 ? 64 : 32);
    const bool use_bl = is_bf16 && bl_ok(g);
    for (const TileCfg& t : kTiles) {
        if (t.bl != use_bl) continue;
        const double eff = g_tile_eff[t.id] > 0 ? g_tile_eff[t.id] : t.eff;
        const long long tiles = (long long)((g.M + t.bm - 1) / t.bm) * ((g.N + t.bn - 1) / t.bn) * g.batch;
        for (int s = 1; s <= 32; s *= 2) {
            if (fixed_split > 0 && s != fixed_split) continue;
            if (s > 1 && (g.act == ACT_GEGLU || nkt / s < 6)) break;
            const long long blocks = tiles * s;
            const long long rounds = (blocks + 511) / 512;
            const double wg_us = 2.0 * t.bm * t.bn * ((double)g.K / s) / (cu_flops * eff) * 2.0 + 4.0;
            double cost = rounds * wg_us;
            if (s > 1) cost += 2.0 + (double)g.M * g.N * g.batch * (4.0 * s * 2 + 2) / 3.0e6;  // slab write+read
            if (cost < best) { best = cost; bt = t.id; bs = s; }
        }
    }
    if (fixed_split > 0 && best >= 1e300) bs = fixed_split;
    *tile_out = bt;
    *split_out = bs;
}

// ---- plan-time autotuner ------------------------------------------------------------------------
// The first time a GEMM signature is planned (during the dry pass that sizes a model's workspace - never inside a
// stream capture) every admissible (tile, stages, split-K) candidate is timed on scratch operands of the real
// shape and the winner is cached for the life of the process.  MRISR_AUTOTUNE=0 falls back to the cost model.
__global__ void tune_fill_kernel(bf16* p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (bf16)(((float)(x & 0xFFFF) / 32768.0f - 1.0f) * 0.5f);
    }
}
struct TuneScratch {
    void* p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[6] = {0, 0, 0, 0, 0, 0};
    int reserve(int i, size_t bytes, bool randomize) {
        if (bytes <= cap[i]) return 0;
        if (p[i]) (void)hipFree(p[i]);
        p[i] = nullptr; cap[i] = 0;
        bytes = (bytes + (1 << 20)) & ~(size_t)((1 << 20) - 1);
        MRISR_CHECK_HIP(hipMalloc(&p[i], bytes));
        cap[i] = bytes;
        if (randomize) hipLaunchKernelGGL(tune_fill_kernel, dim3(2048), dim3(256), 0, nullptr, (bf16*)p[i], (long long)(bytes / 2), 17u + i);
        else MRISR_CHECK_HIP(hipMemset(p[i], 0, bytes));
        return 0;
    }
    void release() {
        for (int i = 0; i < 6; ++i) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
    }
};
static TuneScratch g_ts;
static std::map<std::string, std::pair<int, int>> g_tuned;
static int g_tune_count = 0;
static double g_tune_ms = 0.0;
extern "C" int mrisr_autotune_stats(int* shapes, double* ms) { if (shapes) *shapes = g_tune_count; if (ms) *ms = g_tune_ms; return 0; }
extern "C" void mrisr_autotune_release(void) { g_ts.release(); }

static bool autotune_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRISR_AUTOTUNE"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
int mrisr_prof_enable_internal(int on);

template <typename T> int launch_gemm(const GemmArgs& g, hipStream_t st);

static int tune(const GemmArgs& g0, int* tile_out, int* split_out) {
    char key[200];
    snprintf(key, sizeof(key), "%d,%d,%d,c%d,s%d,u%d,%d,%d,a%d,o%d,b%d,r%d,v%d,h%d,w%d", g0.M, g0.N, g0.K, g0.conv, g0.stride + 8 * (1 - g0.pad), g0.ups, g0.c0,
             g0.c1, g0.act, g0.out_mode, g0.batch, g0.resid ? 1 : 0, g0.rowvec ? 1 : 0, g0.Hin, g0.Win);
    if (g0.kw != 3 || g0.subpix) snprintf(key + strlen(key), sizeof(key) - strlen(key), ",q%d", g0.kw * 2 + g0.subpix);  // (older tables have no such keys)
    static bool cache_loaded = false;
    const char* cache_path = getenv("MRISR_TUNE_CACHE");  // optional on-disk table: "key<TAB>tile<TAB>split" per line
    if (!cache_loaded) {
        cache_loaded = true;
        if (cache_path) {
            if (FILE* f = fopen(cache_path, "r")) {
                char k[256];
                int t, sp;
                while (fscanf(f, "%255[^\t]\t%d\t%d\n", k, &t, &sp) == 3) g_tuned[k] = {t, sp};
                fclose(f);
            }
        }
    }
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) { *tile_out = it->second.first; *split_out = it->second.second; return 0; }
    // scratch operands
    GemmArgs g = g0;
    const long long a_rows = g.conv ? (long long)g.B * g.Hin * g.Win : (long long)g.M;
    const size_t zb = (size_t)(g.batch > 1 ? g.batch : 1);
    if (g_ts.reserve(0, (size_t)a_rows * g.lda0 * 2 * zb + 4096, true)) return 1;
    if (g.c1 && g_ts.reserve(1, (size_t)a_rows * g.lda1 * 2 + 4096, true)) return 1;
    if (g_ts.reserve(2, (size_t)g.N * g.K * 2 * zb + 4096, true)) return 1;
    if (g_ts.reserve(3, (size_t)g.M * g.N * 4 * zb + 4096, false)) return 1;
    if (g_ts.reserve(5, (size_t)(g.N + 64) * 4 + ((size_t)(g.rowvec ? g.M / (g.rowvec_div > 0 ? g.rowvec_div : 1) + 1 : 1)) * (g.N + 64) * 4, false)) return 1;
    g.a0 = g_ts.p[0]; g.a_bs = g.batch > 1 ? (long long)a_rows * g.lda0 : 0;
    g.a1 = g.c1 ? g_ts.p[1] : nullptr;
    g.w = g_ts.p[2]; g.w_bs = g.batch > 1 ? (long long)g.N * g.K : 0;
    g.bias = g.bias ? (const float*)g_ts.p[5] : nullptr;
    if (g.rowvec) { g.rowvec = (const float*)g_ts.p[5] + g.N + 64; g.rowvec_ld = g.N + 64; }
    g.heads = 1; g.o_hs = 0; g.o_bs = g.batch > 1 ? (long long)g.M * g.N : 0;
    if (g.out_mode == OUT_HEADS) g.out_mode = OUT_ROWS;
    g.out = g_ts.p[3]; g.ldo = g.act == ACT_GEGLU ? g.N / 2 : g.N;
    if (g.resid) { g.resid = g_ts.p[3]; g.ldr = g.ldo; }
    g.lora_z = nullptr;  // the rank-r epilogue term is negligible for ranking the candidates
    g.lora_a = nullptr;
    g.lora_zout = nullptr;
    const int nkt = g.K / 64;
    const long long t128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
    double best = 1e30;
    int bt = 14, bs = 1;
    hipEvent_t e0, e1;
    MRISR_CHECK_HIP(hipEventCreate(&e0));
    MRISR_CHECK_HIP(hipEventCreate(&e1));
    const bool prof_was = prof_enabled();
    mrisr_prof_enable_internal(0);
    hipEvent_t t0, t1;
    MRISR_CHECK_HIP(hipEventCreate(&t0));
    MRISR_CHECK_HIP(hipEventCreate(&t1));
    MRISR_CHECK_HIP(hipEventRecord(t0, nullptr));
    // 50-52 (weight-stationary short-K kernels) are NOT candidates: correct, but 30-60 % slower than the tiled kernels on
    // every shape they fit (profiles/r01b_ws_sweep.log: one A fragment per wave makes them LDS-read bound); kept for the record
    // (32-36, the counted-ring variants, are NOT candidates: on the shapes they were built for - level-2 linears, M = 2,048 - they
    // tie or lose: 64 x 64: 16.7 us with 2 stages, 17.1 with 3, 24.4 with 4 (LDS then admits 2 workgroups per CU instead of 5);
    // these GEMMs move 210 MB through LDS in 16.7 us = 12.6 TB/s, i.e. they sit at the L2 -> LDS bandwidth, not on a latency chain:
    // profiles/r02i_ring_probe.log.  Kept, tested, reachable through mrisr_debug_force_tile.)
    static const int cand[] = {14, 15, 16, 17, 18, 25, 26, 28, 41, 42, 43, 44, 45, 60, 61, 64, 65};  // (62, 63: fp8 operands, chosen by the launch, not the tuner)  // 27, 29-31 never won a shape (profiles/r01_gemm_sweep_tiles.log)
    for (int tile : cand) {
        const bool deep = false;
        {   // debugging aid: MRISR_TUNE_SKIP="41,43" removes candidates
            static const std::string skip = [] { const char* e = getenv("MRISR_TUNE_SKIP"); return std::string(e ? e : ""); }();
            if (!skip.empty() && ("," + skip + ",").find("," + std::to_string(tile) + ",") != std::string::npos) continue;
        }
        if (tile >= 32 && tile <= 39 && !ring_ok(g, tile)) continue;  // counted-ring variants: exact plain GEMMs only
        if (is_halo8(tile)) { if (!halo8_ok(g)) continue; }
        else if (tile >= 40 && tile < 50 && !halo_ok(g, halo_bm(tile))) continue;  // LDS-halo conv kernels: stride-1 3x3, whole tiles per image
        if (tile >= 50 && tile < 60 && !ws_ok(g, tile)) continue;  // weight-stationary kernels: short-K plain GEMMs that tile exactly
        if (tile >= 60 && !rp_ok(g, tile)) continue;  // row-panel kernels: K = 320 / 640 row GEMMs
        // 5 fragments per wave along N (BN = 160): no (u, gate) pairing for the GEGLU epilogue
        if ((tile == 25 || tile == 26 || tile == 27 || tile == 31) && g.act == ACT_GEGLU) continue;
        if (deep && t128 >= 2048) continue;  // plenty of workgroups per CU: the 2-stage structure wins (sweep)
        for (int s = 1; s <= 32; s *= 2) {
            if (s > 1 && (g.act == ACT_GEGLU || nkt / s < 4 || t128 * s > 4096 || tile >= 50 || (tile >= 32 && tile <= 39) || g.kw != 3 || g.subpix)) break;
            const size_t pbytes = s > 1 ? (size_t)s * g.M * g.N * 4 * zb : 0;
            if (pbytes > ((size_t)1 << 30)) break;
            if (s > 1 && g_ts.reserve(4, pbytes, false)) return 1;
            g.splitk = s;
            g.partial = s > 1 ? (float*)g_ts.p[4] : nullptr;
            g.tile = tile;
            if (launch_gemm<bf16>(g, nullptr)) return 1;  // warm-up
            MRISR_CHECK_HIP(hipEventRecord(e0, nullptr));
            static const int iters = [] { const char* e = getenv("MRISR_TUNE_ITERS"); const int v = e ? atoi(e) : 3; return v > 0 ? v : 3; }();
            for (int i = 0; i < iters; ++i) (void)launch_gemm<bf16>(g, nullptr);
            MRISR_CHECK_HIP(hipEventRecord(e1, nullptr));
            MRISR_CHECK_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            MRISR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (ms / iters < best) { best = ms / iters; bt = tile; bs = s; }
        }
    }
    MRISR_CHECK_HIP(hipEventRecord(t1, nullptr));
    MRISR_CHECK_HIP(hipEventSynchronize(t1));
    float tms = 0.f;
    (void)hipEventElapsedTime(&tms, t0, t1);
    g_tune_ms += tms;
    ++g_tune_count;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    mrisr_prof_enable_internal(prof_was ? 1 : 0);
    g_tuned[key] = {bt, bs};
    // the on-disk table is READ-ONLY unless MRISR_TUNE_WRITE=1 (building a table to ship): a bench or test run must not modify
    // a tracked file, and the ranks of a multi-process launch must not append to one file concurrently.  One line per
    // signature, written with a single O_APPEND write.
    static const bool cache_write = [] { const char* e = getenv("MRISR_TUNE_WRITE"); return e && e[0] == '1'; }();
    if (cache_path && cache_write) {
        if (FILE* f = fopen(cache_path, "a")) {
            char line[256];
            const int len = snprintf(line, sizeof(line), "%s\t%d\t%d\n", key, bt, bs);
            if (len > 0) (void)fwrite(line, 1, (size_t)len, f);
            fclose(f);
        }
    }
    *tile_out = bt;
    *split_out = bs;
    return 0;
}

// Chooses tile + split-K for g (sets g.tile / g.splitk).  Deterministic per signature within a process.
static int g_force_split = 0;  // test hook: split-K factor for every GEMM that can be split (exercises the reduce kernel's epilogue)
extern "C" void mrisr_debug_force_split(int s) { g_force_split = s; }
// test hook / MRISR_GEMM_FLAGS: 8 = LoRA up-projection in the scalar epilogue instead of the matrix cores, 16 = head-major outputs
// stored straight from the accumulator layout instead of through LDS (the older code paths, kept as cross-checks)
static int g_gemm_flags = [] { const char* e = getenv("MRISR_GEMM_FLAGS"); return e ? atoi(e) : 0; }();
extern "C" void mrisr_debug_gemm_flags(int f) { g_gemm_flags = f; }
static int gemm_flags_now() { return g_gemm_flags; }
int xattn_tail_flags() { return g_gemm_flags; }
// tools/table_search.py: overrides one entry of the tile table in this process (key as in the table file)
extern "C" void mrisr_debug_set_tuned(const char* key, int tile, int split) { g_tuned[key] = {tile, split}; }
int g_subpix_override = -1;
extern "C" void mrisr_debug_subpix(int min_rows) { g_subpix_override = min_rows; }
static int g_prefer_tile = 0;  // test hook: use this specialised kernel (halo 41-45 / weight-stationary 50-52) wherever it is eligible
extern "C" void mrisr_debug_prefer_tile(int t) { g_prefer_tile = t; }

int gemm_choose(GemmArgs& g, bool is_bf16) {
    int t = 0, s = 1;
    if (g.ln_gamma || g.w8) {  // a fused LayerNorm prologue / fp8 operands exist only in the row-panel kernel (the caller checked gemm_rp_tile)
        g.tile = gemm_rp_tile(g);
        g.splitk = 1;
        MRISR_REQUIRE(g.tile != 0, "LayerNorm prologue / fp8 operands need the row-panel kernel");
        return 0;
    }
    if (g_force_tile) { g.tile = g_force_tile; if (g.splitk < 1) g.splitk = 1; return 0; }
    if (g_prefer_tile && is_bf16 && bl_ok(g) &&
        ((g_prefer_tile >= 60 && rp_ok(g, g_prefer_tile)) || (g_prefer_tile >= 50 && g_prefer_tile < 60 && ws_ok(g, g_prefer_tile)) || (g_prefer_tile >= 40 && g_prefer_tile < 50 && (is_halo8(g_prefer_tile) ? halo8_ok(g) : halo_ok(g, halo_bm(g_prefer_tile)))))) {
        g.tile = g_prefer_tile;
        g.splitk = 1;
        return 0;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (is_bf16 && bl_ok(g) && autotune_enabled()) {
        if (tune(g, &t, &s)) return 1;
    } else {
        plan(g, is_bf16, 0, &t, &s);
    }
    (void)cs;
    if ((t >= 60 && !rp_ok(g, t)) || (t >= 32 && t <= 39 && (!ring_ok(g, t) || s > 1)) || (is_halo8(t) && !halo8_ok(g))) plan(g, is_bf16, 0, &t, &s);  // (a table entry tuned without this launch's operand forms)
    if (g_force_split > 1 && g.act != ACT_GEGLU && g.K / (is_bf16 ? 64 : 32) >= g_force_split && !g.lora_a) {
        s = g_force_split;
        if (t >= 50) t = is_bf16 ? 14 : 1;
    }
    g.tile = t;
    g.splitk = s;
    return 0;
}

int gemm_prepare() {
    if (init_zero_page()) return 1;
    if (prepare_all<float>()) return 1;
    if (prepare_all<bf16>()) return 1;
    if (direct_conv_prepare()) return 1;
    if (xattn_tail_prepare()) return 1;
    if (mlp_fused_prepare()) return 1;
    return prepare_bls();
}

// per-stream arrival counters for the in-kernel split-K reduction: launches on one stream are ordered and every launch leaves its
// counters at zero, so one zeroed array per stream serves them all.  Not created during stream capture (the eager warm-up on the same
// stream has created it before; otherwise the launch falls back to the separate reduce kernel).
// OFF by default (MRISR_SK_INKERNEL=1 turns it on): measured on the bench workload it is no faster than the reduce launches it removes -
// 10.92 vs 10.82 ms per denoising step in the replayed graph (41 fewer launches, but every workgroup's agent-scope release is an L2
// write-back and the last arriver's tail is latency-bound); launched eagerly the split GEMMs take 3-4x longer (M=512 s=8 conv: 33 ->
// 138 us), profiles/r02m_splitk_inkernel.log.  Kept as a tested alternative (tests/test_gpu_ops.py::test_splitk_reduced_inside_the_gemm_kernel).
static bool tile_reduces_in_kernel(int tile) {
#define X(id, bm, bn, wm, wn, ...) if (tile == id) return ((bm) / (wm) / 16) * ((bn) / (wn) / 16) <= 20;  // as the kernels' `if constexpr`
    BL_CFGS(X)
    HALO_CFGS(X)
#undef X
    return false;
}
static long long tile_count(int, const GemmArgs& g) { return (long long)((g.M + 63) / 64) * ((g.N + 63) / 64); }  // upper bound: no tile is smaller than 64 x 64
static constexpr int kSkCounters = 16384;
static int g_sk_inkernel = -1;  // test hook: 0 = the separate reduce kernel, 1 = in-kernel, -1 = MRISR_SK_INKERNEL (default 0)
extern "C" void mrisr_debug_sk_inkernel(int on) { g_sk_inkernel = on; }
static std::mutex g_sk_mu;
static std::unordered_map<hipStream_t, unsigned*> g_sk_map;
static unsigned* sk_counters_for(hipStream_t st) {
    static const int env = [] { const char* e = getenv("MRISR_SK_INKERNEL"); return e ? atoi(e) : 0; }();
    if (g_sk_inkernel < 0 ? !env : !g_sk_inkernel) return nullptr;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    auto it = g_sk_map.find(st);
    if (it != g_sk_map.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) return nullptr;
    unsigned* p = nullptr;
    if (hipMalloc(&p, kSkCounters * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kSkCounters * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_sk_map[st] = p;
    return p;
}

template <typename T>
int launch_gemm(const GemmArgs& g, hipStream_t st) {
    constexpr int BK = 128 / (int)sizeof(T);
    MRISR_REQUIRE(g.K % BK == 0, "K must be a multiple of the 128-byte K tile");
    MRISR_REQUIRE(g.N % 4 == 0, "N must be a multiple of 4");
    MRISR_REQUIRE(g.c0 % BK == 0 && g.c1 % BK == 0, "channel extents must be multiples of the K tile");
    MRISR_REQUIRE(g.splitk == 1 || g.partial != nullptr, "split-K needs a partial buffer");
    MRISR_REQUIRE(g.splitk == 1 || g.act != ACT_GEGLU, "GEGLU epilogue cannot be split");
    MRISR_REQUIRE(zero_page() != nullptr, "zero page not initialised");
    if (g.conv) { MRISR_REQUIRE(g.K == g.kw * g.kw * (g.c0 + g.c1) && (g.kw == 3 || g.kw == 2), "conv K"); }
    else { MRISR_REQUIRE(g.K == g.c0 + g.c1, "plain K"); }
    if (g.kw != 3 || g.subpix) {
        MRISR_REQUIRE(sizeof(T) == 2 && g.conv && !g.c1 && g.stride == 1 && !g.ups && !g.zstuff && g.splitk == 1 && (!g.subpix || g.batch == 4),
                      "2 x 2 / sub-pixel conv: bf16 buffer-addressed kernels, single source, stride 1, un-split, four parity batches");
    }
    static const int stage_env = [] { const char* e = getenv("MRISR_STAGE_OUT"); return e ? atoi(e) : 2; }();  // 0 off, 1 tiled kernels, 2 + halo kernels (default), 3: as 2 but residual added before staging
    const_cast<GemmArgs&>(g).stage_out = stage_env;
    const_cast<GemmArgs&>(g).dbg = g_gemm_flags;
    int tile = g.tile ? g.tile : g_force_tile, s = g.splitk;
    if (!tile) plan(g, sizeof(T) == 2, g.splitk, &tile, &s);
    if ((sizeof(T) != 2 || !bl_ok(g)) && tile > 4) tile = 1;
    if ((g.kw != 3 || g.subpix) && !(tile >= 14 && tile <= 31)) tile = 25;  // only gemm_bl_kernel's gather knows these forms
    // split-K: the bf16 DMA kernels (tiled and halo) finish the reduction themselves
    {   // cold-weight pre-touch: matrices of at least MRISR_PRETOUCH_MIN_KB (default 512; 0 ... 1024 measure the same); MRISR_PRETOUCH=0 turns it off
        static const int on = [] { const char* e = getenv("MRISR_PRETOUCH"); return e ? atoi(e) : 1; }();
        static const long long min_b = [] { const char* e = getenv("MRISR_PRETOUCH_MIN_KB"); return (e ? atoll(e) : 512ll) * 1024; }();
        static const int cap = [] { const char* e = getenv("MRISR_PRETOUCH_CAP"); return e ? atoi(e) : 32; }();  // 1-KiB pieces per wave at most
        const_cast<GemmArgs&>(g).pretouch = (on && sizeof(T) == 2 && g.batch == 1 && (long long)g.N * g.K * 2 >= min_b) ? cap : 0;
    }
    const_cast<GemmArgs&>(g).sk_counters = nullptr;
    if (g.splitk > 1 && sizeof(T) == 2 && !g.defer_reduce && tile_reduces_in_kernel(tile)) {
        const long long ntiles = tile_count(tile, g) * (long long)g.batch;
        if (ntiles > 0 && ntiles <= kSkCounters) const_cast<GemmArgs&>(g).sk_counters = sk_counters_for(st);
    }
    int rc;
    switch (tile) {
        case 2: rc = launch_cfg<T, 256, 64, 4, 1>(g, st); break;
        case 3: rc = launch_cfg<T, 128, 64, 2, 2>(g, st); break;
        case 4: rc = launch_cfg<T, 64, 64, 2, 2>(g, st); break;
#define X(id, bm, bn, wm, wn, ns) case id: rc = launch_bl<bm, bn, wm, wn, ns>(g, st); break;
        BL_CFGS(X)
#undef X
#define X(id, bm, bn, wm, wn) case id: rc = launch_halo<bm, bn, wm, wn>(g, st); break;
        HALO_CFGS(X)
#undef X
#define X(id, bn, ns) case id: rc = launch_halo8<bn, ns>(g, st); break;
        HALO8_CFGS(X)
#undef X
#define X(id, ks, nb) case id: rc = launch_ws<ks, nb>(g, st); break;
        WS_CFGS(X)
#undef X
#define X(id, ks, nf, f8, nw) case id: rc = launch_rp<ks, nf, f8, nw>(id, g, st); break;
        RP_CFGS(X)
#undef X
        default: rc = launch_cfg<T, 128, 128, 2, 2>(g, st); break;
    }
    if (rc) return rc;
    if (g.splitk > 1 && !g.sk_counters && !g.defer_reduce) return launch_splitk_reduce<T>(g, st);
    return 0;
}

template <typename T>
int launch_splitk_reduce(const GemmArgs& g, hipStream_t st) {
    const long long total = (long long)g.M * (g.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("splitk_reduce", 0.0, (double)g.batch * g.M * g.N * (4.0 * g.splitk + sizeof(T)), st);
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks, 1, g.batch), dim3(256), 0, st, g);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

template int launch_gemm<float>(const GemmArgs&, hipStream_t);
template int launch_gemm<bf16>(const GemmArgs&, hipStream_t);
template int launch_splitk_reduce<float>(const GemmArgs&, hipStream_t);
template int launch_splitk_reduce<bf16>(const GemmArgs&, hipStream_t);

}  // namespace mrisr
