// Flash-style attention on the bf16 matrix cores, computed "transposed" so that every softmax statistic is
// lane-local:
//     S^T[key][q] = K . Q^T            (MFMA: first operand = K rows, second = Q rows; both d-contiguous)
//     O^T[d][q]  += V^T[d][key] . P^T  (first operand = V^T rows, key-contiguous; second = P^T from registers)
// A lane owns one query column q = lane & 15 of every tile: its S^T accumulators are 4 consecutive keys, which
// is exactly the k-slice the second MFMA wants from it (the key permutation inside a 32-key step is applied
// to the V^T fragment addresses instead), so P never leaves registers and the running max / rescale factor
// are per-lane scalars.  Operands arrive head-major ([b*h][token][dpad], V already transposed
// [b*h][dpad][token]) straight from the QKV GEMM epilogue; padding rows/columns are zero.
#include "common.h"
#include "prof.h"

namespace mrisr {

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

typedef __attribute__((ext_vector_type(4))) short short4v;

template <int DPAD, int DB, int QF>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int KT = 64;                  // keys per LDS tile
    constexpr int KSTEPS = DPAD / 32;       // MFMA k-steps over the head dim for S
    // DB: 16-row blocks of O^T actually needed = ceil(hd / 16)  (hd = 40 -> 3 of the 4 in DPAD = 64)
    constexpr int KP = DPAD * 2 + 16;       // K tile row pitch (bytes), padded
    constexpr int VP = KT * 2 + 16;         // V^T tile row pitch (bytes), padded
    constexpr int CPT = DPAD / 32;          // 16-byte chunks per thread per tile (K and V^T alike)
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds[DPAD * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QF);
    const bf16* qb = reinterpret_cast<const bf16*>(a.q) + (size_t)bh * ((a.nq + 63) / 64 * 64) * DPAD;
    const bf16* kb = reinterpret_cast<const bf16*>(a.k) + (size_t)bh * a.nkpad * DPAD;
    const bf16* vb = reinterpret_cast<const bf16*>(a.vt) + (size_t)bh * DPAD * a.nkpad;
    const float sl2 = a.scale * 1.4426950408889634f;

    // Q fragments (second MFMA operand: rows = q, d-contiguous), straight from global
    bf16x8 qf[QF][KSTEPS];
    const int q_last = (a.nq + 63) / 64 * 64 - 1;  // last row of this head's (padded) q buffer
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
            qf[f][kk] = *reinterpret_cast<const bf16x8*>(qb + (size_t)min(q0 + f * 16 + fr, q_last) * DPAD + kk * 32 + fg * 8);

    f32x4 oacc[QF][DB];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = -INFINITY;
        l_run[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d) oacc[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // tile staging: global -> registers (issued a tile ahead) -> LDS
    bf16x8 kreg[CPT], vreg[CPT];
    auto fetch = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            {   // K tile: KT rows x DPAD/8 chunks
                const int row = c / (DPAD / 8), ch = c % (DPAD / 8);
                kreg[i] = *reinterpret_cast<const bf16x8*>(kb + (size_t)(kt0 + row) * DPAD + ch * 8);
            }
            {   // V^T tile: DPAD rows x 8 chunks
                const int row = c >> 3, ch = c & 7;
                vreg[i] = *reinterpret_cast<const bf16x8*>(vb + (size_t)row * a.nkpad + kt0 + ch * 8);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            {
                const int row = c / (DPAD / 8), ch = c % (DPAD / 8);
                *reinterpret_cast<bf16x8*>(k_lds + row * KP + ch * 16) = kreg[i];
            }
            {
                const int row = c >> 3, ch = c & 7;
                *reinterpret_cast<bf16x8*>(v_lds + row * VP + ch * 16) = vreg[i];
            }
        }
    };

    const int ntiles = (a.nk + KT - 1) / KT;
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int kt0 = t * KT;
        __syncthreads();  // previous tile fully consumed
        commit();
        __syncthreads();
        if (t + 1 < ntiles) fetch(kt0 + KT);
        // ---- S^T for 64 keys x (QF*16) queries: 4 key blocks of 16 ----
        f32x4 s[QF][4];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) s[f][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + (tt * 16 + fr) * KP + (kk * 32 + fg * 8) * 2);
#pragma unroll
                for (int f = 0; f < QF; ++f) s[f][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[f][kk], s[f][tt], 0, 0, 0);
            }
        }
        const bool ragged = kt0 + KT > a.nk;  // only the last tile can hold masked keys (uniform branch)
        // ---- online softmax, once per 64 keys: lane owns q = fr; its 16 values are keys 16tt + 4fg + r ----
        bf16x8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            if (ragged) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt0 + tt * 16 + fg * 4 + r >= a.nk) s[f][tt][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3]));
#pragma unroll
            for (int tt = 1; tt < 4; ++tt) mx = fmaxf(mx, fmaxf(fmaxf(s[f][tt][0], s[f][tt][1]), fmaxf(s[f][tt][2], s[f][tt][3])));
            mx = xor_max(mx) * sl2;                       // running max kept in the scaled (log2) domain
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
            m_run[f] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[f][tt][r], sl2, -m_new));  // exp(scale*(s - max))
                    ps += p;
                    pf[f][tt >> 1][(tt & 1) * 4 + r] = (bf16)p;
                }
            l_run[f] = l_run[f] * alpha + ps;
#pragma unroll
            for (int d = 0; d < DB; ++d) oacc[f][d] *= alpha;
        }
        // ---- O^T += V^T . P^T : two 32-key steps ----
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const char* vrow = v_lds + (d * 16 + fr) * VP + (sub * 32 + fg * 4) * 2;
                const short4v lo = *reinterpret_cast<const short4v*>(vrow);
                const short4v hi = *reinterpret_cast<const short4v*>(vrow + 32);
                typedef __attribute__((ext_vector_type(8))) short short8v;
                const short8v packed = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bf16x8 vf = __builtin_bit_cast(bf16x8, packed);
#pragma unroll
                for (int f = 0; f < QF; ++f) oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[f][sub], oacc[f][d], 0, 0, 0);
            }
        }
    }

    // ---- normalise and store: lane holds O^T[d = 16db + 4fg + r][q = fr] ----
    const int b = bh / a.H, h = bh - b * a.H;
    bf16* ob = reinterpret_cast<bf16*>(a.out);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const float inv = 1.0f / xor_sum(l_run[f]);
        const int q = q0 + f * 16 + fr;
        if (q < a.nq) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int dd = d * 16 + fg * 4;
                if (dd < a.hd) {
                    bf16x4 o = {(bf16)(oacc[f][d][0] * inv), (bf16)(oacc[f][d][1] * inv), (bf16)(oacc[f][d][2] * inv),
                                (bf16)(oacc[f][d][3] * inv)};
                    *reinterpret_cast<bf16x4*>(ob + ((size_t)b * a.nq + q) * (a.H * a.hd) + h * a.hd + dd) = o;
                }
            }
        }
    }
}

template <int DPAD, int DB>
static int launch_dpad(const AttnArgs& a, hipStream_t st) {
    const int BH = a.B * a.H;
    ProfScope ps("flash_attention", 4.0 * BH * (double)a.nq * a.nk * a.hd,
                 2.0 * BH * (2.0 * a.nq * a.hd + 2.0 * a.nk * a.hd), st);
    if (a.nq >= 128) {
        hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 2>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 1>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention_bf16(const AttnArgs& a, hipStream_t st) {
    MRISR_REQUIRE(a.nkpad % 64 == 0 && a.nk >= 1 && a.nk <= a.nkpad, "attention: key padding");
    MRISR_REQUIRE(a.hd % 4 == 0 && a.hd <= a.dpad, "attention: head dim");
    const int db = (a.hd + 15) / 16;
#define ATT_CASE(DP, DBV) if (a.dpad == DP && db == DBV) return launch_dpad<DP, DBV>(a, st);
    ATT_CASE(32, 1) ATT_CASE(32, 2)
    ATT_CASE(64, 3) ATT_CASE(64, 4)
    ATT_CASE(96, 5) ATT_CASE(96, 6)
    ATT_CASE(128, 7) ATT_CASE(128, 8)
    ATT_CASE(160, 9) ATT_CASE(160, 10)
#undef ATT_CASE
    MRISR_REQUIRE(false, "attention: unsupported (padded head dim, head dim) combination");
    return 0;
}

}  // namespace mrisr
