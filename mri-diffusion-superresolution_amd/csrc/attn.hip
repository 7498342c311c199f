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
#include <type_traits>

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
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned uint4v;

// (cross-row-group reductions stay on ds_bpermute: the gfx950 v_permlane16/32_swap builtins were tried - both results of the
// builtin come back as {row 0, row 0, row 2, row 2}, i.e. only the odd rows receive their partner, tools/probes/permlane_test.hip -
// so one swap does not give the symmetric exchange a reduction needs)
// The key loop of this kernel is VALU-issue bound (one exp per 160 FLOP at hd = 40), so everything that is not an exp, a
// max or a pack has been moved onto the matrix cores or out of the loop:
//   * Q is pre-scaled by scale * log2(e) once per wave (f32, rounded back to bf16), so S comes out in log2 units;
//   * the running reference -m is the C INPUT of the first S MFMA: the product leaves the matrix core as s - m, ready for
//     v_exp_f32 - no per-score FMA;
//   * the row sum l is one more row of the P.V product: ONES = the 16-row block holding row `hd` of V^T has spare rows (hd = 40:
//     rows 40..47 of the third block), the tile commit writes 1.0 into row hd, and O^T[hd] accumulates sum_k p - no per-score add;
//   * the reference m moves only when a tile's maximum exceeds it by more than THR (2^8) - or on the first tile - so the
//     accumulator rescale is a rarely taken wave-uniform branch (exact: everything at the old reference is rescaled once,
//     nothing else; p <= 2^8 keeps bf16 P relative precision and f32 sums far from overflow);
template <int DPAD, int DB, int QF, bool ONES>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int KT = 64;                  // keys per LDS tile
    constexpr int KSTEPS = DPAD / 32;       // MFMA k-steps over the head dim for S
    // DB: 16-row blocks of O^T actually needed = ceil(hd / 16)  (hd = 40 -> 3 of the 4 in DPAD = 64)
    constexpr int KP = DPAD * 2 + 16;       // K tile row pitch (bytes), padded
    constexpr int VP = KT * 2 + 16;         // V^T tile row pitch (bytes), padded
    constexpr int CPT = DPAD / 32;          // 16-byte chunks per thread per tile (K and V^T alike)
    constexpr float THR = 8.0f;
    // two tile buffers: tile t + 1 is committed while tile t is consumed - ONE barrier per tile
    __shared__ __attribute__((aligned(16))) char k_lds2[2][KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds2[2][DPAD * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs by their linear id, so the default (query block, head)
    // grid spreads the query blocks of ONE head over all 8 private L2s and each of them fetches that head's K / V.  With the
    // remap every query block of a head lands on the same XCD: its K / V tiles are fetched once and hit in L2 afterwards.
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (a.xcd_map) {
        const int L = blockIdx.x + blockIdx.y * gridDim.x;
        const int slot = L >> 3;
        bh = (slot / (int)gridDim.x) * 8 + (L & 7);
        qblk = slot % (int)gridDim.x;
    }
    const int q0 = (qblk * 4 + wave) * (16 * QF);
    const bf16* qb = reinterpret_cast<const bf16*>(a.q) + (size_t)bh * ((a.nq + 63) / 64 * 64) * DPAD;
    const bf16* kb = reinterpret_cast<const bf16*>(a.k) + (size_t)bh * a.nkpad * DPAD;
    const bf16* vb = reinterpret_cast<const bf16*>(a.vt) + (size_t)bh * DPAD * a.nkpad;
    const float sl2 = a.scale * 1.4426950408889634f;

    // Q fragments (second MFMA operand: rows = q, d-contiguous), straight from global, pre-scaled to log2 units
    bf16x8 qf[QF][KSTEPS];
    const int q_last = (a.nq + 63) / 64 * 64 - 1;  // last row of this head's (padded) q buffer
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qb + (size_t)min(q0 + f * 16 + fr, q_last) * DPAD + kk * 32 + fg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[f][kk][e] = (bf16)((float)raw[e] * sl2);
        }

    f32x4 oacc[QF][DB];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = 0.f;
        l_run[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d) oacc[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // tile staging: global -> registers (issued a tile ahead) -> LDS
    bf16x8 kreg[CPT], vreg[CPT];
    const bf16* kptr[CPT];
    const bf16* vptr[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = i * 256 + tid;
        kptr[i] = kb + (size_t)(c / (DPAD / 8)) * DPAD + (c % (DPAD / 8)) * 8;   // K tile: KT rows x DPAD/8 chunks
        vptr[i] = vb + (size_t)(c >> 3) * a.nkpad + (c & 7) * 8;                 // V^T tile: DPAD rows x 8 chunks
    }
    auto fetch = [&]() {  // called with consecutive tiles: the pointers just advance
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            kreg[i] = *reinterpret_cast<const bf16x8*>(kptr[i]);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vptr[i]);
            kptr[i] += KT * DPAD;
            vptr[i] += KT;
        }
    };
    const bf16 one = (bf16)1.0f;
    auto commit = [&](int buf) {
        char* k_l = k_lds2[buf];
        char* v_l = v_lds2[buf];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            {
                const int row = c / (DPAD / 8), ch = c % (DPAD / 8);
                *reinterpret_cast<bf16x8*>(k_l + row * KP + ch * 16) = kreg[i];
            }
            {
                const int row = c >> 3, ch = c & 7;
                bf16x8 v = vreg[i];
                if (ONES && row == a.hd) v = bf16x8{one, one, one, one, one, one, one, one};  // the row-sum row of V^T
                *reinterpret_cast<bf16x8*>(v_l + row * VP + ch * 16) = v;
            }
        }
    };

    const int ntiles = (a.nk + KT - 1) / KT;
    fetch();
    commit(0);
    if (ntiles > 1) fetch();
    for (int t = 0; t < ntiles; ++t) {
        const int kt0 = t * KT;
        const char* k_lds = k_lds2[t & 1];
        const char* v_lds = v_lds2[t & 1];
        __syncthreads();  // tile t is visible; every wave has finished tile t - 1 (whose buffer receives tile t + 1 now)
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch();
        }
        // the tile's work as a function of its number of 16-key blocks: 4 (a full tile), or 1 when the LAST tile holds at most 16 keys
        // - the 77-key cross-attention is 64 + 13 keys: its second tile then costs a quarter of the S MFMAs and exps and half the P V
        // MFMAs instead of a whole masked tile (16 of the 32 attention launches of a step)
        auto tile_work = [&](auto ttc) {
            constexpr int TT = decltype(ttc)::value;
            // ---- S^T - m for 64 keys x (QF*16) queries: 4 key blocks of 16 ----
            f32x4 s[QF][TT];
    #pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
    #pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + (tt * 16 + fr) * KP + (kk * 32 + fg * 8) * 2);
    #pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        const float nm = -m_run[f];
                        const f32x4 c = kk == 0 ? f32x4{nm, nm, nm, nm} : s[f][tt];
                        s[f][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[f][kk], c, 0, 0, 0);
                    }
                }
            }
            const bool ragged = kt0 + KT > a.nk;  // only the last tile can hold masked keys (uniform branch)
            uint4v pw[QF][2];  // P^T fragments as packed bf16 pairs (second MFMA operand)
    #pragma unroll
            for (int f = 0; f < QF; ++f) {
                if (ragged) {
    #pragma unroll
                    for (int tt = 0; tt < TT; ++tt)
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kt0 + tt * 16 + fg * 4 + r >= a.nk) s[f][tt][r] = -INFINITY;
                }
                // lane owns q = fr; its 16 values are keys 16tt + 4fg + r, already relative to the running reference
                float mx = fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3]));
    #pragma unroll
                for (int tt = 1; tt < TT; ++tt) {
                    mx = fmaxf(fmaxf(mx, s[f][tt][0]), s[f][tt][1]);
                    mx = fmaxf(fmaxf(mx, s[f][tt][2]), s[f][tt][3]);
                }
                // move the reference only when it is needed: first tile (any sign), or some score of the tile is beyond 2^THR - tested
                // on the lane-LOCAL maxima (no lane above THR <=> no score above THR), so the cross-lane exchange that makes the four
                // lanes of a query agree on the new reference is paid only inside the rarely taken branch
                if (t == 0 || __builtin_amdgcn_ballot_w64(mx > THR) != 0) {
                    mx = xor_max(mx);
                    const float dlt = t == 0 ? mx : fmaxf(mx, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-dlt);  // (tile 0: the accumulators are zero, any factor is fine)
                    m_run[f] += dlt;
    #pragma unroll
                    for (int tt = 0; tt < TT; ++tt)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) s[f][tt][r] -= dlt;
                    if (t != 0) {
                        l_run[f] *= alpha;
    #pragma unroll
                        for (int d = 0; d < DB; ++d) oacc[f][d] *= alpha;
                    }
                }
                float ps = 0.f;
                if (TT < 4) pw[f][0] = pw[f][1] = uint4v{0u, 0u, 0u, 0u};  // key blocks past TT: p = 0
    #pragma unroll
                for (int tt = 0; tt < TT; ++tt)
    #pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 p = {__builtin_amdgcn_exp2f(s[f][tt][r]), __builtin_amdgcn_exp2f(s[f][tt][r + 1])};
                        if (!ONES) ps += p[0] + p[1];
                        pw[f][tt >> 1][(tt & 1) * 2 + (r >> 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));  // one v_cvt_pk
                    }
                if (!ONES) l_run[f] += ps;
            }
            // ---- O^T += V^T . P^T : two 32-key steps ----
    #pragma unroll
            for (int sub = 0; sub < (TT + 1) / 2; ++sub) {
    #pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const char* vrow = v_lds + (d * 16 + fr) * VP + (sub * 32 + fg * 4) * 2;
                    const short4v lo = *reinterpret_cast<const short4v*>(vrow);
                    const short4v hi = *reinterpret_cast<const short4v*>(vrow + 32);
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    const short8v packed = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, packed);
    #pragma unroll
                    for (int f = 0; f < QF; ++f)
                        oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pw[f][sub]), oacc[f][d], 0, 0, 0);
                }
            }
        };
        if (a.nk - kt0 <= 16) tile_work(std::integral_constant<int, 1>{});
        else tile_work(std::integral_constant<int, 4>{});
    }
    // ---- normalise and store: lane holds O^T[d = 16db + 4fg + r][q = fr] ----
    const int b = bh / a.H, h = bh - b * a.H;
    bf16* ob = reinterpret_cast<bf16*>(a.out);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float lsum;
        if (ONES) {
            // O^T row hd is the row sum: block hd / 16, lane group (hd % 16) / 4, register hd % 4
            float cand = 0.f;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d * 16 + fg * 4 + r == a.hd) cand = oacc[f][d][r];
            lsum = xor_sum(cand);  // the three other lane groups contribute 0
        } else {
            lsum = xor_sum(l_run[f]);
        }
        const float inv = 1.0f / lsum;
        const int q = q0 + f * 16 + fr;
        if (a.lse && fg == 0 && q <= q_last)  // log2-domain log-sum-exp per query; +inf on the padding rows (P = 0 there)
            a.lse[(size_t)bh * (q_last + 1) + q] = q < a.nq ? m_run[f] + __builtin_amdgcn_logf(lsum) : INFINITY;
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
    static const int xcd_env = [] { const char* e = getenv("MRISR_ATTN_XCD"); return e ? atoi(e) : 1; }();
    const_cast<AttnArgs&>(a).xcd_map = (xcd_env && BH % 8 == 0) ? 1 : 0;
    ProfScope ps("flash_attention", 4.0 * BH * (double)a.nq * a.nk * a.hd,
                 2.0 * BH * (2.0 * a.nq * a.hd + 2.0 * a.nk * a.hd), st);
    // ONES: the row sum rides along the P.V product as row `hd` of V^T - needs a spare row in the last 16-row block
    const bool ones = a.hd < DB * 16;
    // QF = 4 (a wave owns 64 queries): every K / V^T fragment read from LDS serves twice the MFMAs - the fp8 kernel, whose only
    // difference in the key loop is half the LDS bytes, runs 17 % faster than this one, so the loop is not purely VALU bound.
    // Only where the registers allow it (head dims up to 64) and enough query blocks remain.  OFF by default (MRISR_ATTN_QF4=1 turns it on):
    // isolated it is 8 % faster (85.3 -> 78.3 us at N = 1,024), inside the step it is SLOWER - 112 vs 75 us per launch in the replayed graph,
    // 61.7 vs 62.7 slices/s on one box, alternating runs (profiles/r03o_qf4_ab.log): at 236 VGPRs only two waves per SIMD are resident, and
    // the kernel no longer hides the cold Q / K / V^T tiles the QKV GEMM has just written.
    static const int qf4_env = [] { const char* e = getenv("MRISR_ATTN_QF4"); return e ? atoi(e) : 0; }();
    if constexpr (DPAD <= 64) {
        if (qf4_env && a.nq >= 512 && a.nk >= 256 && ones) {  // (77-key cross-attention: 25.0 -> 26.2 us with QF = 4; self-attention at N = 1024: 85.3 -> 78.3 us)
            hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 4, true>), dim3((a.nq + 255) / 256, BH), dim3(256), 0, st, a);
            MRISR_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    if (a.nq >= 128) {
        if (ones) hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 2, true>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 2, false>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
    } else {
        if (ones) hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 1, true>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<DPAD, DB, 1, false>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
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

// ================================================================================================
// fp8 attention (BASELINE configs[4]: "fp8 (CDNA4 MFMA) attention"): Q K^T and P V on v_mfma_f32_16x16x32_fp8_fp8 (OCP e4m3
// operands, f32 accumulate), softmax in f32 exactly as above.  Scales are PER HEAD and chosen so that no score ever needs a
// multiply on the (VALU-bound) softmax path:
//   * Q K^T: with c = sqrt(amax|Q| * scale*log2e / amax|K|) the operands K8 = e4m3(K * c) and Q8 = e4m3(Q * scale*log2e / c) have
//     the same amax and their product is the score in log2 units directly - the matrix core's output (minus the running
//     reference, which rides in as the C operand as before) goes straight into v_exp_f32;
//   * P V: p = 2^(s - m) <= 2^8 fits e4m3 as it is; V8 = e4m3(V / sV), sV = amax|V| / 448, and sV is folded into the final
//     1 / rowsum factor.  The row sum is still one more row of V^T (row hd = 1.0, exact in e4m3), so it is the sum of the
//     QUANTISED p - numerator and denominator see the same rounding.
// A pre-pass (one workgroup per (batch, head)) finds the three amax values and writes K8 / V8^T (half the bytes of the bf16
// tiles); Q is quantised in the attention kernel's own prologue, where it is pre-scaled anyway.  The log-sum-exp output for the
// (bf16) backward is unchanged.
// ================================================================================================
typedef long f8x8;  // 8 e4m3 values = one MFMA operand
struct AttnQuantArgs {
    const void* q; const void* k; const void* vt;
    void* k8; void* vt8;
    float* scales;  // [B*H][4]: Q multiplier (scale*log2e / c), c, sV, unused
    int nqpad, nkpad, dpad;
    float sl2;
};
__device__ __forceinline__ float amax8(const bf16x8& v, float m) {
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
    return m;
}
__device__ __forceinline__ f8x8 quant8_clamped(const bf16x8& x, float mul) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf((float)x[e] * mul, -448.f), 448.f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (f8x8)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
__global__ __launch_bounds__(256) void attn_quant_fp8_kernel(const AttnQuantArgs a) {
    __shared__ float red[3][4];
    const int bh = blockIdx.x, tid = threadIdx.x;
    const bf16* q = reinterpret_cast<const bf16*>(a.q) + (size_t)bh * a.nqpad * a.dpad;
    const bf16* k = reinterpret_cast<const bf16*>(a.k) + (size_t)bh * a.nkpad * a.dpad;
    const bf16* vt = reinterpret_cast<const bf16*>(a.vt) + (size_t)bh * a.dpad * a.nkpad;
    const int nq8 = a.nqpad * a.dpad / 8, nk8 = a.nkpad * a.dpad / 8;
    float mq = 0.f, mk = 0.f, mv = 0.f;
    for (int i = tid; i < nq8; i += 256) mq = amax8(*reinterpret_cast<const bf16x8*>(q + (size_t)i * 8), mq);
    for (int i = tid; i < nk8; i += 256) {
        mk = amax8(*reinterpret_cast<const bf16x8*>(k + (size_t)i * 8), mk);
        mv = amax8(*reinterpret_cast<const bf16x8*>(vt + (size_t)i * 8), mv);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        mq = fmaxf(mq, __shfl_xor(mq, o));
        mk = fmaxf(mk, __shfl_xor(mk, o));
        mv = fmaxf(mv, __shfl_xor(mv, o));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = mq; red[1][tid >> 6] = mk; red[2][tid >> 6] = mv; }
    __syncthreads();
    mq = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    mk = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    mv = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    const float c = (mq > 0.f && mk > 0.f) ? sqrtf(mq * a.sl2 / mk) : 1.0f;
    const float sv = mv > 0.f ? mv * (1.0f / 448.0f) : 1.0f;
    if (tid == 0) {
        float* sc = a.scales + (size_t)bh * 4;
        sc[0] = a.sl2 / c; sc[1] = c; sc[2] = sv; sc[3] = 0.f;
    }
    const float isv = 1.0f / sv;
    char* k8 = reinterpret_cast<char*>(a.k8) + (size_t)bh * a.nkpad * a.dpad;
    char* v8 = reinterpret_cast<char*>(a.vt8) + (size_t)bh * a.dpad * a.nkpad;
    for (int i = tid; i < nk8; i += 256) {  // (re-read: the head's 2 x nkpad x dpad x 2 bytes were just streamed through this CU's L2)
        *reinterpret_cast<f8x8*>(k8 + (size_t)i * 8) = quant8_clamped(*reinterpret_cast<const bf16x8*>(k + (size_t)i * 8), c);
        *reinterpret_cast<f8x8*>(v8 + (size_t)i * 8) = quant8_clamped(*reinterpret_cast<const bf16x8*>(vt + (size_t)i * 8), isv);
    }
}

template <int DPAD, int DB, int QF, bool ONES>
__global__ __launch_bounds__(256) void attn_fwd_fp8_kernel(const AttnArgs a) {
    constexpr int KT = 64;
    constexpr int KSTEPS = DPAD / 32;
    constexpr int KP = DPAD + 16;           // K8 tile row pitch (bytes): 16-byte aligned rows, fragment reads (8 B) <= 2-way conflicts
    constexpr int VP = KT + 16;             // V8^T tile row pitch (bytes): the 4-byte fragment reads of a wave hit 64 distinct banks
    constexpr int NCH = KT * DPAD / 16;     // 16-byte chunks per tile (K and V^T alike)
    constexpr int CPT = (NCH + 255) / 256;
    constexpr float THR = 8.0f;
    __shared__ __attribute__((aligned(16))) char k_lds2[2][KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds2[2][DPAD * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (a.xcd_map) {
        const int L = blockIdx.x + blockIdx.y * gridDim.x;
        const int slot = L >> 3;
        bh = (slot / (int)gridDim.x) * 8 + (L & 7);
        qblk = slot % (int)gridDim.x;
    }
    const int q0 = (qblk * 4 + wave) * (16 * QF);
    const bf16* qb = reinterpret_cast<const bf16*>(a.q) + (size_t)bh * ((a.nq + 63) / 64 * 64) * DPAD;
    const char* kb = reinterpret_cast<const char*>(a.k8) + (size_t)bh * a.nkpad * DPAD;
    const char* vb = reinterpret_cast<const char*>(a.vt8) + (size_t)bh * DPAD * a.nkpad;
    const float qmul = a.f8_scales[(size_t)bh * 4 + 0];
    const float sv = a.f8_scales[(size_t)bh * 4 + 2];

    f8x8 qf[QF][KSTEPS];
    const int q_last = (a.nq + 63) / 64 * 64 - 1;
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
            qf[f][kk] = quant8_clamped(*reinterpret_cast<const bf16x8*>(qb + (size_t)min(q0 + f * 16 + fr, q_last) * DPAD + kk * 32 + fg * 8), qmul);

    f32x4 oacc[QF][DB];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = 0.f;
        l_run[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d) oacc[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    uint4v kreg[CPT], vreg[CPT];
    const char* kptr[CPT];
    const char* vptr[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = min(i * 256 + tid, NCH - 1);
        kptr[i] = kb + (size_t)(c / (DPAD / 16)) * DPAD + (c % (DPAD / 16)) * 16;  // K8 tile: KT rows x DPAD/16 chunks
        vptr[i] = vb + (size_t)(c >> 2) * a.nkpad + (c & 3) * 16;                  // V8^T tile: DPAD rows x 4 chunks
    }
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            kreg[i] = *reinterpret_cast<const uint4v*>(kptr[i]);
            vreg[i] = *reinterpret_cast<const uint4v*>(vptr[i]);
            kptr[i] += KT * DPAD;
            vptr[i] += KT;
        }
    };
    auto commit = [&](int buf) {
        char* k_l = k_lds2[buf];
        char* v_l = v_lds2[buf];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            if (NCH % 256 != 0 && c >= NCH) continue;
            {
                const int row = c / (DPAD / 16), ch = c % (DPAD / 16);
                *reinterpret_cast<uint4v*>(k_l + row * KP + ch * 16) = kreg[i];
            }
            {
                const int row = c >> 2, ch = c & 3;
                uint4v v = vreg[i];
                if (ONES && row == a.hd) v = uint4v{0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u};  // 1.0 in e4m3: the row-sum row
                *reinterpret_cast<uint4v*>(v_l + row * VP + ch * 16) = v;
            }
        }
    };

    const int ntiles = (a.nk + KT - 1) / KT;
    fetch();
    commit(0);
    if (ntiles > 1) fetch();
    for (int t = 0; t < ntiles; ++t) {
        const int kt0 = t * KT;
        const char* k_lds = k_lds2[t & 1];
        const char* v_lds = v_lds2[t & 1];
        __syncthreads();
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch();
        }
        f32x4 s[QF][4];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f8x8 kf = *reinterpret_cast<const f8x8*>(k_lds + (tt * 16 + fr) * KP + kk * 32 + fg * 8);
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    const float nm = -m_run[f];
                    const f32x4 c = kk == 0 ? f32x4{nm, nm, nm, nm} : s[f][tt];
                    s[f][tt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf, qf[f][kk], c, 0, 0, 0);
                }
            }
        }
        const bool ragged = kt0 + KT > a.nk;
        f8x8 pw[QF][2];
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
            for (int tt = 1; tt < 4; ++tt) {
                mx = fmaxf(fmaxf(mx, s[f][tt][0]), s[f][tt][1]);
                mx = fmaxf(fmaxf(mx, s[f][tt][2]), s[f][tt][3]);
            }
            if (t == 0 || __builtin_amdgcn_ballot_w64(mx > THR) != 0) {
                mx = xor_max(mx);
                const float dlt = t == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-dlt);
                m_run[f] += dlt;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[f][tt][r] -= dlt;
                if (t != 0) {
                    l_run[f] *= alpha;
#pragma unroll
                    for (int d = 0; d < DB; ++d) oacc[f][d] *= alpha;
                }
            }
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                int w[2] = {0, 0};
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int tt = sub * 2 + h2;
                    const float p0 = __builtin_amdgcn_exp2f(s[f][tt][0]), p1 = __builtin_amdgcn_exp2f(s[f][tt][1]);
                    const float p2 = __builtin_amdgcn_exp2f(s[f][tt][2]), p3 = __builtin_amdgcn_exp2f(s[f][tt][3]);
                    w[h2] = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, w[h2], false);
                    w[h2] = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, w[h2], true);
                    if (!ONES) {  // the sum of the QUANTISED p (what the P V product sees)
                        const float q0f = __builtin_amdgcn_cvt_f32_fp8(w[h2], 0), q1f = __builtin_amdgcn_cvt_f32_fp8(w[h2], 1);
                        const float q2f = __builtin_amdgcn_cvt_f32_fp8(w[h2], 2), q3f = __builtin_amdgcn_cvt_f32_fp8(w[h2], 3);
                        ps += (q0f + q1f) + (q2f + q3f);
                    }
                }
                pw[f][sub] = (f8x8)(((unsigned long)(unsigned)w[1] << 32) | (unsigned)w[0]);
            }
            if (!ONES) l_run[f] += ps;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const char* vrow = v_lds + (d * 16 + fr) * VP + sub * 32 + fg * 4;
                const unsigned lo = *reinterpret_cast<const unsigned*>(vrow);
                const unsigned hi = *reinterpret_cast<const unsigned*>(vrow + 16);
                const f8x8 vf = (f8x8)(((unsigned long)hi << 32) | lo);
#pragma unroll
                for (int f = 0; f < QF; ++f) oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf, pw[f][sub], oacc[f][d], 0, 0, 0);
            }
        }
    }

    const int b = bh / a.H, h = bh - b * a.H;
    bf16* ob = reinterpret_cast<bf16*>(a.out);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float lsum;
        if (ONES) {
            float cand = 0.f;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d * 16 + fg * 4 + r == a.hd) cand = oacc[f][d][r];
            lsum = xor_sum(cand);
        } else {
            lsum = xor_sum(l_run[f]);
        }
        const float inv = sv / lsum;  // V8 = V / sV
        const int q = q0 + f * 16 + fr;
        if (a.lse && fg == 0 && q <= q_last)
            a.lse[(size_t)bh * (q_last + 1) + q] = q < a.nq ? m_run[f] + __builtin_amdgcn_logf(lsum) : INFINITY;
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
static int launch_dpad_fp8(const AttnArgs& a, hipStream_t st) {
    const int BH = a.B * a.H;
    static const int xcd_env = [] { const char* e = getenv("MRISR_ATTN_XCD"); return e ? atoi(e) : 1; }();
    const_cast<AttnArgs&>(a).xcd_map = (xcd_env && BH % 8 == 0) ? 1 : 0;
    ProfScope ps("flash_attention_fp8", 4.0 * BH * (double)a.nq * a.nk * a.hd, 1.0 * BH * (2.0 * 2.0 * a.nq * a.hd + 2.0 * a.nk * a.hd), st);
    const bool ones = a.hd < DB * 16;
    if (a.nq >= 128) {
        if (ones) hipLaunchKernelGGL((attn_fwd_fp8_kernel<DPAD, DB, 2, true>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_fp8_kernel<DPAD, DB, 2, false>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
    } else {
        if (ones) hipLaunchKernelGGL((attn_fwd_fp8_kernel<DPAD, DB, 1, true>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_fp8_kernel<DPAD, DB, 1, false>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

// a.k8 / a.vt8: B*H*nkpad*dpad bytes each; a.f8_scales: B*H*4 floats (scratch of the caller)
int launch_attention_fp8(const AttnArgs& a, hipStream_t st) {
    MRISR_REQUIRE(a.nkpad % 64 == 0 && a.nk >= 1 && a.nk <= a.nkpad, "attention: key padding");
    MRISR_REQUIRE(a.hd % 4 == 0 && a.hd <= a.dpad && a.dpad % 32 == 0, "attention: head dim");
    MRISR_REQUIRE(a.k8 && a.vt8 && a.f8_scales, "fp8 attention: scratch for the quantised K / V^T and the per-head scales");
    {
        AttnQuantArgs qa;
        qa.q = a.q; qa.k = a.k; qa.vt = a.vt; qa.k8 = a.k8; qa.vt8 = a.vt8; qa.scales = a.f8_scales;
        qa.nqpad = (a.nq + 63) / 64 * 64; qa.nkpad = a.nkpad; qa.dpad = a.dpad;
        qa.sl2 = a.scale * 1.4426950408889634f;
        const int BH = a.B * a.H;
        ProfScope ps("attention_quant_fp8", 0.0, (double)BH * a.dpad * (2.0 * qa.nqpad + 2.0 * 2.0 * a.nkpad + 2.0 * a.nkpad), st);
        hipLaunchKernelGGL(attn_quant_fp8_kernel, dim3(BH), dim3(256), 0, st, qa);
        MRISR_CHECK_HIP(hipGetLastError());
    }
    const int db = (a.hd + 15) / 16;
#define ATT_CASE8(DP, DBV) if (a.dpad == DP && db == DBV) return launch_dpad_fp8<DP, DBV>(a, st);
    ATT_CASE8(32, 1) ATT_CASE8(32, 2)
    ATT_CASE8(64, 3) ATT_CASE8(64, 4)
    ATT_CASE8(96, 5) ATT_CASE8(96, 6)
    ATT_CASE8(128, 7) ATT_CASE8(128, 8)
    ATT_CASE8(160, 9) ATT_CASE8(160, 10)
#undef ATT_CASE8
    MRISR_REQUIRE(false, "fp8 attention: unsupported (padded head dim, head dim) combination");
    return 0;
}

// ================================================================================================
// backward.  Two passes over the same tile loop, P recomputed from the log-sum-exp instead of stored:
//   MODE 0 (dQ):     a wave OWNS 16*QF queries (Q, dO rows in registers) and streams key tiles
//                    S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - D) * scale,  dQ^T += K^T dS^T
//   MODE 1 (dK, dV): a wave OWNS 16*QF keys (K, V rows in registers) and streams query tiles
//                    S = Q K^T,  dP = dO V^T,  dS = P o (dP - D) * scale,  dK^T += Q^T dS,  dV^T += dO^T P
// Same transposed trick as the forward: the lane that owns column `fr` of S holds exactly the k-slice the second MFMA
// wants, so P / dS never leave registers; the streamed side comes through LDS both row-major (for S, dP) and
// transposed (for the accumulating products).  No atomics: each output element has one owner.
// ================================================================================================
template <int DPAD, int DB, int KT, int QF, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdArgs a) {
    constexpr int KSTEPS = DPAD / 32;
    constexpr int TT = KT / 16;            // 16-row blocks of the streamed tile
    constexpr int SUBS = KT / 32;          // 32-deep steps of the accumulating products
    constexpr int KP = DPAD * 2 + 16;      // row-major tile pitch (bytes)
    constexpr int VP = KT * 2 + 16;        // transposed tile pitch (bytes)
    constexpr int NCH = KT * DPAD / 8;     // 16-byte chunks per tile (either orientation)
    constexpr int CPT = (NCH + 255) / 256;
    constexpr int NT = MODE == 1 ? 2 : 1;  // transposed tiles
    __shared__ __attribute__((aligned(16))) char y_lds[2][KT * KP];
    __shared__ __attribute__((aligned(16))) char yt_lds[NT][DPAD * VP];
    __shared__ float l_lds[KT], d_lds[KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bh = blockIdx.y, oblk = blockIdx.x;
    if ((gridDim.y & 7) == 0) {  // all owner blocks of a head on one XCD (as in the forward): the streamed tiles hit its L2
        const int L = blockIdx.x + blockIdx.y * gridDim.x;
        const int slot = L >> 3;
        bh = (slot / (int)gridDim.x) * 8 + (L & 7);
        oblk = slot % (int)gridDim.x;
    }
    const int own0 = (oblk * 4 + wave) * (16 * QF);
    const int own_pad = MODE == 0 ? a.npad : a.nkpad, own_valid = MODE == 0 ? a.nq : a.nk;
    const int str_pad = MODE == 0 ? a.nkpad : a.npad, str_valid = MODE == 0 ? a.nk : a.nq;
    const bf16* x1 = reinterpret_cast<const bf16*>(MODE == 0 ? a.q : a.k) + (size_t)bh * own_pad * DPAD;
    const bf16* x2 = reinterpret_cast<const bf16*>(MODE == 0 ? a.doh : a.v) + (size_t)bh * own_pad * DPAD;
    const bf16* y1 = reinterpret_cast<const bf16*>(MODE == 0 ? a.k : a.q) + (size_t)bh * str_pad * DPAD;
    const bf16* y2 = reinterpret_cast<const bf16*>(MODE == 0 ? a.v : a.doh) + (size_t)bh * str_pad * DPAD;
    const bf16* y1t = reinterpret_cast<const bf16*>(MODE == 0 ? a.kt : a.qt) + (size_t)bh * DPAD * str_pad;
    const bf16* y2t = reinterpret_cast<const bf16*>(a.doht) + (size_t)bh * DPAD * str_pad;  // MODE 1 only
    const float* lse = a.lse + (size_t)bh * a.npad;
    const float* dsum = a.dsum + (size_t)bh * a.npad;
    const float sl2 = a.scale * 1.4426950408889634f;

    bf16x8 x1f[QF][KSTEPS], x2f[QF][KSTEPS];
    float Lq[QF], Dq[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int row = min(own0 + f * 16 + fr, own_pad - 1);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            x1f[f][kk] = *reinterpret_cast<const bf16x8*>(x1 + (size_t)row * DPAD + kk * 32 + fg * 8);
            x2f[f][kk] = *reinterpret_cast<const bf16x8*>(x2 + (size_t)row * DPAD + kk * 32 + fg * 8);
        }
        Lq[f] = MODE == 0 ? lse[row] : 0.f;
        Dq[f] = MODE == 0 ? dsum[row] : 0.f;
    }
    f32x4 acc1[QF][DB], acc2[QF][DB];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int d = 0; d < DB; ++d) acc1[f][d] = acc2[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 yreg[2][CPT], ytreg[NT][CPT];
    float lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int s0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            if (NCH % 256 == 0 || c < NCH) {
                {
                    const int row = c / (DPAD / 8), ch = c % (DPAD / 8);
                    yreg[0][i] = *reinterpret_cast<const bf16x8*>(y1 + (size_t)(s0 + row) * DPAD + ch * 8);
                    yreg[1][i] = *reinterpret_cast<const bf16x8*>(y2 + (size_t)(s0 + row) * DPAD + ch * 8);
                }
                {
                    const int row = c / (KT / 8), ch = c % (KT / 8);
                    ytreg[0][i] = *reinterpret_cast<const bf16x8*>(y1t + (size_t)row * str_pad + s0 + ch * 8);
                    if (MODE == 1) ytreg[NT - 1][i] = *reinterpret_cast<const bf16x8*>(y2t + (size_t)row * str_pad + s0 + ch * 8);
                }
            }
        }
        if (MODE == 1 && tid < KT) {
            lreg = lse[s0 + tid];
            dreg = dsum[s0 + tid];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = i * 256 + tid;
            if (NCH % 256 == 0 || c < NCH) {
                {
                    const int row = c / (DPAD / 8), ch = c % (DPAD / 8);
                    *reinterpret_cast<bf16x8*>(y_lds[0] + row * KP + ch * 16) = yreg[0][i];
                    *reinterpret_cast<bf16x8*>(y_lds[1] + row * KP + ch * 16) = yreg[1][i];
                }
                {
                    const int row = c / (KT / 8), ch = c % (KT / 8);
                    *reinterpret_cast<bf16x8*>(yt_lds[0] + row * VP + ch * 16) = ytreg[0][i];
                    if (MODE == 1) *reinterpret_cast<bf16x8*>(yt_lds[NT - 1] + row * VP + ch * 16) = ytreg[NT - 1][i];
                }
            }
        }
        if (MODE == 1 && tid < KT) {
            l_lds[tid] = lreg;
            d_lds[tid] = dreg;
        }
    };

    const int ntiles = (str_valid + KT - 1) / KT;
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int s0 = t * KT;
        __syncthreads();
        commit();
        __syncthreads();
        if (t + 1 < ntiles) fetch(s0 + KT);
        f32x4 sv[QF][TT], dp[QF][TT];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) sv[f][tt] = dp[f][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(y_lds[0] + (tt * 16 + fr) * KP + (kk * 32 + fg * 8) * 2);
                const bf16x8 f2 = *reinterpret_cast<const bf16x8*>(y_lds[1] + (tt * 16 + fr) * KP + (kk * 32 + fg * 8) * 2);
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    sv[f][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, x1f[f][kk], sv[f][tt], 0, 0, 0);
                    dp[f][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2, x2f[f][kk], dp[f][tt], 0, 0, 0);
                }
            }
        }
        // lane holds streamed rows 16tt + 4fg + r of owner column fr
        bf16x8 dsf[QF][SUBS], pf[QF][SUBS];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            f32x4 Lv, Dv;
            if (MODE == 1) {
                Lv = *reinterpret_cast<const f32x4*>(&l_lds[tt * 16 + fg * 4]);
                Dv = *reinterpret_cast<const f32x4*>(&d_lds[tt * 16 + fg * 4]);
            }
#pragma unroll
            for (int f = 0; f < QF; ++f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float L = MODE == 0 ? Lq[f] : Lv[r];
                    const float D = MODE == 0 ? Dq[f] : Dv[r];
                    float p = __builtin_amdgcn_exp2f(fmaf(sv[f][tt][r], sl2, -L));
                    if (MODE == 0 && s0 + tt * 16 + fg * 4 + r >= str_valid) p = 0.f;  // masked key
                    const float ds = p * (dp[f][tt][r] - D) * a.scale;
                    dsf[f][tt >> 1][(tt & 1) * 4 + r] = (bf16)ds;
                    pf[f][tt >> 1][(tt & 1) * 4 + r] = (bf16)p;
                }
            }
        }
#pragma unroll
        for (int sub = 0; sub < SUBS; ++sub) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                typedef __attribute__((ext_vector_type(8))) short short8v;
                {
                    const char* row = yt_lds[0] + (d * 16 + fr) * VP + (sub * 32 + fg * 4) * 2;
                    const short4v lo = *reinterpret_cast<const short4v*>(row);
                    const short4v hi = *reinterpret_cast<const short4v*>(row + 32);
                    const short8v packed = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 tf = __builtin_bit_cast(bf16x8, packed);
#pragma unroll
                    for (int f = 0; f < QF; ++f) acc1[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, dsf[f][sub], acc1[f][d], 0, 0, 0);
                }
                if (MODE == 1) {
                    const char* row = yt_lds[NT - 1] + (d * 16 + fr) * VP + (sub * 32 + fg * 4) * 2;
                    const short4v lo = *reinterpret_cast<const short4v*>(row);
                    const short4v hi = *reinterpret_cast<const short4v*>(row + 32);
                    const short8v packed = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 tf = __builtin_bit_cast(bf16x8, packed);
#pragma unroll
                    for (int f = 0; f < QF; ++f) acc2[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, pf[f][sub], acc2[f][d], 0, 0, 0);
                }
            }
        }
    }
    // ---- store: lane holds out^T[d = 16db + 4fg + r][owner = fr] ----
    const int b = bh / a.H, h = bh - b * a.H;
    const int ld = MODE == 0 ? a.ldq : a.ldkv;
    bf16* o1 = reinterpret_cast<bf16*>(MODE == 0 ? a.dq : a.dk);
    bf16* o2 = reinterpret_cast<bf16*>(a.dv);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int own = own0 + f * 16 + fr;
        if (own >= own_valid) continue;
        const size_t base = ((size_t)b * own_valid + own) * ld + h * a.hd;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int dd = d * 16 + fg * 4;
            if (dd >= a.hd) continue;
            const bf16x4 v1 = {(bf16)acc1[f][d][0], (bf16)acc1[f][d][1], (bf16)acc1[f][d][2], (bf16)acc1[f][d][3]};
            *reinterpret_cast<bf16x4*>(o1 + base + dd) = v1;
            if (MODE == 1) {
                const bf16x4 v2 = {(bf16)acc2[f][d][0], (bf16)acc2[f][d][1], (bf16)acc2[f][d][2], (bf16)acc2[f][d][3]};
                *reinterpret_cast<bf16x4*>(o2 + base + dd) = v2;
            }
        }
    }
}

template <int DPAD, int DB>
static int launch_bwd_dpad(const AttnBwdArgs& a, hipStream_t st) {
    constexpr int KT = DPAD <= 96 ? 64 : 32;
    const int BH = a.B * a.H;
    bool big = false;
    {
        ProfScope ps("flash_attention_bwd_dq", 6.0 * BH * (double)a.nq * a.nk * a.hd, 2.0 * BH * (3.0 * a.nq * a.hd + 3.0 * a.nk * a.hd), st);
        if constexpr (DPAD <= 64) {
            if (a.nq >= 128) big = true;
        }
        if constexpr (DPAD <= 64) {
            if (big) hipLaunchKernelGGL((attn_bwd_kernel<DPAD, DB, KT, 2, 0>), dim3((a.nq + 127) / 128, BH), dim3(256), 0, st, a);
        }
        if (!big)
            hipLaunchKernelGGL((attn_bwd_kernel<DPAD, DB, KT, 1, 0>), dim3((a.nq + 63) / 64, BH), dim3(256), 0, st, a);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    if (a.dk) {
        ProfScope ps("flash_attention_bwd_dkv", 8.0 * BH * (double)a.nq * a.nk * a.hd, 2.0 * BH * (4.0 * a.nq * a.hd + 4.0 * a.nk * a.hd), st);
        bool big2 = false;
        if constexpr (DPAD <= 64) {
            if (a.nk >= 128) {
                big2 = true;
                hipLaunchKernelGGL((attn_bwd_kernel<DPAD, DB, KT, 2, 1>), dim3((a.nk + 127) / 128, BH), dim3(256), 0, st, a);
            }
        }
        if (!big2)
            hipLaunchKernelGGL((attn_bwd_kernel<DPAD, DB, KT, 1, 1>), dim3((a.nk + 63) / 64, BH), dim3(256), 0, st, a);
        MRISR_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int launch_attention_bwd_bf16(const AttnBwdArgs& a, hipStream_t st) {
    MRISR_REQUIRE(a.nkpad % 64 == 0 && a.npad % 64 == 0 && a.nk >= 1 && a.nk <= a.nkpad && a.nq >= 1 && a.nq <= a.npad, "attention bwd: padding");
    MRISR_REQUIRE(a.hd % 4 == 0 && a.hd <= a.dpad && a.ldq % 4 == 0 && a.ldkv % 4 == 0, "attention bwd: head dim");
    MRISR_REQUIRE(a.q && a.k && a.v && a.doh && a.kt && a.lse && a.dsum && a.dq, "attention bwd: operands");
    MRISR_REQUIRE(!a.dk || (a.dv && a.qt && a.doht), "attention bwd: dK/dV operands");
    const int db = (a.hd + 15) / 16;
#define ATT_CASE(DP, DBV) if (a.dpad == DP && db == DBV) return launch_bwd_dpad<DP, DBV>(a, st);
    ATT_CASE(32, 1) ATT_CASE(32, 2)
    ATT_CASE(64, 3) ATT_CASE(64, 4)
    ATT_CASE(96, 5) ATT_CASE(96, 6)
    ATT_CASE(128, 7) ATT_CASE(128, 8)
    ATT_CASE(160, 9) ATT_CASE(160, 10)
#undef ATT_CASE
    MRISR_REQUIRE(false, "attention bwd: unsupported (padded head dim, head dim) combination");
    return 0;
}

// one wave per token row: lanes sweep the row's 8-byte chunks (coalesced), per-head partial dot products fold through LDS
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16* __restrict__ dO, const bf16* __restrict__ O, bf16* __restrict__ doh,
                                                            float* __restrict__ dsum, int B, int N, int H, int hd, int npad, int dpad) {
    __shared__ float hs[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = blockIdx.x * 4ll + wave;  // b * N + q
    const bool live = row < (long long)B * N;
    if (lane < H) hs[wave][lane] = 0.f;
    __syncthreads();
    const int q = live ? (int)(row % N) : 0, b = live ? (int)(row / N) : 0;
    if (live) {
        const int C = H * hd;
        const bf16* dr = dO + row * C;
        const bf16* orow = O + row * C;
        for (int c = lane * 4; c < C; c += 256) {
            const bf16x4 dv = *reinterpret_cast<const bf16x4*>(dr + c);
            const bf16x4 ov = *reinterpret_cast<const bf16x4*>(orow + c);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += (float)dv[e] * (float)ov[e];
            const int h = c / hd, dd = c - h * hd;
            *reinterpret_cast<bf16x4*>(doh + (((size_t)b * H + h) * npad + q) * dpad + dd) = dv;
            atomicAdd(&hs[wave][h], s);
        }
    }
    __syncthreads();
    if (live && lane < H) dsum[((size_t)b * H + lane) * npad + q] = hs[wave][lane];
}
int launch_attention_bwd_prep(const void* dO_rows, const void* O_rows, void* doh, float* dsum, int B, int N, int H, int hd, int npad,
                              int dpad, hipStream_t st) {
    MRISR_REQUIRE(hd % 4 == 0 && dpad % 4 == 0 && H <= 64, "attention bwd prep: head dim / head count");
    const long long rows = (long long)B * N;
    ProfScope ps("flash_attention_bwd_prep", 0.0, 6.0 * rows * H * hd, st);
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const bf16*>(dO_rows),
                       reinterpret_cast<const bf16*>(O_rows), reinterpret_cast<bf16*>(doh), dsum, B, N, H, hd, npad, dpad);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace mrisr
