// mrisr - the row-local middle of a BasicTransformerBlock at C = 320 in ONE kernel ("xtail"):
//
//     x1  = attn1.to_out(ao) + x0                    (ao: the self-attention output rows, x0: the residual stream)
//     q   = attn2.to_q(LayerNorm2(x1))
//     o   = softmax(q K_ctx^T / sqrt(d)) V_ctx        (8 heads of 40 channels, <= 80 prompt tokens, K / V cached per prompt)
//     x2  = attn2.to_out(o) + x1                     -> written over x0
//
// Reference: diffusers' BasicTransformerBlock between `attn1` and `norm3` (third-party, restated by the test oracle); the
// call site is the UNet forward behind src/adapters/res_srdiff.py:73-78.  Unfused this is four launches (row-panel GEMM + residual,
// LayerNorm + row-panel GEMM, flash attention over 77 keys, row-panel GEMM + residual: 105 us at M = 32,768), each of which
// streams the 21 MB row tensor through HBM once or twice and pays its own load -> first MFMA -> last store latency chain.
//
// Everything after the self-attention is per token, so a workgroup can own 128 rows (a wave: 32 rows = two 16-row MFMA fragments)
// from `ao` to `x2` without the rows ever leaving its registers:
//   * the row-panel scheme of gemm_rp_kernel / mlp_fused_kernel (gemm.hip): rows resident in registers in MFMA operand layout,
//     weights streamed through LDS by LDS-DMA in double-buffered 40 KB chunks (64 output columns x K = 320), one barrier per chunk;
//   * a GEMM's accumulators (lane: 4 consecutive columns of row fr) ARE the next GEMM's row operand once the K order inside
//     every 32-block is permuted (slot 8 fg + e  <->  column 32 kk + (e < 4 ? 4 fg + e : 16 + 4 fg + e - 4)); the weights of
//     to_q / to_out(attn2) are packed with that permutation (launch_pack_mlp_w2), the cached K / V are packed per (image, head
//     pair) as ready-made LDS images (launch_pack_xattn_kv: permuted, zero outside the head's 40 columns, pre-swizzled);
//   * a head's 40 columns always lie inside two consecutive 32-column K steps of q (40 h mod 32 + 40 <= 64), so Q K^T is
//     5 key fragments x 2 K steps; all <= 80 keys of a row are in registers at once: plain softmax, no running maximum; P V is
//     3 output blocks (the 16-column blocks the head touches) x 3 K steps; a block shared by two heads (columns 32-47, ...) is
//     simply accumulated by both, P being normalised before the product;
//   * rank-4 LoRA on all three projections as in gemm_rp_kernel: z = x A^T from the resident rows (A rows in LDS, natural K order:
//     for the permuted operands the fragment is read as two 8-byte halves), B as one more MFMA K step per chunk;
//   * LayerNorm2 on the bf16-rounded x1 in registers (statistics: two cross-lane adds), x1 kept as the residual of the last GEMM.
// 19 chunks (5 + 5 + 4 + 5) in one double-buffered stream; every small global load (bias, LoRA B, residual) is issued one chunk
// ahead of its use so that the `s_waitcnt vmcnt(0)` at the top of a chunk never waits on a fresh round trip.
// One wave per SIMD (launch bound (256, 1)): a wave may use the whole 512-register file - x1 (80) + LayerNorm'ed rows / q (80) +
// attention output (80) + accumulators.
#include <algorithm>

#include "common.h"
#include "prof.h"

namespace mrisr {

typedef __attribute__((address_space(3))) void* xt_lds_ptr_t;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t xt_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ void xt_dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (xt_lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
#define XT_OOB 0x80000000u
typedef __attribute__((ext_vector_type(4))) unsigned xt_u4;

#ifndef XT_PF
#define XT_PF 2
#endif
#ifndef XT_INTERLEAVE
#define XT_INTERLEAVE 1
#endif
constexpr int XT_KVP_PAIR = 40960;     // bytes of one (image, head pair) K / V image
constexpr int XT_KVP_HEAD = 19456;     // K part 80 keys x 64 slots (10,240 B) + V part 48 columns x 96 slots (9,216 B)
constexpr int XT_KVP_KPART = 10240;

struct XTailDev {
    const void* ao; int ldao;
    void* t; int ldt;
    int M, ntok;
    const void* w1; const float* b1; const void* a1; const float* lb1;   // attn1.to_out: W natural K order
    const float* ln_g; const float* ln_b; float ln_eps;
    const void* wq; const float* bq; const void* aq; const float* lbq;   // attn2.to_q: W K-permuted
    const void* kvp; unsigned kvp_bytes; int nk; float sl2;              // sl2 = scale * log2(e)
    const void* w2; const float* b2; const void* a2; const float* lb2;   // attn2.to_out: W K-permuted
    int lora_r;  // 4 or 0 (all three projections alike)
    int poison;
    unsigned long long* stamps;  // probe: per workgroup 40 clock stamps (s_memtime) at the phase boundaries, wave 0
    int rot;     // per-workgroup rotation of the piece order inside a chunk
    int dbg;     // probe builds of the timing (MRISR_XTAIL_DBG; results are garbage): 1 no weight / K-V DMA after chunk 0, 2 no attention math, 4 no GEMM K loops
};

template <bool LORA>
__global__ __launch_bounds__(256, 1) void xattn_tail_kernel(const XTailDev a) {
    constexpr int K = 320, KS = 10, MF = 2, NF = 4, CH = 64 * K * 2, CPR = 40;
    constexpr int LA_OFF = 2 * CH, LA_SZ = 12288, GB_OFF = LA_OFF + 3 * LA_SZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const int img = m0 / a.ntok;
    int stamp_k = 0;
    auto stamp = [&]() {
        if (a.stamps) {
            if (tid == 0) a.stamps[(size_t)blockIdx.x * 40 + stamp_k] = __builtin_readcyclecounter();
            ++stamp_k;
        }
    };
    auto substamp = [&](int k) { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 40 + k] = __builtin_readcyclecounter(); };
    stamp();
    if (a.poison) {
        const unsigned bytes = ((const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_dispatch_ptr())[7];
        for (unsigned o = threadIdx.x * 16u; o + 16u <= bytes; o += 256u * 16u) *reinterpret_cast<uint4*>(smem + o) = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rao = xt_rsrc(a.ao, (unsigned)min((long long)a.M * a.ldao * 2, 0x7FFFFFFFll));
    const __amdgpu_buffer_rsrc_t rw1 = xt_rsrc(a.w1, (unsigned)(K * K * 2));
    const __amdgpu_buffer_rsrc_t rwq = xt_rsrc(a.wq, (unsigned)(K * K * 2));
    const __amdgpu_buffer_rsrc_t rw2 = xt_rsrc(a.w2, (unsigned)(K * K * 2));
    const __amdgpu_buffer_rsrc_t rkv = xt_rsrc(a.kvp, a.kvp_bytes);

    // ---- DMA geometry of a weight chunk: LDS position L (16-byte units) = row * 40 + (c ^ (row & 7)) ----
    // Every workgroup streams the SAME chunk at about the same time; walked in the same order by all 32 CUs of an XCD the requests
    // pile up on one L2 channel after the other.  Each workgroup therefore walks the 40 pieces of a chunk in its own rotation
    // (piece = 4 q + w with q rotated by the workgroup id / 4 and w by the workgroup id): MRISR_XTAIL_ROT=0 turns it off.
    unsigned wvo[10];
    int ldo[10];
    const int rot_w = a.rot ? (int)(blockIdx.x & 3) : 0, rot_q = a.rot ? (int)((blockIdx.x >> 2) % 10) : 0;
#pragma unroll
    for (int p = 0; p < 10; ++p) {
        int q = p + rot_q;
        if (q >= 10) q -= 10;
        const int piece = q * 4 + ((wave + rot_w) & 3);
        const int L = piece * 64 + lane;
        const int row = L / CPR, cs = L - row * CPR;
        wvo[p] = (unsigned)(row * (K * 2) + (cs ^ (row & 7)) * 16);
        ldo[p] = __builtin_amdgcn_readfirstlane(piece * 1024);
    }
    auto stage_w = [&](const __amdgpu_buffer_rsrc_t& rw, int c, int buf) {
        if (a.dbg & 1) return;
#pragma unroll
        for (int p = 0; p < 10; ++p) xt_dma16(rw, smem + buf * CH + ldo[p], wvo[p], (unsigned)c * (unsigned)CH);
    };
    auto stage_kv = [&](int pair, int buf) {  // a ready-made LDS image: linear copy
        const unsigned base = (unsigned)(img * 4 + pair) * (unsigned)XT_KVP_PAIR;
        if (a.dbg & 1) return;
#pragma unroll
        for (int p = 0; p < 10; ++p) xt_dma16(rkv, smem + buf * CH + ldo[p], (unsigned)(ldo[p] + lane * 16), base);
    };
    // The NEXT chunk is not staged in one go at the top of a chunk: clock stamps showed the ten DMA instructions of a wave taking ~2,100
    // cycles to ISSUE (every CU of an XCD pulls the same 40 KB through the same L2 at the same moment: 1.3 MB per chunk step against
    // ~1 KB / clk of L2 hit bandwidth), during which the wave issues nothing else - as long as the whole K loop (1,950 cycles).  One piece
    // per K step instead: the MFMAs of the previous step run while the wave sits in the VMEM queue.
    struct Next { __amdgpu_buffer_rsrc_t r; unsigned soff; int kv; int live; int buf; };
    auto next_w = [&](const __amdgpu_buffer_rsrc_t& rw, int c, int buf) { return Next{rw, (unsigned)c * (unsigned)CH, 0, 1, buf}; };
    auto next_kv = [&](int pair, int buf) { return Next{rkv, (unsigned)(img * 4 + pair) * (unsigned)XT_KVP_PAIR, 1, 1, buf}; };
    auto issue_piece = [&](const Next& nx, int p) {
        if (!nx.live || (a.dbg & 1)) return;
        const unsigned vo = nx.kv ? (unsigned)(ldo[p] + lane * 16) : wvo[p];
        xt_dma16(nx.r, smem + nx.buf * CH + ldo[p], vo, nx.soff);
    };
    // chunk stream: 0-4 to_out(attn1), 5-9 to_q, 10-13 K / V head pairs, 14-18 to_out(attn2)
    auto stage = [&](int cid, int buf) {
        if (cid < 5) stage_w(rw1, cid, buf);
        else if (cid < 10) stage_w(rwq, cid - 5, buf);
        else if (cid < 14) stage_kv(cid - 10, buf);
        else if (cid < 19) stage_w(rw2, cid - 14, buf);
    };

    // ---- LoRA A rows (3 x [16][320], rows >= r zero) and the LayerNorm vectors: dedicated LDS regions, loaded once ----
    if (LORA) {
        const unsigned abytes = (unsigned)(a.lora_r * K * 2);
        const __amdgpu_buffer_rsrc_t ra1 = xt_rsrc(a.a1, abytes), raq = xt_rsrc(a.aq, abytes), ra2 = xt_rsrc(a.a2, abytes);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int piece = p * 4 + wave;  // 0..11 (10 real)
            const int L = piece * 64 + lane;
            const int row = L / CPR, cs = L - row * CPR;
            const unsigned vo = (piece < 10 && row < a.lora_r) ? (unsigned)(row * (K * 2) + (cs ^ (row & 7)) * 16) : XT_OOB;
            xt_dma16(ra1, smem + LA_OFF + piece * 1024, vo, 0u);
            xt_dma16(raq, smem + LA_OFF + LA_SZ + piece * 1024, vo, 0u);
            xt_dma16(ra2, smem + LA_OFF + 2 * LA_SZ + piece * 1024, vo, 0u);
        }
    }
    {
        const __amdgpu_buffer_rsrc_t rg = xt_rsrc(a.ln_g, (unsigned)(K * 4));
        const __amdgpu_buffer_rsrc_t rb = xt_rsrc(a.ln_b, (unsigned)(K * 4));
        const int pc = wave & 1;
        xt_dma16(wave < 2 ? rg : rb, smem + GB_OFF + (wave >> 1) * 2048 + pc * 1024, (unsigned)(pc * 1024 + lane * 16), 0u);
    }

    // ---- the panel rows of `ao` through LDS (the ring is still empty) into MFMA operand layout ----
    const bf16* tp = reinterpret_cast<const bf16*>(a.t);
    bf16x4 rv[NF][MF];  // residual (x0) of the CURRENT chunk of the first GEMM, in accumulator layout; loaded a chunk ahead
    auto load_resid = [&](int c, bf16x4 (&dst)[NF][MF]) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = m0 + wave * 32 + j * 16 + fr;
                dst[i][j] = *reinterpret_cast<const bf16x4*>(tp + (size_t)m * a.ldt + c * 64 + i * 16 + fg * 4);
            }
    };
    bf16x8 af[MF][KS];
    {
        char* areg = smem + wave * (32 * K * 2);
#pragma unroll
        for (int p = 0; p < 32 * CPR / 64; ++p) {
            const int L = p * 64 + lane;
            const int row = L / CPR, cs = L - row * CPR;
            const int m = m0 + wave * 32 + row;
            const unsigned vo = m < a.M ? (unsigned)(((size_t)m * a.ldao + (cs ^ (row & 7)) * 8) * 2) : XT_OOB;
            xt_dma16(rao, areg + p * 1024, vo, 0u);
        }
        load_resid(0, rv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                af[j][kk] = *reinterpret_cast<const bf16x8*>(areg + (j * 16 + fr) * (K * 2) + (((kk * 4 + fg) ^ (fr & 7)) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();  // every wave has its rows in registers (and every wave's LoRA / LayerNorm pieces have landed)
    }
    stamp();  // 1: rows resident
    stage(0, 0);

    // ---- building blocks ----
    // bias + LoRA B of a chunk, loaded one chunk ahead
    float pbn[NF][4];
    f32x4 lbn[NF];
    auto prefetch_cols = [&](const float* bias, const float* lb, int c) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c * 64 + i * 16 + fg * 4);
            pbn[i][0] = b[0]; pbn[i][1] = b[1]; pbn[i][2] = b[2]; pbn[i][3] = b[3];
            if (LORA) lbn[i] = *reinterpret_cast<const f32x4*>(lb + (size_t)(c * 64 + i * 16 + fr) * 4);
        }
    };
    // z = x A^T (rank <= 4, one 16-column fragment) -> the row operand of the up-projection K step
    auto lora_z = [&](const char* abase, const bf16x8 (&x)[MF][KS], const bool perm, bf16x8 (&zf)[MF]) {
        f32x4 zacc[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) zacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* sl = abase + fr * (K * 2);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 lf;
            if (!perm) {
                lf = *reinterpret_cast<const bf16x8*>(sl + (((kk * 4 + fg) ^ (fr & 7)) * 16));
            } else {  // slots 0-3: columns 32 kk + 4 fg .., slots 4-7: columns 32 kk + 16 + 4 fg ..
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sl + (((kk * 4 + (fg >> 1)) ^ (fr & 7)) * 16) + (fg & 1) * 8);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sl + (((kk * 4 + 2 + (fg >> 1)) ^ (fr & 7)) * 16) + (fg & 1) * 8);
                lf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < MF; ++j) zacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, x[j][kk], zacc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < MF; ++j)
            zf[j] = bf16x8{(bf16)zacc[j][0], (bf16)zacc[j][1], (bf16)zacc[j][2], (bf16)zacc[j][3], (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    };
    // one 64-column chunk: acc = bias + x W^T (+ z B^T)
    auto gemm_chunk = [&](int buf, const bf16x8 (&x)[MF][KS], const float (&pb)[NF][4], const f32x4 (&lb)[NF], const bf16x8 (&zf)[MF], f32x4 (&acc)[NF][MF],
                          auto&& per_step, const bool probe = false) {
        const char* sw = smem + buf * CH + fr * (K * 2);
        if (a.dbg & 4) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) per_step(kk);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{pb[i][0], pb[i][1], pb[i][2], pb[i][3]};
            return;
        }
        // weight fragments software-pipelined XT_PF K steps ahead: with one wave per SIMD nothing else hides the ds_read latency, and one
        // K step of 8 MFMAs (128 cycles) is shorter than it (clock stamps: 4,400 cycles per chunk at depth 1 = 31 % of the MFMA time)
        constexpr int PF = XT_PF;
        bf16x8 wf[PF + 1][NF];
        auto load_w = [&](bf16x8 (&dst)[NF], int kk) {
            const int off = ((kk * 4 + fg) ^ (fr & 7)) * 16;
#pragma unroll
            for (int i = 0; i < NF; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(sw + i * 16 * (K * 2) + off);
        };
#pragma unroll
        for (int kk = 0; kk < PF; ++kk) load_w(wf[kk], kk);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (probe && (kk == 0 || kk == 5)) substamp(kk == 0 ? 33 : 34);
            per_step(kk);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + PF < KS) load_w(wf[(kk + PF) % (PF + 1)], kk + PF);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const f32x4 cz = kk == 0 ? f32x4{pb[i][0], pb[i][1], pb[i][2], pb[i][3]} : acc[i][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk % (PF + 1)][i], x[j][kk], cz, 0, 0, 0);
                }
            // the four fragment reads of step kk + PF between the MFMA pairs of step kk: a read's issue (the four waves share the LDS
            // array) then hides behind a running MFMA instead of holding up the first one (clock stamps: 196 -> cycles per K step)
            if (XT_INTERLEAVE && kk + PF < KS) {
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (probe) substamp(35);
        if (LORA) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                bf16x8 lbf = bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                if (fg == 0) { lbf[0] = (bf16)lb[i][0]; lbf[1] = (bf16)lb[i][1]; lbf[2] = (bf16)lb[i][2]; lbf[3] = (bf16)lb[i][3]; }
#pragma unroll
                for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbf, zf[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto chunk_top = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk (and the small loads issued with it) has landed: issued a whole chunk ago
        __syncthreads();                                   // ... for every wave; every wave has left the previous chunk's buffer
    };

    // The chunk loops below are REAL loops (not unrolled): fully unrolled the kernel is ~100 KB of straight-line code that every wave
    // executes once - it ran at the instruction-fetch rate (90 us per launch, 12 % of the MFMA time).  Register arrays cannot be indexed by
    // the loop counter, so the row arrays ROTATE instead: every iteration consumes the front two K steps / appends its two new K steps at
    // the back (64 v_mov per chunk), and after five iterations the array is in natural order again.
    auto rotate2 = [&](bf16x8 (&x)[MF][KS], const bf16x8 (&nw)[MF][2]) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
#pragma unroll
            for (int kk = 0; kk + 2 < KS; ++kk) x[j][kk] = x[j][kk + 2];
            x[j][KS - 2] = nw[j][0];
            x[j][KS - 1] = nw[j][1];
        }
    };
    auto swap_halves = [&](bf16x8 (&x)[MF][KS]) {
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int kk = 0; kk < KS / 2; ++kk) { const bf16x8 tmp = x[j][kk]; x[j][kk] = x[j][kk + KS / 2]; x[j][kk + KS / 2] = tmp; }
    };
    const bf16x8 zero8 = bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};

    // =========================== GEMM 1: x1 = to_out(ao) + x0 ===========================
    bf16x8 zf[MF];
    if (LORA) lora_z(smem + LA_OFF, af, false, zf);
    prefetch_cols(a.b1, a.lb1, 0);
    stamp();  // 2: z of the first projection
    bf16x8 x1f[MF][KS];  // x1 (bf16) in the permuted operand layout = accumulator layout of fragment pairs
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) x1f[j][kk] = zero8;
#pragma nounroll
    for (int c = 0; c < 5; ++c) {
        const int buf = c & 1;
        chunk_top();
        float pb[NF][4];
        f32x4 lb[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[i][r] = pbn[i][r];
            if (LORA) lb[i] = lbn[i];
        }
        bf16x4 rvn[NF][MF];
        const Next nx = c + 1 < 5 ? next_w(rw1, c + 1, buf ^ 1) : next_w(rwq, 0, buf ^ 1);
        f32x4 acc[NF][MF];
        gemm_chunk(buf, af, pb, lb, zf, acc, [&](int kk) {
            issue_piece(nx, kk);
            if (kk == 1) {
                load_resid(min(c + 1, 4), rvn);
                if (c + 1 < 5) prefetch_cols(a.b1, a.lb1, c + 1); else prefetch_cols(a.bq, a.lbq, 0);
            }
        });
        bf16x8 nw[MF][2];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16 v = (bf16)acc[i][j][r];                                  // the projection's bf16 output ...
                    nw[j][i >> 1][(i & 1) * 4 + r] = (bf16)((float)v + (float)rv[i][j][r]);  // ... plus the residual, rounded again (as the unfused epilogue)
                }
        rotate2(x1f, nw);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) rv[i][j] = rvn[i][j];
        stamp();  // 3-7
    }
    // =========================== LayerNorm 2 on the resident x1 ===========================
    bf16x8 xn[MF][KS];
    {
        float mean[MF], rstd[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const float c0 = __shfl((float)x1f[j][0][0], fr);  // the row's element 0 (slot 0 of K step 0 in the lane with fg = 0)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)x1f[j][kk][e] - c0; s1 += d; s2 = fmaf(d, d, s2); }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float md = s1 * (1.0f / K);
            mean[j] = c0 + md;
            rstd[j] = rsqrtf(fmaxf(s2 * (1.0f / K) - md * md, 0.f) + a.ln_eps);
        }
        const float* gp = reinterpret_cast<const float*>(smem + GB_OFF) + fg * 4;
        const float* bp = reinterpret_cast<const float*>(smem + GB_OFF + 2048) + fg * 4;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp + kk * 32), g1 = *reinterpret_cast<const f32x4*>(gp + kk * 32 + 16);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + kk * 32), b1 = *reinterpret_cast<const f32x4*>(bp + kk * 32 + 16);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
                    o[e] = (bf16)(((float)x1f[j][kk][e] - mean[j]) * rstd[j] * ga + be);
                }
                xn[j][kk] = o;
            }
        }
    }
    // =========================== GEMM 2: q = to_q(LayerNorm2(x1)) ===========================
    stamp();  // 8: LayerNorm
    if (LORA) lora_z(smem + LA_OFF + LA_SZ, xn, true, zf);
    stamp();  // 9
    bf16x8 qf[MF][KS];
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[j][kk] = zero8;
#pragma nounroll
    for (int c = 0; c < 5; ++c) {
        const int buf = (c + 1) & 1;  // chunk id 5 + c
        if (c == 2) substamp(25);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c == 2) substamp(26);
        __syncthreads();
        if (c == 2) substamp(27);
        float pb[NF][4];
        f32x4 lb[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[i][r] = pbn[i][r];
            if (LORA) lb[i] = lbn[i];
        }
        const Next nx = c + 1 < 5 ? next_w(rwq, c + 1, buf ^ 1) : next_kv(0, buf ^ 1);
        if (c == 2) substamp(28);
        f32x4 acc[NF][MF];
        gemm_chunk(buf, xn, pb, lb, zf, acc, [&](int kk) {
            issue_piece(nx, kk);
            if (kk == 1 && c + 1 < 5) prefetch_cols(a.bq, a.lbq, c + 1);
        });
        if (c == 2) substamp(29);
        bf16x8 nw[MF][2];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) nw[j][i >> 1][(i & 1) * 4 + r] = (bf16)acc[i][j][r];
        rotate2(qf, nw);
        stamp();  // 10-14
    }
    // =========================== cross-attention over the cached prompt K / V ===========================
    // Heads come in groups of four = 160 columns = five K steps, after which the column pattern repeats: the loop body handles one group
    // (two K / V chunks) with static indices into the FRONT halves of q and o, and the halves are swapped after each group.
    bf16x8 of[MF][KS];
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) of[j][kk] = zero8;
#pragma nounroll
    for (int grp = 0; grp < 2; ++grp) {
        f32x4 carry[MF];
#pragma unroll
        for (int j = 0; j < MF; ++j) carry[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int buf = pp;  // chunk id 10 + 2 grp + pp
            chunk_top();
            const Next nx = pp == 0 ? next_kv(2 * grp + 1, buf ^ 1) : (grp == 0 ? next_kv(2, buf ^ 1) : next_w(rw2, 0, buf ^ 1));
            if (pp == 1 && grp == 1) prefetch_cols(a.b2, a.lb2, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (a.dbg & 2) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) issue_piece(nx, s * 5 + q);
                    continue;
                }
                const int hl = 2 * pp + s;          // head inside the group: columns 40 hl .. 40 hl + 39 of the group's 160
                const int kk0 = (40 * hl) / 32;     // 0, 1, 2, 3
                const char* hb = smem + buf * CH + s * XT_KVP_HEAD;
                // S = Q K^T: 5 key fragments x 2 K steps
                f32x4 sacc[5][MF];
                {
                    const char* kb = hb + fr * 128;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        issue_piece(nx, s * 5 + t);
                        bf16x8 kf[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) kf[i] = *reinterpret_cast<const bf16x8*>(kb + i * 16 * 128 + (((t * 4 + fg) ^ (fr & 7)) * 16));
#pragma unroll
                        for (int i = 0; i < 5; ++i)
#pragma unroll
                            for (int j = 0; j < MF; ++j) {
                                const f32x4 cz = t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sacc[i][j];
                                sacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[j][kk0 + t], cz, 0, 0, 0);
                            }
                    }
                }
                // softmax over the keys of each row (a row's keys: 5 fragments x 4 registers in each of the 4 lanes {fr + 16 fg'})
                bf16x8 pk[3][MF];
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    // raw scores: max first (the scale is positive), then exp2(s * sl2 - max * sl2) as one FMA + v_exp per key; only the
                    // last key fragment can hold keys >= nk when nk > 64 (the K image is zero there: a finite score, masked here)
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 5; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (i == 4 || a.nk <= 64) { if (i * 16 + fg * 4 + r >= a.nk) sacc[i][j][r] = -INFINITY; }
                            mx = fmaxf(mx, sacc[i][j][r]);
                        }
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float nm = -mx * a.sl2;
                    float l = 0.f;
#pragma unroll
                    for (int i = 0; i < 5; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(sacc[i][j][r], a.sl2, nm));
                            sacc[i][j][r] = e;
                            l += e;
                        }
                    l += __shfl_xor(l, 16);
                    l += __shfl_xor(l, 32);
                    const float inv = 1.0f / l;
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int i = 2 * t + (e >> 2);
                            pk[t][j][e] = i < 5 ? (bf16)(sacc[i < 5 ? i : 0][j][e & 3] * inv) : (bf16)0.f;
                        }
                }
                // O = P V: the 3 column blocks the head touches x 3 K steps of keys
                f32x4 oacc[3][MF];
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int j = 0; j < MF; ++j) oacc[b][j] = (s == 1 && b == 0) ? carry[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    const char* vb = hb + XT_KVP_KPART + fr * 192;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        issue_piece(nx, s * 5 + 2 + t);
                        bf16x8 vf[3];
#pragma unroll
                        for (int b = 0; b < 3; ++b) vf[b] = *reinterpret_cast<const bf16x8*>(vb + b * 16 * 192 + (((t * 4 + fg) ^ ((fr >> 2) & 3)) * 16));
#pragma unroll
                        for (int b = 0; b < 3; ++b)
#pragma unroll
                            for (int j = 0; j < MF; ++j) oacc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[b], pk[t][j], oacc[b][j], 0, 0, 0);
                    }
                }
                // finished blocks -> the row operand of the last GEMM; the block an even head shares with the next head is carried
                const int gb0 = (40 * hl) / 16;     // 0, 2, 5, 7 inside the group's ten blocks
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (s == 0 && b == 2) {
#pragma unroll
                        for (int j = 0; j < MF; ++j) carry[j] = oacc[b][j];
                    } else {
                        const int gb = gb0 + b;
#pragma unroll
                        for (int j = 0; j < MF; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) of[j][gb >> 1][(gb & 1) * 4 + r] = (bf16)oacc[b][j][r];
                    }
                }
            }
        }
        swap_halves(qf);
        swap_halves(of);
        stamp();  // 15-16
    }
    // =========================== GEMM 3: x2 = to_out(o) + x1 ===========================
    if (LORA) lora_z(smem + LA_OFF + 2 * LA_SZ, of, true, zf);
    stamp();  // 17
#pragma nounroll
    for (int c = 0; c < 5; ++c) {
        const int buf = c & 1;  // chunk id 14 + c
        if (c == 4) substamp(30);
        // the eight x2 stores of the previous chunk are the youngest VMEM operations of this wave and may stay in flight (vmcnt retires in
        // order: everything older - this chunk's DMA pieces, the bias / LoRA loads - has landed once at most eight are outstanding)
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        if (c == 4) substamp(31);
        float pb[NF][4];
        f32x4 lb[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[i][r] = pbn[i][r];
            if (LORA) lb[i] = lbn[i];
        }
        Next nx = next_w(rw2, min(c + 1, 4), buf ^ 1);
        nx.live = c + 1 < 5;
        f32x4 acc[NF][MF];
        gemm_chunk(buf, of, pb, lb, zf, acc, [&](int kk) {
            issue_piece(nx, kk);
            if (kk == 1 && c + 1 < 5) prefetch_cols(a.b2, a.lb2, c + 1);
        }, c == 4);
        if (c == 4) substamp(32);
        bf16x8 nw[MF][2];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16 v = (bf16)acc[i][j][r];
                    nw[j][i >> 1][(i & 1) * 4 + r] = (bf16)((float)v + (float)x1f[j][i >> 1][(i & 1) * 4 + r]);  // the front two K steps: this chunk's x1
                }
        // x2 leaves chunk by chunk, straight from the accumulator layout (8 bytes per lane; the four fg lanes x four fragments of a row
        // fill one 128-byte line within this chunk, L2 merges them): the write is spread over the GEMM instead of one 21 MB burst of
        // all workgroups at the end (clock stamps: staging + whole-row stores took as long as three chunks)
        {
            bf16* ob = reinterpret_cast<bf16*>(a.t);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = m0 + wave * 32 + j * 16 + fr;
                    const bf16x8 v = nw[j][i >> 1];
                    const bf16x4 o = (i & 1) ? bf16x4{v[4], v[5], v[6], v[7]} : bf16x4{v[0], v[1], v[2], v[3]};
                    if (m < a.M) *reinterpret_cast<bf16x4*>(ob + (size_t)m * a.ldt + c * 64 + i * 16 + fg * 4) = o;
                }
        }
        rotate2(x1f, nw);
        stamp();  // 18-22
    }
    stamp();  // 23: stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();  // 24: stores done
}

// K [B*8][ctx_pad][dpad] and V^T [B*8][dpad][ctx_pad] of the prompt (bf16, head-major, as the attention kernel reads them) ->
// per (image, head pair) the 40 KB LDS image the kernel above DMAs linearly: for each of the two heads
//   K part  [80 keys][64 slots]: slot 32 t + 8 fg + e = column 32 (kk0 + t) + (e < 4 ? 4 fg + e : 16 + 4 fg + e - 4) of q's row,
//           zero outside the head's 40 columns and for keys >= nk; 16-byte chunk c of key row r at position c ^ (r & 7);
//   V part  [48 columns][96 slots]: the three 16-column blocks of the 320-wide output row the head touches; slot 32 t + 8 fg + e =
//           key 32 t + (e < 4 ? 4 fg + e : 16 + 4 fg + e - 4); zero outside the head / beyond nk; chunk c of row r at c ^ ((r >> 2) & 3).
__global__ __launch_bounds__(256) void pack_xattn_kv_kernel(const bf16* __restrict__ kc, const bf16* __restrict__ vtc, bf16* __restrict__ dst,
                                                            int ctx_pad, int dpad, int nk) {
    const int b = blockIdx.x >> 2, pair = blockIdx.x & 3;
    bf16* out = dst + (size_t)blockIdx.x * (XT_KVP_PAIR / 2);
    for (int L = threadIdx.x; L < XT_KVP_PAIR / 16; L += 256) {
        const int byte = L * 16;
        bf16x8 v = bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        const int s = byte / XT_KVP_HEAD;
        if (s < 2) {
            const int rem = byte - s * XT_KVP_HEAD;
            const int h = 2 * pair + s;
            const size_t bh = (size_t)b * 8 + h;
            if (rem < XT_KVP_KPART) {
                const int key = rem / 128, cp = (rem % 128) / 16;
                const int c = cp ^ (key & 7);
                const int kk0 = (40 * h) / 32;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sg = c * 8 + e, t = sg >> 5, fgs = (sg & 31) >> 3;
                    const int col = 32 * (kk0 + t) + (e < 4 ? 4 * fgs + e : 16 + 4 * fgs + e - 4);
                    const int d = col - 40 * h;
                    if (d >= 0 && d < 40 && key < nk) v[e] = kc[(bh * ctx_pad + key) * dpad + d];
                }
            } else {
                const int rem2 = rem - XT_KVP_KPART;
                const int row = rem2 / 192, cp = (rem2 % 192) / 16;
                const int c = cp ^ ((row >> 2) & 3);
                const int col = 16 * ((40 * h) / 16 + row / 16) + (row & 15);
                const int d = col - 40 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sg = c * 8 + e, t = sg >> 5, fgs = (sg & 31) >> 3;
                    const int key = 32 * t + (e < 4 ? 4 * fgs + e : 16 + 4 * fgs + e - 4);
                    if (d >= 0 && d < 40 && key < nk) v[e] = vtc[(bh * dpad + d) * ctx_pad + key];
                }
            }
        }
        *reinterpret_cast<bf16x8*>(out + (size_t)L * 8) = v;
    }
}

size_t xattn_tail_kv_bytes(int B) { return (size_t)B * 4 * XT_KVP_PAIR; }

int launch_pack_xattn_kv(const void* kc, const void* vtc, void* dst, int B, int ctx_pad, int dpad, int nk, hipStream_t st) {
    MRISR_REQUIRE(nk >= 1 && nk <= 80 && nk <= ctx_pad && dpad >= 40, "xattn tail: at most 80 prompt tokens, 40-channel heads");
    hipLaunchKernelGGL(pack_xattn_kv_kernel, dim3(B * 4), dim3(256), 0, st, reinterpret_cast<const bf16*>(kc), reinterpret_cast<const bf16*>(vtc),
                       reinterpret_cast<bf16*>(dst), ctx_pad, dpad, nk);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

static unsigned long long* g_xt_stamps = nullptr;
extern "C" void mrisr_debug_xattn_tail_stamps(void* dev_buf) { g_xt_stamps = static_cast<unsigned long long*>(dev_buf); }
static int g_xtail = -1;  // test hook: -1 = MRISR_XTAIL (default 1), 0 off, 1 on
extern "C" void mrisr_debug_xattn_tail(int on) { g_xtail = on; }
bool xattn_tail_enabled() {
    static const int env = [] { const char* e = getenv("MRISR_XTAIL"); return e ? atoi(e) : 1; }();
    return g_xtail < 0 ? env != 0 : g_xtail != 0;
}
bool xattn_tail_ok(int C, int heads, int M, int ntok, int nk) {
    return C == 320 && heads == 8 && M > 0 && M % 128 == 0 && ntok % 128 == 0 && M % ntok == 0 && nk >= 1 && nk <= 80;
}

int xattn_tail_flags();  // gemm.hip: the GEMM debug flags (LDS poison)

constexpr int XT_SMEM = 2 * 40960 + 3 * 12288 + 4096;
// launch attributes (dynamic LDS above 64 KB); also called from gemm_prepare() so that it never happens inside a stream capture
int xattn_tail_prepare() {
    static bool attr = false;
    if (!attr) {
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XT_SMEM));
        MRISR_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XT_SMEM));
        attr = true;
    }
    return 0;
}

int launch_xattn_tail(const XTailArgs& x, hipStream_t st) {
    MRISR_REQUIRE(xattn_tail_ok(320, 8, x.M, x.ntok, x.nk), "xattn tail: C = 320, 8 heads, whole 128-row panels inside one image, <= 80 keys");
    MRISR_REQUIRE(x.ao && x.t && x.w1 && x.wq && x.w2 && x.kvp && x.ln_g && x.ln_b && x.ldao % 8 == 0 && x.ldt % 8 == 0, "xattn tail: operands");
    MRISR_REQUIRE(x.lora_r == 0 || (x.lora_r == 4 && x.a1 && x.aq && x.a2 && x.lb1 && x.lbq && x.lb2), "xattn tail: rank-4 adapters on all three projections, or none");
    constexpr int smem = XT_SMEM;
    if (xattn_tail_prepare()) return 1;
    const float* zp = static_cast<const float*>(zero_page());
    MRISR_REQUIRE(zp != nullptr, "zero page not initialised");
    XTailDev d;
    d.ao = x.ao; d.ldao = x.ldao; d.t = x.t; d.ldt = x.ldt; d.M = x.M; d.ntok = x.ntok;
    d.w1 = x.w1; d.b1 = x.b1 ? x.b1 : zp; d.a1 = x.a1; d.lb1 = x.lb1;
    d.ln_g = x.ln_g; d.ln_b = x.ln_b; d.ln_eps = x.ln_eps;
    d.wq = x.wq; d.bq = x.bq ? x.bq : zp; d.aq = x.aq; d.lbq = x.lbq;
    d.kvp = x.kvp; d.kvp_bytes = (unsigned)std::min<size_t>(xattn_tail_kv_bytes(x.M / x.ntok), 0x7FFFFFFFu); d.nk = x.nk;
    d.sl2 = x.scale * 1.4426950408889634f;
    d.w2 = x.w2; d.b2 = x.b2 ? x.b2 : zp; d.a2 = x.a2; d.lb2 = x.lb2;
    d.lora_r = x.lora_r;
    d.poison = (xattn_tail_flags() & 2048) ? 1 : 0;
    static const int dbg = [] { const char* e = getenv("MRISR_XTAIL_DBG"); return e ? atoi(e) : 0; }();
    d.dbg = dbg;
    static const int rot = [] { const char* e = getenv("MRISR_XTAIL_ROT"); return e ? atoi(e) : 1; }();
    d.rot = rot;
    d.stamps = g_xt_stamps;
    const double fl = 2.0 * x.M * (3.0 * 320 * 320 + 2.0 * 320 * x.nk) + (x.lora_r ? 2.0 * x.M * 3 * 2 * 320 * 4 : 0.0);
    const double by = 2.0 * x.M * 320 * 3 + 3.0 * 320 * 320 * 2;
    ProfScope ps("xattn_tail_c320", fl, by, st);
    if (x.lora_r) hipLaunchKernelGGL(xattn_tail_kernel<true>, dim3(x.M / 128), dim3(256), smem, st, d);
    else hipLaunchKernelGGL(xattn_tail_kernel<false>, dim3(x.M / 128), dim3(256), smem, st, d);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace mrisr
