// Plain bf16 GEMM micro-benchmark: the 256 x 256 tile, 8-wave, 8-phase schedule of guides/cdna_hip_programming.md ("The 256^2
// 8-phase template": counted vmcnt, raw s_barrier, half-tile stages, LDS-DMA, s_setprio around the MFMA clusters), written from
// that description for VERDICT r2 item 6: reproduce >= 1.3 PFLOP/s at 4096^3 on random operands BEFORE calling the conv kernels'
// 0.9-1.26 PF a ceiling.   C[M][N] = A[M][K] . B[N][K]^T, bf16 operands (both K-contiguous, as the library's weights are), f32
// accumulate, bf16 out.   Build: hipcc -O3 --offload-arch=gfx950 tools/gemm8p/gemm8p.hip -o tools/gemm8p/gemm8p
//
// Schedule (one workgroup = one 256 x 256 tile, 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 8 x 4 MFMA blocks):
//   * LDS: two K tiles (BK = 64) x {A, B} x two 128-row HALF-tiles of 16 KiB = 128 KiB.  A half-tile is one stage item: two
//     16-byte LDS-DMA instructions per thread; the image is lane-linear, the XOR swizzle (16-byte slot ^ (row & 7)) sits on the
//     SOURCE address and on the fragment reads.
//   * a K tile is four PHASES of 16 MFMAs per wave (one quadrant 64 x 32 of the wave's tile x K = 64).  Phase body:
//         s_waitcnt lgkmcnt(0)           the fragments read during the previous phase are in registers
//         [s_waitcnt vmcnt(4)]           phases 3 and 4 only: this wave's pieces of the K tile about to be read have landed
//         s_barrier                      ... and every other wave's; every wave's previous reads have returned (WAR for the stage)
//         stage item phi + 7             (2 LDS-DMA) into the half-tile whose last reads returned before this barrier
//         4 or 8 ds_read_b128            the operands of the NEXT phase, into registers the running MFMAs do not use
//         s_setprio 1; 16 MFMA; s_setprio 0
//     ONE barrier per phase; the DMA stays in flight across them (3-6 phases between a stage and its first read); vmcnt never 0 in
//     the loop.  Quadrant walk: even K tiles (0,0) (0,1) (1,1) (1,0), odd K tiles (0,1) (0,0) (1,0) (1,1) - consecutive phases share
//     one operand, and the operand a phase loads always goes into the register set that became free one phase earlier (two A sets,
//     two B sets: 96 VGPRs of fragments + 128 of accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <type_traits>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;     // one half-tile: 128 rows x 128 B = 16 KiB
constexpr int TILE = 4 * HALF;         // one K tile: A0h A1h B0h B1h
constexpr int LDS_BYTES = 2 * TILE;    // 128 KiB

// 16-byte slot swizzle of a 128-byte LDS row.  SWZ16 = 1: slot ^ ((row >> 1) & 7) - the 16 lanes of a ds_read_b128 group (16 consecutive rows, one
// k-chunk) land on 16 distinct 16-byte slots of the 256-byte bank row (two LDS rows per bank row: the row's parity picks the half, (row >> 1) & 7
// the slot): conflict-free.  SWZ16 = 0: slot ^ (row & 7) - rows r and r + 8 share a slot: 2-way conflict per group (what the library's kernels use).
#ifndef SWZ16
#define SWZ16 1
#endif
#if SWZ16
#define SWZ(row) (((row) >> 1) & 7)
#else
#define SWZ(row) ((row) & 7)
#endif
#ifndef SETPRIO
#define SETPRIO 1
#endif

template <int STAGGER>
__global__ __launch_bounds__(512) void gemm8p_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware, bijective: each XCD gets a contiguous run of tiles; inside it walk 4 M tiles x all N tiles first
    const int ntn = N / BN, ntm = M / BM, nwg = ntn * ntm;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = logical / ntn, tn = logical - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)((size_t)M * K * 2));
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B, (unsigned)((size_t)N * K * 2));

    // ---- stage geometry: half-tile chunk c = tid + 512 i (i = 0, 1): LDS byte c * 16; row = c >> 3, slot = c & 7 holds source k-chunk slot ^ (row & 7)
    // per-lane part: (row * K + kc * 8) * 2 with row = tid >> 3 - the SAME for A and B, for both halves and for the second chunk
    // (row + 64: same row & 7); everything else - tile origin, half, second chunk, K tile - is a scalar offset
    const unsigned vo = (unsigned)((((size_t)(tid >> 3)) * K + (((tid & 7) ^ SWZ(tid >> 3)) * 8)) * 2);
    const unsigned sa0 = (unsigned)((size_t)m0 * K * 2), sb0 = (unsigned)((size_t)n0 * K * 2);
    const unsigned s64 = (unsigned)((size_t)64 * K * 2), s128 = 2 * s64;
    const int nkt = K / BK;
    // stage item sigma = 4 U + j of K tile U: j = 0: B0h, 1: A0h, 2: A1h, 3: B1h.  LDS: buffer (U & 1); order inside a buffer: A0h A1h B0h B1h.
    // j is a compile-time constant of the phase position (no per-lane or per-item branching around the DMA); items past the last
    // K tile are simply not issued (the waits of the last K tiles drain with vmcnt(0) instead of counting on them).
    auto stage = [&](auto jc, int U) {
        constexpr int j = decltype(jc)::value;
        // items past the last K tile re-load an earlier K tile of the same parity into the (dead) half-tile: nobody reads it, and
        // the loop stays straight-line code with exact vmcnt counts to its end (no conditional issue, no tail variants)
        char* buf = smem + (U & 1) * TILE;
        const unsigned soff = (unsigned)(U < nkt ? U : U - 2) * (BK * 2);
        if constexpr (j == 0) { dma16(rb, buf + 2 * HALF + wave * 1024, vo, sb0 + soff); dma16(rb, buf + 2 * HALF + 8192 + wave * 1024, vo, sb0 + s64 + soff); }
        else if constexpr (j == 1) { dma16(ra, buf + wave * 1024, vo, sa0 + soff); dma16(ra, buf + 8192 + wave * 1024, vo, sa0 + s64 + soff); }
        else if constexpr (j == 2) { dma16(ra, buf + HALF + wave * 1024, vo, sa0 + s128 + soff); dma16(ra, buf + HALF + 8192 + wave * 1024, vo, sa0 + s128 + s64 + soff); }
        else { dma16(rb, buf + 3 * HALF + wave * 1024, vo, sb0 + s128 + soff); dma16(rb, buf + 3 * HALF + 8192 + wave * 1024, vo, sb0 + s128 + s64 + soff); }
    };
#define JC(j) std::integral_constant<int, (j)>{}

    // ---- fragment reads.  A: this wave's 128 rows = half-tile A[wr]; register half mh = rows mh*64 .. +63 (4 blocks of 16).
    //      B: cols wc*64 .. +63 = rows (wc & 1)*64 .. of half-tile B[wc >> 1]; register half nh = 32 cols (2 blocks of 16).
    bf16x8 Ax[4][2], Ay[4][2], B0[2][2], B1[2][2];  // [block][k-step]
    const int a_base = wr * HALF + fr * 128;
    const int b_base = (2 + (wc >> 1)) * HALF + ((wc & 1) * 64 + fr) * 128;
    const int sw0 = ((0 * 4 + fg) ^ SWZ(fr)) * 16, sw1 = ((1 * 4 + fg) ^ SWZ(fr)) * 16;  // SWZ(row) == SWZ(fr): blocks are 16-row aligned
    auto read_a = [&](bf16x8 (&dst)[4][2], int buf, int mh) {
        const char* p = smem + buf * TILE + a_base + mh * 64 * 128;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            dst[mb][0] = *reinterpret_cast<const bf16x8*>(p + mb * 16 * 128 + sw0);
            dst[mb][1] = *reinterpret_cast<const bf16x8*>(p + mb * 16 * 128 + sw1);
        }
    };
    auto read_b = [&](bf16x8 (&dst)[2][2], int buf, int nh) {
        const char* p = smem + buf * TILE + b_base + nh * 32 * 128;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            dst[nb][0] = *reinterpret_cast<const bf16x8*>(p + nb * 16 * 128 + sw0);
            dst[nb][1] = *reinterpret_cast<const bf16x8*>(p + nb * 16 * 128 + sw1);
        }
    };
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2], int mh, int nh) {
#if SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mh][nh][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nb][kk], a[mb][kk], acc[mh][nh][mb][nb], 0, 0, 0);
#if SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // lgkmcnt(0) as the BUILTIN (0xC07F): hipcc's own wait-count model then knows the LDS queue is empty (an inline-asm wait it
    // does not see, and it would add its own lgkmcnt(0) in front of the MFMAs - behind the reads issued for the next phase)
#define PHASE_HEAD(VM)                                                   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
    if (VM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             \
    __builtin_amdgcn_s_barrier();                                        \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: items 0..6 (K tile 0 whole, K tile 1 without its B1h), then the operands of phase 0
    stage(JC(0), 0); stage(JC(1), 0); stage(JC(2), 0); stage(JC(3), 0);
    stage(JC(0), 1); stage(JC(1), 1); stage(JC(2), 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K tile 0 (items 0-3): the three younger items may stay in flight
    __builtin_amdgcn_s_barrier();
    read_a(Ax, 0, 0);
    read_b(B0, 0, 0);

    // phase position pos = 0..7 inside an (even, odd) K-tile pair starting at T: stage item sigma = 4 T + pos + 7.  nkt is even;
    // the reads issued in the last two phases of the last pair fetch a K tile that does not exist (unused registers).
    if constexpr (STAGGER == 2) {
    // ---- LONG-INTERVAL staggered form: ONE L and ONE M interval per K tile.  L(T): all 24 fragment reads of K tile T (16 A + 8 B:
    // 96 VGPRs - no second register set: inside a wave L and M alternate, only the two wave GROUPS overlap); M(T): 64 MFMAs = 1,024
    // cycles of the SIMD's matrix pipe.  Group g runs L(T) in interval 2T + g and M(T) in 2T + g + 1, so in every interval one wave of
    // each SIMD streams MFMAs while the other reads LDS; two barriers per K tile instead of eight.
    //   * K tile T + 1 goes into the buffer K tile T - 1 was read from (last reads: group 1, interval 2T - 1): BOTH groups issue their
    //     share of it in interval 2T (group 0 at the start of L(T), group 1 at the start of M(T - 1)) and BOTH wait for it at the end of
    //     interval 2T + 1 (group 0: end of M(T), group 1: end of L(T)) - nothing younger is in flight then, so the wait is vmcnt(0), one
    //     barrier interval after the issue; the reads follow in intervals 2T + 2 (g = 0) and 2T + 3 (g = 1).
    bf16x8 Af[8][2], Bf[4][2];
    auto read_all = [&](int buf) {
        const char* pa = smem + buf * TILE + a_base;
        const char* pb = smem + buf * TILE + b_base;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            Bf[nb][0] = *reinterpret_cast<const bf16x8*>(pb + nb * 16 * 128 + sw0);
            Bf[nb][1] = *reinterpret_cast<const bf16x8*>(pb + nb * 16 * 128 + sw1);
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            Af[mb][0] = *reinterpret_cast<const bf16x8*>(pa + mb * 16 * 128 + sw0);
            Af[mb][1] = *reinterpret_cast<const bf16x8*>(pa + mb * 16 * 128 + sw1);
        }
    };
    auto mma_all = [&]() {
#if SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb >> 2][nb >> 1][mb & 3][nb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[nb][kk], Af[mb][kk], acc[mb >> 2][nb >> 1][mb & 3][nb & 1], 0, 0, 0);
#if SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto stage_tile = [&](int U) { stage(JC(0), U); stage(JC(1), U); stage(JC(2), U); stage(JC(3), U); };
    const bool g1 = wr == 1;
    // (the generic prologue above staged K tile 0 and three items of K tile 1 and read phase-0 fragments this form does not use:
    //  re-stage K tile 1 whole - same data into the same, unread buffer - and drain)
    stage_tile(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // interval "-1" for group 1 (it starts one interval late); K tiles 0 and 1 are resident, so the first stage is K tile 2 in interval 2
    if (g1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int T = 0; T < nkt; ++T) {
        const int buf = T & 1;
        // ---- L(T) ----
        if (!g1 && T >= 1) stage_tile(T + 1);              // group 0: interval 2T (K tile T - 1's buffer is free since interval 2T)
        read_all(buf);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (g1 && T >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // group 1: end of interval 2T + 1: K tile T + 1 has landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M(T) ----
        if (g1) stage_tile(T + 2);                          // group 1: interval 2T + 2 = 2(T + 1): its share of K tile T + 2
        __builtin_amdgcn_sched_barrier(0);
        mma_all();
        if (!g1 && T >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group 0: end of interval 2T + 1
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    } else if constexpr (STAGGER == 0) {
    for (int T = 0; T < nkt; T += 2) {
        // ===== even K tile T (LDS buffer 0): (0,0) (0,1) (1,1) (1,0) =====
        PHASE_HEAD(0)
        stage(JC(3), T + 1);
        read_b(B1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ax, B0, 0, 0);
        PHASE_HEAD(0)
        stage(JC(0), T + 2);
        read_a(Ay, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ax, B1, 0, 1);
        PHASE_HEAD(1)   // K tile T+1's A halves (staged 3-4 phases ago) must have landed; younger: B1h of T+1, B0h of T+2
        stage(JC(1), T + 2);
        read_a(Ax, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ay, B1, 1, 1);
        PHASE_HEAD(1)   // ... and its B halves (younger: B0h, A0h of T+2)
        stage(JC(2), T + 2);
        read_b(B1, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ay, B0, 1, 0);
        // ===== odd K tile T+1 (LDS buffer 1): (0,1) (0,0) (1,0) (1,1) =====
        PHASE_HEAD(0)
        stage(JC(3), T + 2);
        read_b(B0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ax, B1, 0, 1);
        PHASE_HEAD(0)
        stage(JC(0), T + 3);
        read_a(Ay, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ax, B0, 0, 0);
        PHASE_HEAD(1)
        stage(JC(1), T + 3);
        read_a(Ax, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ay, B0, 1, 0);
        PHASE_HEAD(1)
        stage(JC(2), T + 3);
        read_b(B0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Ay, B1, 1, 1);
    }
    } else {
    // ---- STAGGERED form: every phase is two barrier intervals, L (stage + the reads of the next phase) and M (the 16 MFMAs), and
    // the wave group wr = 1 runs ONE INTERVAL behind the group wr = 0 - in every interval one wave of each SIMD is in its MFMA
    // cluster while the other issues its DMA and LDS reads.  Intervals: group g runs L(phi) in interval 2 phi + g and M(phi) in
    // 2 phi + g + 1.  Ordering rules (RAW: issuing wave's counted vmcnt, then a barrier the reader has passed; WAR: the readers'
    // lgkmcnt(0), then a barrier the stager has passed):
    //   * reads of K tile U in phase phi_r run in intervals 2 phi_r (g = 0) and 2 phi_r + 1 (g = 1): EVERY wave does its vmcnt wait
    //     at the end of interval 2 phi_r - 1 - for g = 0 that is the end of M(phi_r - 1), for g = 1 the end of L(phi_r - 1); both
    //     groups have then issued the items up to phi_r + 6, two of them younger than the ones needed: vmcnt(4) in both;
    //   * lgkmcnt(0) closes every L, so a half-tile's last reads (phase phi_l) have returned by the end of interval 2 phi_l + 1 and
    //     the stage of phase phi_l + 1 (intervals 2 phi_l + 2 and + 3) may overwrite it.
    // (one copy of the loop for both groups - the group only decides, by scalar branches, where the counted wait and the one
    // extra barrier sit; two specialised copies made the allocator spill at their join)
    const bool g1 = wr == 1;
#define L_END(NEEDS_NEXT)                                                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
    if ((NEEDS_NEXT) && g1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                                                  \
    __builtin_amdgcn_sched_barrier(0);
#define M_END(NEEDS_NEXT)                                                          \
    if ((NEEDS_NEXT) && !g1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                                                  \
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (g1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int T = 0; T < nkt; T += 2) {
        // even K tile T (LDS buffer 0): (0,0) (0,1) (1,1) (1,0); phases 2, 3 (pos) read K tile T+1
        stage(JC(3), T + 1); read_b(B1, 0, 1); L_END(0)  mma(Ax, B0, 0, 0); M_END(0)
        stage(JC(0), T + 2); read_a(Ay, 0, 1); L_END(1)  mma(Ax, B1, 0, 1); M_END(1)
        stage(JC(1), T + 2); read_a(Ax, 1, 0); L_END(1)  mma(Ay, B1, 1, 1); M_END(1)
        stage(JC(2), T + 2); read_b(B1, 1, 1); L_END(0)  mma(Ay, B0, 1, 0); M_END(0)
        // odd K tile T+1 (LDS buffer 1): (0,1) (0,0) (1,0) (1,1); phases 6, 7 read K tile T+2
        stage(JC(3), T + 2); read_b(B0, 1, 0); L_END(0)  mma(Ax, B1, 0, 1); M_END(0)
        stage(JC(0), T + 3); read_a(Ay, 1, 1); L_END(1)  mma(Ax, B0, 0, 0); M_END(1)
        stage(JC(1), T + 3); read_a(Ax, 0, 0); L_END(1)  mma(Ay, B0, 1, 0); M_END(1)
        stage(JC(2), T + 3); read_b(B0, 0, 0); L_END(0)  mma(Ay, B1, 1, 1); M_END(0)
    }
    if (!g1) __builtin_amdgcn_s_barrier();  // (the group behind has one interval left)
    }
    // ---- epilogue: lane holds C[m = block row fr][n = 4 fg + r] (weights were the first MFMA operand)
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int m = m0 + wr * 128 + mh * 64 + mb * 16 + fr;
                    const int n = n0 + wc * 64 + nh * 32 + nb * 16 + fg * 4;
                    const f32x4 v = acc[mh][nh][mb][nb];
                    *reinterpret_cast<bf16x4*>(C + (size_t)m * N + n) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                }
}

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int zero = argc > 5 ? atoi(argv[5]) : 0;  // 1: zero-filled operands (the guide's other figure)
    const int stagger = argc > 6 ? atoi(argv[6]) : 2;  // 0: all waves in step; 1: 8-phase, groups one interval apart; 2: one L + one M interval per K tile (default)
    auto kern = stagger == 2 ? gemm8p_kernel<2> : (stagger ? gemm8p_kernel<1> : gemm8p_kernel<0>);
    if (M % BM || N % BN || K % (2 * BK) || K < 4 * BK) { fprintf(stderr, "M, N multiples of 256; K a multiple of 128, >= 256\n"); return 2; }
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };  // uniform [-1, 1)
    for (auto& v : hA) v = zero ? 0 : f2bf(rnd());
    for (auto& v : hB) v = zero ? 0 : f2bf(rnd());
    bf16 *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 2)); CHECK(hipMalloc(&dB, hB.size() * 2)); CHECK(hipMalloc(&dC, hC.size() * 2));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const dim3 grid((M / BM) * (N / BN)), block(512);
    hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, 0, dA, dB, dC, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    // check: 64 random rows x all columns against a double reference (asymmetric operands: a transposed map would show)
    double num = 0, den = 0, worst = 0;
    for (int t = 0; t < 64 && !zero; ++t) {
        const int m = (int)(((unsigned long long)t * 2654435761ull + 17) % M);
        for (int n = 0; n < N; n += 7) {
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
            const double d = bf2f(hC[(size_t)m * N + n]) - r;
            num += d * d; den += r * r;
            worst = fmax(worst, fabs(d));
        }
    }
    const double relerr = zero ? 0.0 : sqrt(num / den);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f, tot = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, 0, dA, dB, dC, M, N, K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters; tot += ms; if (ms < best) best = ms;
    }
    const double fl = 2.0 * M * N * (double)K;
    printf("{\"kernel\": \"gemm8p 256x256x64 8-wave 8-phase\", \"M\": %d, \"N\": %d, \"K\": %d, \"operands\": \"%s\", \"setprio\": %d, \"swz16\": %d, \"stagger\": %d, \"rel_err\": %.3e, \"max_abs_err\": %.3e, "
           "\"us_best\": %.2f, \"us_mean\": %.2f, \"tflops_best\": %.1f, \"tflops_mean\": %.1f, \"frac_of_2p5pf\": %.3f}\n",
           M, N, K, zero ? "zero" : "uniform[-1,1)", SETPRIO, SWZ16, stagger, relerr, worst, best * 1e3, tot / 5 * 1e3, fl / best / 1e9, fl / (tot / 5) / 1e9, fl / (tot / 5) / 1e9 / 2500.0);
    return relerr < 1e-2 ? 0 : 1;
}
