/* mrisr_debug.h - test hooks and tuning tools exported by libmrisr.so.
 *
 * NOT part of the drop-in surface (include/mrisr.h): these are process-global switches used by the parity tests
 * (tests/test_gpu_*.py force split-K, a tile, an older code path, so that every kernel variant is exercised against
 * the oracle) and by the measurement tools under tools/ (tile sweeps, autotuner statistics).  A product caller never
 * needs them.  All are no-ops for correctness: every setting selects between implementations of the same arithmetic.
 */
#ifndef MRISR_DEBUG_H
#define MRISR_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* GEMM planner overrides (0 restores the planner): a split-K factor for every GEMM that can be split; one kernel
 * configuration for every GEMM; a specialised kernel (halo conv 41-45, weight-stationary 50-52, row-panel 60+) wherever
 * it is eligible. */
void mrisr_debug_force_split(int splitk);
void mrisr_debug_force_tile(int tile);
void mrisr_debug_prefer_tile(int tile);
/* bit flags: 8 = rank-4 LoRA up-projection in the scalar epilogue instead of on the matrix cores; 16 = head-major outputs
 * stored straight from the accumulator layout instead of through LDS; 2048 = every GEMM workgroup fills its LDS allocation with
 * 0xFF bytes (NaN) before its first LDS-DMA, so that a read that runs ahead of its DMA cannot return plausible stale data */
void mrisr_debug_gemm_flags(int flags);
/* cost-model efficiency of one tile id (only used when the autotuner is off) */
void mrisr_debug_set_tile_eff(int tile, double eff);
/* override one entry of the in-process tile table (key as in the MRISR_TUNE_CACHE file) */
void mrisr_debug_set_tuned(const char* key, int tile, int splitk);
/* 0: two-kernel GroupNorm everywhere; 1: the one-pass kernel where the geometry allows (default) */
void mrisr_debug_gn_fused(int on);

/* plan-time autotuner: number of signatures tuned and the time spent; free its scratch operands */
int mrisr_autotune_stats(int* shapes, double* ms);
void mrisr_autotune_release(void);

/* micro-benchmark of one GEMM / conv signature on random operands (tools/gemm_sweep.py): average ms over `iters` */
int mrisr_bench_gemm(int M, int N, int K, int conv, int B, int H, int W, int stride, int ups, int c1, int tile, int splitk,
                     int iters, float* ms_out);

/* split-K: 1 the last split of a tile to arrive reduces inside the GEMM kernel, 0 the separate reduce kernel,
 * -1 (default) as MRISR_SK_INKERNEL says (unset: the reduce kernel - measured no slower, see gemm.hip sk_counters_for) */
void mrisr_debug_sk_inkernel(int on);

/* the sub-pixel form of the decoder's `nearest x2 -> conv3x3` layers (bf16 inference): -1 default (MRISR_SUBPIX, 2048 low-resolution
 * rows), 0 off (the literal up-sampled conv), n > 0: from n rows on */
void mrisr_debug_subpix(int min_rows);

/* fused sampler: time embeddings of all steps of a run computed once (a table, one row copied per step) instead of sinusoid + three GEMVs in
 * every step: -1 default (MRISR_TEMB_TABLE, on), 0 per step, 1 on */
void mrisr_debug_temb_table(int on);

/* the transformer's proj_out + outer residual as a continuation of the fused feed-forward kernel (C = 320): -1 default (MRISR_MLP_PROJ, on),
 * 0 its own launch, 1 on */
void mrisr_debug_mlp_proj(int on);

/* split-K conv whose only consumer is a GroupNorm: -1 default (MRISR_GN_SLABS, on) the GroupNorm kernel sums the f32 slabs itself
 * (one launch instead of splitk_reduce + GroupNorm), 0 the two launches, 1 on */
void mrisr_debug_gn_slabs(int on);

/* the fused row-local middle of the C = 320 transformer blocks (csrc/xtail.hip: attn1.to_out + residual, LayerNorm2, attn2.to_q,
 * cross-attention, attn2.to_out + residual in one launch): -1 default (MRISR_XTAIL, on), 0 the four separate launches, 1 on */
void mrisr_debug_xattn_tail(int on);
/* probe: device buffer of (workgroups x 40) u64 that every launch of that kernel fills with clock stamps at its phase boundaries
 * (tools/probes/xtail_stamps.py); NULL (default) = off */
void mrisr_debug_xattn_tail_stamps(void* dev_buf);

/* micro-benchmark of the fused feed-forward kernel (tools/mlp_probe.py): M rows of width 320, random operands */
int mrisr_bench_mlp(int M, int hidden, int iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MRISR_DEBUG_H */
