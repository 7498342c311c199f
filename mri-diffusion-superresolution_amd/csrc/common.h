// mrisr - MI355X (gfx950) kernels for the diffusion super-resolution denoiser hot path.
// Shared declarations for the kernel translation units and the model runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace mrisr {

// ---------------------------------------------------------------------------------------------
// element types.  Activations/weights are stored either as bf16 ("fast") or f32 ("parity").
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2, DT_I64 = 3 };

template <typename T> struct TypeTag;
template <> struct TypeTag<float> { static constexpr int id = DT_F32; };
template <> struct TypeTag<bf16> { static constexpr int id = DT_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 + the rcp / exp2 approximations, ~1e-6): one v_rcp, one v_exp and
// eight FMAs instead of the ~40-instruction libm erff.  Used where the result is rounded to bf16 anyway (the GEGLU
// epilogue evaluates 42 M of these per feed-forward projection at the bench geometry: it was VALU-bound on erff).
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-p, e, 1.0f), x);
}
// the same two at a time on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: the polynomial and the products at half the issue slots; rcp / exp2
// stay per element), operation for operation the scalar erf_fast / gelu_erf_t<bf16> - same bits
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 erf_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 d = ax * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    p = p * t;
    const f32x2 a2 = (ax * (-1.4426950408889634f)) * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
    const f32x2 r = 1.0f - p * e;
    return f32x2{copysignf(r[0], x[0]), copysignf(r[1], x[1])};
}
__device__ __forceinline__ f32x2 gelu_erf2_bf16(f32x2 x) { return (x * 0.5f) * (erf_fast2(x * 0.70710678118654752f) + 1.0f); }
template <typename T> __device__ __forceinline__ float gelu_erf_t(float x);
template <> __device__ __forceinline__ float gelu_erf_t<float>(float x) { return gelu_erf_f(x); }
template <> __device__ __forceinline__ float gelu_erf_t<bf16>(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }

// ---------------------------------------------------------------------------------------------
// error plumbing: kernels' launchers return hipError_t-like ints; C-ABI turns them into messages.
// ---------------------------------------------------------------------------------------------
void set_error(const std::string& msg);
#define MRISR_CHECK_HIP(expr)                                                                      \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::mrisr::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)
#define MRISR_REQUIRE(cond, msg)                                                                   \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::mrisr::set_error(std::string("requirement failed: ") + #cond + " - " + (msg));       \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)

// ---------------------------------------------------------------------------------------------
// implicit GEMM  D[m][n] = sum_k A[m][k] * W[n][k]     (gemm.hip)
//   A rows are either plain rows of up to two row-major matrices concatenated along k
//   (skip-concat, LoRA rank tail) or the 3x3 im2col of an NHWC image (optionally nearest-x2
//   up-sampled, stride 1/2, zero padded).  Both operands are K-contiguous.
// ---------------------------------------------------------------------------------------------
enum GemmAct : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GEGLU = 3 };
enum GemmOut : int { OUT_ROWS = 0, OUT_HEADS = 1, OUT_F32 = 2, OUT_NONE = 3 /* micro-benchmark only: epilogue math, no store */ };

struct GemmArgs {
    // ---- A operand ----
    const void* a0 = nullptr;  // [rows][lda0] elements of T
    const void* a1 = nullptr;  // second source (channels c0..c0+c1), may be null
    int c0 = 0, c1 = 0;        // channels (= k extent per tap) of each source
    int lda0 = 0, lda1 = 0;    // row pitch in elements
    int conv = 0;              // 0: plain rows, 1: 3x3 conv (pad 1)
    int B = 0, Hin = 0, Win = 0, Hout = 0, Wout = 0, stride = 1, ups = 0;
    int pad = 1;               // zero padding on the low (top / left) side; the high side is whatever Hout / Wout imply
                               // (pad = 0 with Hout = Hin / 2: diffusers' Downsample2D(padding=0), asymmetric (0,1,0,1))
    int zstuff = 0;            // with ups = 1: the x2 image is ZERO-STUFFED (odd rows/cols are 0) instead of nearest -
                               // the input of a stride-2 conv's dgrad (transposed convolution)
    // ---- W operand: [N][K] elements of T, K = taps*(c0+c1) ----
    const void* w = nullptr;
    int M = 0, N = 0, K = 0;
    // ---- batching (blockIdx.z): element offsets ----
    // sub-pixel form of `nearest x2 -> 3x3 conv` (runner.h::upsample_conv): kw = 2 taps per kernel row (K = kw * kw * C), and with
    // subpix the batch index z is the output parity (py, px) = (z >> 1, z & 1): the 2 x 2 window starts at (oy - pad + py, ox - pad + px)
    int kw = 3, subpix = 0;
    int batch = 1;
    long long a_bs = 0, w_bs = 0;
    int heads = 1;             // output offset = (z / heads) * o_bs + (z % heads) * o_hs
    long long o_bs = 0, o_hs = 0;
    // ---- split-K (blockIdx.y); partials go to `partial` as f32 [split][M][N] ----
    int splitk = 1;
    float* partial = nullptr;
    // per-tile arrival counters (zero between launches): the last split of a tile to arrive sums the slabs and runs the kernel's own
    // epilogue, so no separate reduce launch follows (set by launch_gemm for the bf16 DMA kernels; null = splitk_reduce_kernel)
    unsigned* sk_counters = nullptr;
    int defer_reduce = 0;      // split-K: leave the f32 slabs in `partial`, launch no reduce (the caller's next kernel sums them: launch_groupnorm_slabs)
    int pretouch = 0;  // set by launch_gemm (> 0: pieces per wave at most): the workgroups DMA-touch the whole weight matrix at kernel start (cold weights, see gemm.hip)
    // ---- epilogue ----
    float alpha = 1.0f;
    const float* bias = nullptr;    // [N] (GEGLU: interleaved like the weight rows)
    const float* rowvec = nullptr;  // + rowvec[(m / rowvec_div) * rowvec_ld + n]   (time-embedding projection)
    int rowvec_div = 1, rowvec_ld = 0;
    int act = ACT_NONE;
    const void* resid = nullptr;    // + resid[m][n] (T), pitch ldr; added after the activation
    int ldr = 0;
    int out_mode = OUT_ROWS;
    void* out = nullptr;            // OUT_ROWS: T [M][ldo]; OUT_F32: float [M][ldo]
    int ldo = 0;
    // OUT_HEADS: column n -> section s = n / secC, head h, dim dd; row m -> (b, tok)
    void* sec_ptr[3] = {nullptr, nullptr, nullptr};
    int sec_tr[3] = {0, 0, 0};      // 1: store transposed [b][h][dd][tok] (V^T), else [b][h][tok][dd]
    int secC = 0, hd = 0, dpad = 0, ntok = 0, npad = 0, nheads = 0;
    // profiler only: algorithmic work of this launch (0 -> derived from M, N, K)
    double alg_flops = 0.0, alg_bytes = 0.0;
    int tile = 0;  // kernel configuration chosen by gemm_choose (0: let launch_gemm plan)
    int stage_out = 1;  // row outputs leave through LDS as whole rows (set by launch_gemm; MRISR_STAGE_OUT=0/1/2)
    // LoRA rank-r update applied in the epilogue: out[m][n] += sum_q z[m][zoff(n)+q] * lb[n][q],
    //   z = x A^T (f32, from launch_lora_down), lb = (alpha/r) * B (f32); zoff(n) = (n / lora_secN) * lora_r
    const float* lora_z = nullptr;
    const float* lora_b = nullptr;
    int lora_r = 0, lora_zld = 0, lora_secN = 1;
    // bf16 buffer-addressed kernels, un-split plain GEMMs: compute z inside the GEMM instead (lora_z unused):
    //   lora_a = A [lora_R <= 16][K] (compute dtype); lora_zout (optional): z is also written there, f32 [M][lora_R]
    const void* lora_a = nullptr;
    int lora_R = 0;
    float* lora_zout = nullptr;
    // row-panel kernel (tiles 60+): optional LayerNorm prologue on the A rows (statistics over K, affine from these vectors)
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    float ln_eps = 1e-5f;
    // fp8 (OCP e4m3) operands of the row-panel kernel: weights [N][K] bytes + one f32 scale per output channel; LoRA A rows
    // [lora_R][K] bytes + one scale per adapter row (packed at finalize when the model's fp8 flag is set)
    const void* w8 = nullptr;
    const float* w_scale = nullptr;
    const void* lora_a8 = nullptr;
    const float* lora_a_scale = nullptr;
    int no_rp = 0;    // the caller's operand forms rule the row-panel kernel out (e.g. LoRA of a rank it does not fuse)
    int group_m = 1;  // M tiles per group of the tile order (set by the launchers: auto_group_m; MRISR_GROUP_M)
    int dbg = 0;  // cross-check switches (mrisr_debug_gemm_flags): 8 scalar LoRA up-projection, 16 unstaged head-major stores
};

template <typename T> int launch_gemm(const GemmArgs& g, hipStream_t st);
template <typename T> int launch_splitk_reduce(const GemmArgs& g, hipStream_t st);
const void* zero_page();  // >= 256 bytes of device zeros, valid after init_zero_page()
int init_zero_page();
int gemm_prepare();
// fused feed-forward of a transformer block (C = 320): out = resid + FF2(GEGLU(FF1(LayerNorm(x)))), hidden activation on-chip
struct MlpArgs {
    const void* x = nullptr; int ldx = 0, M = 0;
    const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 1e-5f;
    const void* w1 = nullptr; const float* b1 = nullptr;    // [2H][C] (u, gate) interleaved, bias alike
    const void* w2p = nullptr; const float* b2 = nullptr;   // [N2][H] packed by launch_pack_mlp_w2
    const void* resid = nullptr; int ldr = 0;
    void* out = nullptr; int ldo = 0;
    int C = 0, H = 0, N2 = 0;
    // optional continuation: out2 = out Wp^T + bp + xres (the transformer's proj_out + outer residual); `out` itself is then not stored
    const void* wp = nullptr; const float* bp = nullptr;   // [N2][N2] packed by launch_pack_mlp_w2
    const void* xres = nullptr; int ldxr = 0;
    void* out2 = nullptr; int ldo2 = 0;
};
bool mlp_fused_ok(int C, int H, int N2);
bool mlp_proj_enabled();  // the transformer's proj_out + outer residual as a continuation of the fused feed-forward kernel (MRISR_MLP_PROJ, default 1)
int launch_pack_mlp_w2(const void* w2_bf16, void* dst, int N2, int H, hipStream_t st);
int launch_mlp_fused(const MlpArgs& m, hipStream_t st);
// the row-local middle of a transformer block at C = 320 in one kernel (xtail.hip): attn1.to_out + residual, LayerNorm2, attn2.to_q,
// cross-attention over the cached prompt K / V, attn2.to_out + residual
struct XTailArgs {
    const void* ao = nullptr; int ldao = 0;   // self-attention output rows [M][320]
    void* t = nullptr; int ldt = 0;           // residual stream rows [M][320]: read as x0, overwritten with x2
    int M = 0, ntok = 0;                      // rows; tokens per image
    const void* w1 = nullptr; const float* b1 = nullptr; const void* a1 = nullptr; const float* lb1 = nullptr;  // attn1.to_out: W [320][320] (natural K), LoRA A [r][320], B f32 [320][4]
    const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 1e-5f;
    const void* wq = nullptr; const float* bq = nullptr; const void* aq = nullptr; const float* lbq = nullptr;  // attn2.to_q: W K-permuted (launch_pack_mlp_w2)
    const void* kvp = nullptr; int nk = 0; float scale = 1.f;                                                  // launch_pack_xattn_kv image; keys; 1 / sqrt(d)
    const void* w2 = nullptr; const float* b2 = nullptr; const void* a2 = nullptr; const float* lb2 = nullptr;  // attn2.to_out: W K-permuted
    int lora_r = 0;                           // 4 (all three projections carry a rank-4 adapter) or 0
};
int xattn_tail_prepare();
bool xattn_tail_enabled();
bool xattn_tail_ok(int C, int heads, int M, int ntok, int nk);
size_t xattn_tail_kv_bytes(int B);
int launch_pack_xattn_kv(const void* kc, const void* vtc, void* dst, int B, int ctx_pad, int dpad, int nk, hipStream_t st);
int launch_xattn_tail(const XTailArgs& x, hipStream_t st);
int gemm_rp_tile(const GemmArgs& g);  // row-panel kernel id for this (plain, short-K, bf16) GEMM, 0 if it is not eligible
int gemm_choose(GemmArgs& g, bool is_bf16);  // sets g.tile / g.splitk (autotuned per signature for bf16)  // set launch attributes of every GEMM instantiation (call before graph capture)

// ---------------------------------------------------------------------------------------------
// normalisation (norm.hip)
// ---------------------------------------------------------------------------------------------
struct GroupNormArgs {
    const void* x0 = nullptr;  // [B][HW][c0] T
    const void* x1 = nullptr;  // [B][HW][c1] T or null (skip-concat)
    int c0 = 0, c1 = 0;
    int B = 0, HW = 0, groups = 32;
    float eps = 1e-5f;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    int silu = 0;
    void* y = nullptr;            // [B][HW][c0+c1] T
    float* partial = nullptr;     // workspace: [B][nsplit][groups][2]
    int nsplit = 1;
};
extern int g_plan_salt;  // model.hip: part of every workspace key; bumped by debug toggles that change what a forward allocates
int groupnorm_nsplit(int B, int HW);
// The epilogue of a split-K conv / linear whose only consumer is a GroupNorm, done by the GroupNorm kernel itself: it sums the f32 slabs
// (+ bias, time-embedding row, residual - the operations of splitk_reduce_kernel in the same order), rounds to bf16 as the reduce would
// have, optionally stores that raw tensor, and normalises from registers.  One launch instead of reduce + GroupNorm.
struct GnSlabSrc {
    const float* partial = nullptr;   // [splitk][M][C] f32
    int splitk = 1;
    float alpha = 1.f;
    const float* bias = nullptr;      // [C] or null
    const float* rowvec = nullptr;    // [(M / rowvec_div)][rowvec_ld] or null
    int rowvec_ld = 0, rowvec_div = 1;
    const void* resid = nullptr;      // [M][ldr] bf16 or null
    int ldr = 0;
    void* raw_out = nullptr;          // [M][C] bf16 or null: the un-normalised tensor, if anything else reads it
};
bool groupnorm_slabs_ok(int C, int groups, int HW);
int launch_groupnorm_slabs(const GroupNormArgs& a, const GnSlabSrc& s, hipStream_t st);
// geometry of the one-pass (image slab in registers) GroupNorm kernels, forward and backward; false: two-kernel path
bool gn_fused_geometry(int c0, int c1, int groups, int HW, int* slab, int* slots, int* rl, int* nv, int ve = 8);
template <typename T> int launch_groupnorm(const GroupNormArgs& a, hipStream_t st);
template <typename T>
int launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, int M, int C, float eps,
                     hipStream_t st);
// in-place row softmax over the first nk columns of f32 rows of pitch ld; writes P as T into `p` (pitch ldp),
// zero-filling columns nk..ldp-1
template <typename T>
int launch_softmax_rows(const float* s, int ld, void* p, int ldp, long long rows, int nk, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// backward-pass kernels (bwd.hip): LoRA fine-tuning step
// ---------------------------------------------------------------------------------------------
struct GroupNormBwdArgs {
    const void* x0 = nullptr;   // forward inputs (as GroupNormArgs)
    const void* x1 = nullptr;
    int c0 = 0, c1 = 0, B = 0, HW = 0, groups = 32;
    float eps = 1e-5f;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    int silu = 0;
    const void* dy = nullptr;   // [B][HW][c0+c1] T
    void* dx0 = nullptr;        // [B][HW][c0] T
    void* dx1 = nullptr;        // [B][HW][c1] T
    int acc0 = 0, acc1 = 0;     // accumulate into dx0 / dx1 instead of overwriting
    const float* fwd_partial = nullptr;  // the forward's [B][nsplit][groups][2] sums (mean / rstd are re-derived)
    float* bwd_partial = nullptr;        // workspace of the same size
    int nsplit = 1;
};
template <typename T> int launch_groupnorm_bwd(const GroupNormBwdArgs& a, hipStream_t st);
template <typename T>
int launch_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, int M, int C, float eps, int accumulate,
                         hipStream_t st);
template <typename T> int launch_geglu_bwd(const void* pre, const void* dout, void* dpre, long long M, int half, hipStream_t st);
template <typename T> int launch_geglu_fwd(const void* pre, void* out, long long M, int half, hipStream_t st);
template <typename T>
int launch_rows_to_heads(const void* x, int ldx, int col_off, void* dst, int B, int N, int H, int hd, int npad, int dpad, hipStream_t st);
template <typename T>
int launch_softmax_bwd(const void* p, const float* dp, void* ds, int ld, long long rows, int nk, float scale, hipStream_t st);
template <typename T>
int launch_transpose(const void* src, void* dst, int R, int C, int ld_src, int ld_dst, long long bs_src, long long bs_dst, int batch,
                     int r_valid, hipStream_t st);
template <typename T>
int launch_lora_wgrad(const void* P, int ldp, const float* Q, int ldq, int M, int C, int mode, int r, int nmod, int secN,
                      float* const out[3], float scale, float* scratch, hipStream_t st);
size_t lora_wgrad_scratch_bytes(int M, int C, int nq, int elem_size);
template <typename T> int launch_sumpool2(const void* src, void* dst, int B, int H, int W, int C, int accumulate, hipStream_t st);
template <typename T>
int launch_mse_grad(const void* pred, const float* tgt, void* dpred, float* loss, int B, int C, int H, int W, hipStream_t st);
template <typename T>
int launch_im2col_tap_T(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, int ky, int kx,
                        int Mpad, hipStream_t st);
template <typename T> int launch_relu_bwd(const void* dy, const void* h, void* out, long long n, hipStream_t st);
template <typename T> int launch_colsum(const void* dy, float* out, int M, int C, hipStream_t st);
int launch_wgrad_accum(const float* tmp, float* gw, long long n, int taps, int tap, hipStream_t st);
int launch_wgrad_accum_all(const float* tmp, float* gw, int Cout, int Cin, int taps, hipStream_t st);
// full-parameter training: tmp rows with pitch ld_tmp, per-tap channel pitch cin_src (zero-padded layers), GEGLU row interleave (half > 0)
int launch_wgrad_accum_gen(const float* tmp, int ld_tmp, int cin_src, float* gw, int Cout, int Cin, int taps, int geglu_half, hipStream_t st);
template <typename T> int launch_colsum_gen(const void* dy, int ld, int col0, float* out, int M, int C, int geglu_half, hipStream_t st);
template <typename T>
int launch_gn_affine_grad(const void* x, const void* dy, const float* gamma, const float* beta, const float* fwd_partial, int nsplit, int groups, int B, int HW,
                          int C, float eps, int silu, float* g_gamma, float* g_beta, hipStream_t st);
template <typename T> int launch_ln_affine_grad(const void* x, const void* dy, int M, int C, float eps, float* g_gamma, float* g_beta, hipStream_t st);
template <typename T> int launch_rowvec_grad(const void* dh, float* out, int ld_out, int off, int B, int HW, int C, int scalar_t, hipStream_t st);
template <typename T> int launch_silu_fwd(const void* x, void* y, long long n, hipStream_t st);
template <typename T> int launch_silu_bwd(const void* dy, const void* pre, void* dx, long long n, hipStream_t st);
template <typename T> int launch_pack_conv_dgrad_padded(const float* w, void* wd, int Cout, int Cin, int Cout_p, int Cin_p, hipStream_t st);
int launch_small_wgrad(const float* dY, int ldy, const float* X, int ldx, int rows, int N, int K, int silu_in, float* gW, float* gB, hipStream_t st);
template <typename T>
int launch_small_dgrad(const float* dY, int ldy, const void* W, int rows, int N, int K, const float* pre, int ldpre, float* dX, int ldx, hipStream_t st);
template <typename T>
int launch_im2col_all_T(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, int ks, int Mpad,
                        hipStream_t st);
template <typename T> int launch_pack_conv_dgrad(const float* w, void* wd, int Cout, int Cin, hipStream_t st);
int launch_sumsq(const float* g, long long n, float* out, hipStream_t st);
int launch_ema(float* ema, const float* theta, long long n, float decay, hipStream_t st);
int launch_adamw(float* p, const float* g, float* m, float* v, long long n, const float* sumsq, float grad_scale, float max_norm,
                 float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// attention (attn.hip): flash-style, head-major operands written by the QKV GEMM epilogue
//   q  [B*H][nq ][dpad]   k [B*H][nkpad][dpad]   vt [B*H][dpad][nkpad]   -> out [B][nq][H*hd]
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const void* q = nullptr;
    const void* k = nullptr;
    const void* vt = nullptr;
    void* out = nullptr;
    int B = 0, H = 0, nq = 0, nk = 0, nkpad = 0, hd = 0, dpad = 0;
    float scale = 1.0f;
    float* lse = nullptr;  // optional [B*H][npad]: log2-domain log-sum-exp of scale*s (training keeps it for the backward)
    int xcd_map = 0;       // set by the launcher: all query blocks of a head on one XCD (needs B*H % 8 == 0)
    // fp8 attention (launch_attention_fp8): caller's scratch for the e4m3 K / V^T (B*H*nkpad*dpad bytes each) and the per-head scales (B*H*4 floats)
    void* k8 = nullptr;
    void* vt8 = nullptr;
    float* f8_scales = nullptr;
};
int launch_attention_bf16(const AttnArgs& a, hipStream_t st);
int launch_attention_fp8(const AttnArgs& a, hipStream_t st);  // same operands (bf16, head-major) + the scratch fields; Q K^T and P V in OCP e4m3
// flash attention backward (attn.hip): P is recomputed from Q, K and the forward's log-sum-exp, never materialised.
//   head-major operands, pads zero:  q, doh [B*H][npad][dpad];  k, v [B*H][nkpad][dpad];  their transposes qt, doht
//   [B*H][dpad][npad], kt [B*H][dpad][nkpad];  lse, dsum [B*H][npad] (dsum = rowsum(dO o O), 0 on pads).
//   Outputs as token rows (head h -> columns h*hd..): dq [B][nq][ldq]; dk, dv [B][nk][ldkv] (both may be null).
struct AttnBwdArgs {
    const void* q = nullptr;
    const void* k = nullptr;
    const void* v = nullptr;
    const void* doh = nullptr;
    const void* qt = nullptr;
    const void* kt = nullptr;
    const void* doht = nullptr;
    const float* lse = nullptr;
    const float* dsum = nullptr;
    void* dq = nullptr;
    void* dk = nullptr;
    void* dv = nullptr;
    int ldq = 0, ldkv = 0;
    int B = 0, H = 0, nq = 0, nk = 0, npad = 0, nkpad = 0, hd = 0, dpad = 0;
    float scale = 1.0f;
};
int launch_attention_bwd_bf16(const AttnBwdArgs& a, hipStream_t st);
// dO / O token rows [B][N][H*hd] -> doh head-major [B*H][npad][dpad] (valid region only) and dsum[b*H+h][q] = sum_d dO*O
int launch_attention_bwd_prep(const void* dO_rows, const void* O_rows, void* doh, float* dsum, int B, int N, int H, int hd, int npad,
                              int dpad, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// small kernels (misc.hip)
// ---------------------------------------------------------------------------------------------
// y[b][n] = act_out( sum_k act_in(x[b][k]) * W[n][k] + bias[n] ),  x,y f32, W of type T.  rows <= 64.
template <typename T>
int launch_gemv_rows(const float* x, int ldx, const void* w, const float* bias, float* y, int ldy, int rows, int N,
                     int K, int silu_in, hipStream_t st, int f32_inputs = 0);  // f32_inputs: never the matrix-core form (which rounds x to bf16)
// LoRA down-projection z[M][R] = x[M][K] . A[R][K]^T  (x, A of type T; z f32): one wave per row, memory-bound
template <typename T>
int launch_lora_down(const void* x, int ldx, const void* A, float* z, int M, int K, int R, hipStream_t st);
// sinusoidal timestep embedding [rows][dim] = [cos | sin], t from device int64 (scalar broadcast or [rows])
int launch_timestep_embedding(const long long* t, int t_is_scalar, float* out, int rows, int dim, hipStream_t st);
// out[j] = table[(*step - first) * n + j], j < n (the fused sampler's per-run time-embedding table)
int launch_select_row(const float* table, const int* step, int first, float* out, int n, hipStream_t st);
// direct NHWC conv for tiny channel counts (conv_in, conv_out, ControlNet condition embedding)
struct DirectConvArgs {
    const void* x = nullptr;   // [B][Hin][Win][Cin] T
    const void* w = nullptr;   // [Cout][ks*ks*Cin] T
    const float* bias = nullptr;
    void* y = nullptr;         // [B][Hout][Wout][Cout] T
    int B = 0, Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, Cout = 0, ks = 3, stride = 1, pad = 1, act = 0;
    const void* add = nullptr; // + add[m][n] (T), same shape as y
};
template <typename T> int launch_direct_conv(const DirectConvArgs& a, hipStream_t st);
int direct_conv_prepare();
// layout / dtype conversion at the boundary.  src dtype is a DType id.
template <typename T>
int launch_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int B, int C, int H, int W, hipStream_t st);
template <typename T>
int launch_nhwc_to_nchw(const void* src, void* dst, int dst_dtype, int B, int C, int H, int W, float scale,
                        hipStream_t st);
template <typename T> int launch_add_inplace(void* x, const void* y, long long n, hipStream_t st);
template <typename T> int launch_scale_inplace(void* p, float s, long long n, hipStream_t st);
template <typename T> int launch_pixel_unshuffle_nchw(const void* src, int src_dtype, void* dst, int B, int C, int H,
                                                      int W, int r, hipStream_t st);
// generic row copy/cast used by the weight packers:
//   dst[(row_map(r))][col_off + c] = scale * src[r][c]      src f32, dst T
//   row_map: 0 identity (+row_off), 1 GEGLU interleave (u/g blocks of 16; src has 2*half rows)
template <typename T>
int launch_pack_rows(const float* src, int rows, int cols, void* dst, int ld_dst, int row_off, int col_off,
                     int row_map, int half, float scale, hipStream_t st);
// conv weight [Cout][Cin][3][3] f32 -> [Cout][ky][kx][Cin] T
template <typename T>
int launch_pack_conv3x3(const float* src, void* dst, int Cout, int Cin, int ks, hipStream_t st);
// `nearest x2 -> conv3x3` as four 2 x 2 convs on the low-resolution input (one per output parity): dst [4][Cout][2][2][Cin], every tap
// the f32 sum of the 3 x 3 taps that read the same low-resolution pixel; and the pass that interleaves the four parity planes
template <typename T>
int launch_pack_conv_subpix(const float* src, void* dst, int Cout, int Cin, hipStream_t st);
template <typename T>
int launch_subpix_shuffle(const void* planes, void* out, int B, int H, int W, int C, hipStream_t st);
template <typename T>
int launch_pack_conv3x3_padded(const float* src, void* dst, int Cout, int Cin, int ks, int Cout_pad, int Cin_pad, hipStream_t st);
int launch_pack_bias_geglu(const float* src, float* dst, int half, hipStream_t st);
template <typename T> int launch_fill_zero(void* p, long long n, hipStream_t st);
// packed bf16 rows -> OCP e4m3 rows + one f32 scale per row (amax / 448)
int launch_quant_rows_fp8(const void* src_bf16, int rows, int cols, void* dst8, float* scales, hipStream_t st);
// sampler steps (f32 state, NCHW like the reference's latents)
//   ddim: x = cx*x + ce*eps;   coefficients read from a device table indexed by *step_idx
int launch_ddim_step(float* x, const float* eps, const float* coef_table, const int* step_idx, long long n,
                     hipStream_t st);
//   res-srdiff reverse step (reference res_srdiff.py:84-96): coef row = {sqrt_at, sqrt_1mat, sqrt_ap, sigma}
int launch_ddpm_step(float* x, const float* eps, const float* noise, const float* coef_table, const int* step_idx, float clip,
                     long long n, hipStream_t st);
int launch_resshift_step(float* x, const float* eps, const float* lr, const float* noise, const float* coef_table,
                         const int* step_idx, long long n, hipStream_t st);
int launch_advance_step(int* step_idx, hipStream_t st);
//   forward shift (reference res_srdiff.py:7-25): per-sample alpha from table[t[b]]
int launch_resshift_forward(const float* hr, const float* lr, const float* noise, const float* alphas_cumprod,
                            const long long* t, int t_is_scalar, float* out, int B, long long per_sample,
                            hipStream_t st);

}  // namespace mrisr
