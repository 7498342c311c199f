/* mrisr.h - C ABI of libmrisr.so: the MI355X (gfx950) denoiser hot path of
 * Bernat-C/MRI-Diffusion-SuperResolution, behind the call surface the reference uses.
 *
 * What each entry point replaces in the reference (file:line under /root/reference):
 *   mrisr_unet_forward        <- unet(latents, t, encoder_hidden_states=..., down_block_additional_residuals=...,
 *                                     mid_block_additional_residual=...).sample      src/adapters/res_srdiff.py:73-78
 *                                (diffusers UNet2DConditionModel; + down_intrablock_additional_residuals for T2I-Adapter)
 *   mrisr_controlnet_forward  <- controlnet(latents, t, encoder_hidden_states=..., controlnet_cond=...,
 *                                     return_dict=False)                             src/adapters/res_srdiff.py:65-70
 *   mrisr_adapter_forward     <- Adapter_XL.forward                                  src/adapters/modules.py:146-157
 *   mrisr_resshift_forward    <- get_res_shifting_latents                            src/adapters/res_srdiff.py:7-25
 *   mrisr_sampler_*           <- the timestep loop of log_validation                 src/adapters/res_srdiff.py:63-96
 *                                (Res-SRDiff reverse step :84-96, or the DDIM(eta=0) step BASELINE.json names)
 *   mrisr_*_set_param         <- nn.Module.load_state_dict with diffusers / peft / reference key names
 *
 * Conventions
 *   - every function returns 0 on success; otherwise mrisr_last_error() (thread-local) describes the failure.
 *   - all tensors are caller-owned DEVICE memory described by mrisr_tensor (contiguous; NCHW like torch unless
 *     layout says NHWC).  Kernels are enqueued on the caller's stream; nothing synchronises internally, and
 *     nothing allocates inside *_forward after the first call for a given shape (hipGraph-capturable).
 *   - a handle is bound to the device that was current at creation; handles are not thread-safe.
 *   - no torch / C++ types cross this boundary.
 */
#ifndef MRISR_H
#define MRISR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MRISR_F32 = 0, MRISR_BF16 = 1, MRISR_F16 = 2, MRISR_I64 = 3 } mrisr_dtype;
typedef enum { MRISR_NCHW = 0, MRISR_NHWC = 1 } mrisr_layout;

typedef struct {
    void* data;        /* device pointer (host pointer only where a function says so) */
    int32_t dtype;     /* mrisr_dtype */
    int32_t layout;    /* mrisr_layout; ignored for ndim < 4 */
    int32_t ndim;
    int64_t shape[4];  /* logical NCHW order for 4-D image tensors: {B, C, H, W} */
} mrisr_tensor;

/* Mirror of the diffusers config keys the path depends on (SD-1.5 values in comments). */
typedef struct {
    int32_t in_channels;            /* 4 */
    int32_t out_channels;           /* 4 */
    int32_t num_levels;             /* 4 */
    int32_t block_out_channels[4];  /* 320, 640, 1280, 1280 */
    int32_t attn_levels[4];         /* 1, 1, 1, 0  (CrossAttn{Down,Up}Block2D at that level) */
    int32_t layers_per_block;       /* 2 */
    int32_t num_heads;              /* 8 (diffusers "attention_head_dim": 8) */
    int32_t cross_attention_dim;    /* 768 */
    int32_t norm_num_groups;        /* 32 */
    float norm_eps;                 /* 1e-5 */
    int32_t cond_channels;          /* ControlNet: 3 */
    int32_t cond_embed_channels[4]; /* ControlNet: 16, 32, 96, 256 */
    int32_t compute_dtype;          /* MRISR_BF16 (fast: bf16 storage, f32 accumulate/statistics) or
                                       MRISR_F32 (parity: f32 storage, exact-f32 MFMA) */
    int32_t lora_rank;              /* 0 = no LoRA; else rank of the peft adapters that will be set */
    int32_t lora_fused;             /* 1: rank-r tail fused into the projection GEMMs; 0: merged into W at finalize */
    int32_t flash_attention;        /* 1: fused flash kernel (bf16 only); 0: materialised scores */
    int32_t fp8_linears;            /* 0: off; 1: K = 320 projections; 2: K = 320 and 640 (bf16 models): these projections of the transformer blocks - incl. the LoRA
                                       targets to_q/k/v, to_out - run with OCP e4m3 operands on the fp8 MFMA (per-output-channel
                                       weight scales, per-row activation scales computed in-kernel), f32 accumulate; BASELINE configs[4] */
    int32_t fp8_attention;          /* 1 (bf16 models, flash_attention): Q K^T and P V of every attention on the fp8 MFMA (OCP e4m3 operands,
                                       per-head scales, f32 softmax and accumulate); the log-sum-exp kept for the backward is unchanged */
    int32_t fp8_train;              /* 1: mrisr_train_step runs its FORWARD with the fp8 projections / fp8 attention selected above
                                       and its backward in bf16 with f32 accumulation, straight through the quantisers ("mixed-precision
                                       training" of BASELINE configs[4]); 0: the training forward stays bf16 whatever the inference mode is */
} mrisr_unet_cfg;

typedef struct mrisr_model mrisr_model; /* UNet2DConditionModel or ControlNetModel */
typedef struct mrisr_adapter mrisr_adapter;
typedef struct mrisr_sampler mrisr_sampler;

const char* mrisr_last_error(void);
const char* mrisr_version(void);

/* ---- model lifetime & weights ---------------------------------------------------------------- */
int mrisr_unet_create(const mrisr_unet_cfg* cfg, mrisr_model** out);
int mrisr_controlnet_create(const mrisr_unet_cfg* cfg, mrisr_model** out);
void mrisr_model_destroy(mrisr_model* m);
/* key: diffusers state-dict name, peft LoRA name (<module>.lora_A.default.weight /
 * lora_B.default.weight; <module>.base_layer.weight accepted for wrapped layers).  data: f32, host or device
 * (is_device), contiguous, torch shape.  Copied; the caller may free it afterwards. */
int mrisr_model_set_param(mrisr_model* m, const char* key, const float* data, const int64_t* shape, int ndim,
                          int is_device);
int mrisr_model_set_lora_scale(mrisr_model* m, float scale); /* lora_alpha / r */
/* Re-lays weights out for the kernels (NHWC filter order, fused QKV, GEGLU interleave, LoRA tails, ...).
 * Must be called after the last set_param and before forward; may be called again after updating params. */
int mrisr_model_finalize(mrisr_model* m, void* stream);
int64_t mrisr_model_num_params(const mrisr_model* m);
int64_t mrisr_model_workspace_bytes(const mrisr_model* m);

/* ---- UNet2DConditionModel.forward ------------------------------------------------------------- */
/* sample [B,Cin,h,w]; timestep: device int64, ndim 0 (broadcast) or [B]; ehs [B,L,Dctx] or NULL to reuse the
 * projections cached by the previous call / mrisr_model_set_context; down_res: 0 or 12(=n_skips) tensors shaped like
 * the skips; mid_res: NULL or [B,C_mid,h/8,w/8]; intrablock: 0 or num_levels tensors (T2I-Adapter features);
 * out [B,Cout,h,w] in out->dtype. */
int mrisr_unet_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep,
                       const mrisr_tensor* ehs, const mrisr_tensor* down_res, int n_down_res,
                       const mrisr_tensor* mid_res, const mrisr_tensor* intrablock, int n_intrablock,
                       mrisr_tensor* out, void* stream);
/* Pre-compute the cross-attention K/V projections of a fixed prompt embedding (timestep-invariant). */
int mrisr_model_set_context(mrisr_model* m, const mrisr_tensor* ehs, int latent_h, int latent_w, void* stream);
int mrisr_model_num_skips(const mrisr_model* m);
/* shape {B,C,H,W} of skip k for a latent of h x w (k == num_skips -> the mid block tensor) */
int mrisr_model_skip_shape(const mrisr_model* m, int k, int B, int h, int w, int64_t shape[4]);

/* ---- ControlNetModel.forward ------------------------------------------------------------------- */
/* cond [B,3,8h,8w] or NULL to reuse the cached condition embedding; down_out: n_skips tensors, mid_out: 1. */
int mrisr_controlnet_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep,
                             const mrisr_tensor* ehs, const mrisr_tensor* cond, float conditioning_scale,
                             mrisr_tensor* down_out, int n_down_out, mrisr_tensor* mid_out, void* stream);
int mrisr_controlnet_set_cond(mrisr_model* m, const mrisr_tensor* cond, void* stream);

/* ---- T2I-Adapter (Adapter_XL, sk=True) ----------------------------------------------------------- */
typedef struct {
    int32_t channels[4]; /* 320, 640, 1280, 1280 */
    int32_t nums_rb;     /* 3 */
    int32_t cin;         /* 192 = 3*8*8 */
    int32_t ksize;       /* 3 (1 also supported) */
    int32_t use_conv;    /* 1: stride-2 conv downsample */
    int32_t compute_dtype;
} mrisr_adapter_cfg;
int mrisr_adapter_create(const mrisr_adapter_cfg* cfg, mrisr_adapter** out);
void mrisr_adapter_destroy(mrisr_adapter* a);
int mrisr_adapter_set_param(mrisr_adapter* a, const char* key, const float* data, const int64_t* shape, int ndim,
                            int is_device);
int mrisr_adapter_finalize(mrisr_adapter* a, void* stream);
/* x [B,3,8h,8w] -> 4 feature maps (caller-allocated, any dtype/layout) */
int mrisr_adapter_forward(mrisr_adapter* a, const mrisr_tensor* x, mrisr_tensor* feats, int n_feats, void* stream);

/* ---- scheduler math (f32 latents, elementwise) --------------------------------------------------- */
/* x_t = sqrt(a_t) HR + (1-sqrt(a_t)) LR + sqrt(1-a_t) eps, a_t = alphas_cumprod[t]; t: device int64 0-dim or [B] */
int mrisr_resshift_forward(const mrisr_tensor* hr, const mrisr_tensor* lr, const mrisr_tensor* noise,
                           const float* alphas_cumprod_dev, const mrisr_tensor* timestep, mrisr_tensor* out,
                           void* stream);

/* ---- sampler: the timestep loop, one hipGraph per step -------------------------------------------- */
/* MRISR_STEP_DDPM: the ancestral step of diffusers' DDPMScheduler.step ("fixed_small" variance; BASELINE config 1, "10-step
 * DDPM"): t_prev = t - T/n, alpha_t = abar_t / abar_prev, x0 = (x - sqrt(1-abar_t) eps) / sqrt(abar_t) [clipped to
 * +-clip_sample_range when set], x_prev = sqrt(abar_prev) (1-alpha_t)/(1-abar_t) x0 + sqrt(alpha_t) (1-abar_prev)/(1-abar_t) x
 * + sqrt((1-abar_prev)/(1-abar_t) (1-alpha_t)) z for t > 0;  z = step_noise slab i (NULL: the mean only). */
typedef enum { MRISR_STEP_DDIM = 0, MRISR_STEP_RESSHIFT = 1, MRISR_STEP_DDPM = 2 } mrisr_step_kind;
/* timesteps: host int64[n_steps]; alphas_cumprod: host f32[n_train]; unet required, controlnet may be NULL. */
int mrisr_sampler_create(mrisr_model* unet, mrisr_model* controlnet, int step_kind, const int64_t* timesteps,
                         int n_steps, const float* alphas_cumprod, int n_train, mrisr_sampler** out);
void mrisr_sampler_destroy(mrisr_sampler* s);
/* Runs all steps on `stream`, updating latents [B,C,h,w] f32 NCHW in place.
 *   lr_latents: RESSHIFT anchor (NULL for DDIM);  step_noise: RESSHIFT [n_steps-1 or more][B,C,h,w] f32 or NULL (then
 *   the stochastic term is dropped);  ehs / cond / intrablock as in the forward calls (projected once, before step 0).
 *   use_graph: capture step 0's launches into a hipGraph and replay it for the remaining steps. */
int mrisr_sampler_run(mrisr_sampler* s, mrisr_tensor* latents, const mrisr_tensor* lr_latents,
                      const mrisr_tensor* step_noise, const mrisr_tensor* ehs, const mrisr_tensor* cond,
                      const mrisr_tensor* intrablock, int n_intrablock, int use_graph, void* stream);

/* Restrict the next runs to steps [first_step, last_step) of the schedule (default: all).  Step i always uses the
 * schedule's own (t_i, t_{i+1}) pair, so a truncated run reproduces the prefix of the full trajectory. */
int mrisr_sampler_set_range(mrisr_sampler* s, int first_step, int last_step);
/* DDPM only: clip the predicted x0 to [-range, range] (diffusers clip_sample / clip_sample_range); range <= 0 disables
 * (the default, as in the SD-1.5 scheduler config). */
int mrisr_sampler_set_clip(mrisr_sampler* s, float clip_sample_range);

/* ---- T2I-Adapter training (in BASELINE config 3 the adapter runs, and is differentiated, every step)
 * Same ownership model as the LoRA step: the caller owns ONE flat f32 vector of all adapter parameters (PyTorch layouts,
 * state-dict keys via tensor_info) and its gradient.  After train_prepare / train_bind, mrisr_adapter_forward keeps what the
 * backward needs; mrisr_adapter_backward takes the gradients w.r.t. the four feature maps (what mrisr_train_step wrote
 * through mrisr_train_set_intrablock_grads) and ADDS dW, db of every conv to the gradient vector (dgrad convs + one
 * pixel-contraction GEMM per conv).  refresh re-packs the kernels' weight layouts after the optimiser step. */
int mrisr_adapter_train_prepare(mrisr_adapter* a, void* stream);
int64_t mrisr_adapter_train_num_trainable(const mrisr_adapter* a);
int mrisr_adapter_train_num_tensors(const mrisr_adapter* a);
int mrisr_adapter_train_tensor_info(const mrisr_adapter* a, int i, const char** key, int64_t* offset, int64_t shape[4], int* ndim);
int mrisr_adapter_train_bind(mrisr_adapter* a, float* theta_dev, float* grad_dev, int init_from_model, void* stream);
int mrisr_adapter_train_refresh(mrisr_adapter* a, void* stream);
int mrisr_adapter_backward(mrisr_adapter* a, const mrisr_tensor* d_feats, int n_feats, void* stream);
/* The same pass cut at level boundaries, so that the host can start the exchange of a level's finished weight gradients while
 * the lower levels are still being differentiated: call with level = top level, ..., 0 (level 0 also
 * differentiates conv_in).  mrisr_adapter_train_level_range: that level's contiguous range of the flat trainable vector. */
int mrisr_adapter_backward_level(mrisr_adapter* a, const mrisr_tensor* d_feats, int n_feats, int level, void* stream);
int mrisr_adapter_train_level_range(const mrisr_adapter* a, int level, int64_t* offset, int64_t* numel);

/* ---- AutoencoderKL (SD-1.5 VAE): pixel <-> latent, once before / once after the sampling loop --------------
 * Replaces vae.encode(x).latent_dist (res_srdiff.py:49-50) and vae.decode(z).sample (res_srdiff.py:107-110); the
 * arithmetic is diffusers' AutoencoderKL (state-dict keys encoder.* / decoder.* / quant_conv / post_quant_conv).
 *   encode: image [B,3,H,W] -> moments [B, 2*latent, H/8, W/8] = (mean | logvar) of the diagonal Gaussian posterior;
 *           sampling (mean + exp(0.5*logvar)*eps) and the scaling_factor stay with the caller, as in the reference.
 *   decode: latents [B,latent,h,w] (already divided by scaling_factor) -> image [B,3,8h,8w], NCHW in image->dtype. */
typedef struct {
    int32_t in_channels;           /* 3 */
    int32_t out_channels;          /* 3 */
    int32_t latent_channels;       /* 4 */
    int32_t num_levels;            /* 4 */
    int32_t block_out_channels[4]; /* 128, 256, 512, 512 */
    int32_t layers_per_block;      /* 2 */
    int32_t norm_num_groups;       /* 32 */
    int32_t compute_dtype;         /* MRISR_BF16 or MRISR_F32 */
    float scaling_factor;          /* 0.18215 (carried for the host mirror; not applied by encode / decode) */
} mrisr_vae_cfg;
typedef struct mrisr_vae mrisr_vae;
int mrisr_vae_create(const mrisr_vae_cfg* cfg, mrisr_vae** out);
void mrisr_vae_destroy(mrisr_vae* v);
int mrisr_vae_set_param(mrisr_vae* v, const char* key, const float* data, const int64_t* shape, int ndim, int is_device);
int64_t mrisr_vae_num_params(const mrisr_vae* v);
int mrisr_vae_finalize(mrisr_vae* v, void* stream);
int mrisr_vae_encode(mrisr_vae* v, const mrisr_tensor* image, mrisr_tensor* moments, void* stream);
int mrisr_vae_decode(mrisr_vae* v, const mrisr_tensor* latents, mrisr_tensor* image, void* stream);

/* ---- LoRA fine-tuning step ---------------------------------------------------
 * Replaces, for the UNet handle, what the reference's training cell gets from torch autograd + accelerate
 * (notebook ResDif c11:14-41: noise_pred = unet(noisy, t, ehs).sample; loss = mse(noise_pred, noise);
 *  accelerator.backward(loss); clip_grad_norm_(1.0); optimizer.step()).  Base weights are frozen; the trainable
 * parameters are the peft adapters (<module>.lora_A/B.default.weight), kept by the CALLER as ONE flat f32 device
 * vector `theta` with a gradient vector `grad` of the same length - the bucket a data-parallel host all-reduces.
 *   prepare      - packs the transposed / tap-flipped weight copies the dX GEMMs read and lays out theta
 *   tensor_info  - i-th adapter tensor: state-dict key, element offset in theta, shape {rows, cols}
 *   bind         - attaches theta / grad (init_from_model: theta <- the adapters loaded with set_param)
 *   step         - forward, loss = mean((eps_hat - target)^2) -> *loss_dev, backward; adapter gradients are ADDED to
 *                  grad (zero it per optimiser step; accumulating several micro-batches is allowed).  target: f32
 *                  NCHW.  pred_out (optional): eps_hat.  The model must be created with lora_fused = 1.
 *   refresh      - re-packs the adapters from theta after the optimiser changed it
 *   optim_sumsq  - *out_dev += sum(g^2)   (global grad norm; all-reduce it with the gradients)
 *   optim_adamw  - torch.optim.AdamW update on flat vectors; g is first scaled by grad_scale (1/world_size after a
 *                  sum all-reduce) and, when max_norm > 0, by min(1, max_norm / (grad_scale*sqrt(*sumsq_dev) + 1e-6))
 *                  as torch.nn.utils.clip_grad_norm_ does; step counts from 1. */
int mrisr_train_prepare(mrisr_model* m, void* stream);
int64_t mrisr_train_num_trainable(const mrisr_model* m);
int mrisr_train_num_tensors(const mrisr_model* m);
int mrisr_train_tensor_info(const mrisr_model* m, int i, const char** key, int64_t* offset, int64_t shape[2]);
int mrisr_train_bind(mrisr_model* m, float* theta_dev, float* grad_dev, int init_from_model, void* stream);
int mrisr_train_refresh(mrisr_model* m, void* stream);
int mrisr_train_step(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                     const mrisr_tensor* intrablock, int n_intrablock, const mrisr_tensor* target, float* loss_dev,
                     mrisr_tensor* pred_out, void* stream);
/* Gradients w.r.t. the T2I-Adapter features (intrablock[i] of the following mrisr_train_step calls) are written to
 * grads[i] (same shapes; NCHW f32/bf16 or NHWC compute dtype) for the adapter's own backward; n = 0 switches it off. */
int mrisr_train_set_intrablock_grads(mrisr_model* m, const mrisr_tensor* grads, int n);
/* ControlNet residuals for the following mrisr_train_step calls (reference call shape: unet(..., down_block_additional_residuals=down_res,
 * mid_block_additional_residual=mid_res), src/adapters/res_srdiff.py:73-78, inside the training graph): down[k] / mid are added to the skips /
 * the mid block's output (out of place, as diffusers does), and d(loss)/d(down[k]), d(loss)/d(mid) are written to d_down[k] / d_mid (same
 * shapes; may be null) - the seeds of the ControlNet's own backward (mrisr_controlnet_train_step).  n_down = 0, mid = NULL switches it off.
 * The UNet may be frozen (lora_rank 0: mrisr_train_bind(m, NULL, NULL, 0, stream)): the step then only computes these input gradients. */
int mrisr_train_set_controlnet_residuals(mrisr_model* m, const mrisr_tensor* down, const mrisr_tensor* d_down, int n_down,
                                         const mrisr_tensor* mid, const mrisr_tensor* d_mid);
/* ---- ControlNet with its own parameters trainable (NOT in the reference's code: it only runs a ControlNet for inference,
 * src/adapters/res_srdiff.py:65-70; SURVEY.md 3.2 lists it as a training configuration).  The ControlNet handle's raw tensors live in
 * ONE flat f32 vector the caller owns (offsets by mrisr_controlnet_train_tensor_info, key order = sorted state-dict names); one training
 * step is  train_forward (recorded; writes the 12 + 1 residuals) -> mrisr_train_step of the UNet with those residuals
 * (mrisr_train_set_controlnet_residuals: it writes d(loss)/d(residual)) -> train_backward (adds d(loss)/d(parameter) to the gradient
 * vector) -> all-reduce, mrisr_optim_adamw on the flat vectors -> train_refresh (re-packs every weight in place).
 * `differentiated` = 0 marks tensors whose gradient this build leaves at zero (norm affine parameters, the time-embedding MLP and its
 * per-block projections, the condition embedding): keep them out of the optimiser or accept that they stay frozen. */
int mrisr_controlnet_train_prepare(mrisr_model* m, void* stream);
int64_t mrisr_controlnet_train_num_trainable(const mrisr_model* m);
int mrisr_controlnet_train_num_tensors(const mrisr_model* m);
int mrisr_controlnet_train_tensor_info(const mrisr_model* m, int i, const char** key, int64_t* offset, int64_t* numel, int* differentiated);
int mrisr_controlnet_train_bind(mrisr_model* m, float* theta_dev, float* grad_dev, int init_from_model, void* stream);
int mrisr_controlnet_train_refresh(mrisr_model* m, void* stream);
int mrisr_controlnet_train_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                                   const mrisr_tensor* cond, float conditioning_scale, mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out,
                                   void* stream);
int mrisr_controlnet_train_backward(mrisr_model* m, const mrisr_tensor* d_down, int n_down, const mrisr_tensor* d_mid, float conditioning_scale,
                                    void* stream);
int mrisr_optim_sumsq(const float* g_dev, int64_t n, float* out_dev, void* stream);
/* ema = decay * ema + (1 - decay) * theta  (diffusers EMAModel.step on the flat trainable vector) */
int mrisr_optim_ema(float* ema_dev, const float* theta_dev, int64_t n, float decay, void* stream);
int mrisr_optim_adamw(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, const float* sumsq_dev,
                      float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int step, void* stream);

/* ---- image metrics of the reference's evaluator (src/eval/eval.py:15-51) --------------------------------------
 * pred / gt: f32 [batch][height][width] in [0, 1] (the reference divides its 8-bit PNGs by 255).  out: f32 [batch][4] =
 * {PSNR (torchmetrics, data_range 1), SSIM (torchmetrics defaults: 11x11 Gaussian sigma 1.5, k1 .01, k2 .03, mean over
 * the fully-inside windows), HFEN (||LoG(pred) - LoG(gt)|| / (||LoG(gt)|| + 1e-8), sigma 1.5), NMSE}.
 * scratch: 2*batch*height*width floats; sums: batch*6 doubles (zeroed here). */
int mrisr_image_metrics(const float* pred_dev, const float* gt_dev, int batch, int height, int width, float* scratch_dev,
                        double* sums_dev, float* out_dev, void* stream);

/* ---- slice degradation of the reference's data pipeline (nb ResDif c22:102-154) ----------------
 * All images are f32 [batch][height][width] on the device.  scratch sizes come from the *_scratch_bytes functions.
 *   resize_slices:        Pillow Image.resize on mode "F" (filter 0 = BICUBIC a=-0.5, 1 = LANCZOS a=3; antialiased support when
 *                         shrinking) - replaces FastMRILazyDataset._pad_to_target's resize            nb ResDif c22:102-113
 *   gaussian_blur_slices: scipy.ndimage.gaussian_filter(sigma, mode "reflect", truncate)               nb ResDif c22:143-144
 *   simulate_low_field:   blur(sigma 0.5*scale) -> BICUBIC to the small size -> BICUBIC back            nb ResDif c22:140-154
 *                         (small size = (W//scale) rows x (H//scale) columns, the reference's own tuple order) */
size_t mrisr_resize_scratch_bytes(int batch, int height, int width, int out_height, int out_width, int filter);
int mrisr_resize_slices(const float* in_dev, int batch, int height, int width, float* out_dev, int out_height, int out_width, int filter,
                        void* scratch_dev, size_t scratch_bytes, void* stream);
int mrisr_gaussian_blur_slices(const float* in_dev, int batch, int height, int width, float sigma, float truncate, float* tmp_dev,
                               float* out_dev, void* stream);
size_t mrisr_low_field_scratch_bytes(int batch, int height, int width, float scale_factor);
int mrisr_simulate_low_field(const float* hr_dev, int batch, int height, int width, float scale_factor, float* lr_dev, void* scratch_dev,
                             size_t scratch_bytes, void* stream);

/* ---- per-launch HIP-event profiler (bench.py roofline leg; off by default) ------------------------- */
int mrisr_prof_enable(int on);
int mrisr_prof_reset(void);
/* JSON {"<kernel class>": {"launches", "ms", "flops", "bytes"}}; returns the length written or -1 if buf is too small */
int mrisr_prof_report(char* buf, int cap);

/* ---- single-op entry points (used by the parity tests; same kernels the models launch) ------------ */
int mrisr_op_conv3x3(const mrisr_tensor* x_nhwc, const mrisr_tensor* x2_nhwc, const float* w_oihw_dev,
                     const float* bias_dev, int cout, int stride, int upsample, int act, int splitk, int tile,
                     mrisr_tensor* y_nhwc, void* stream);
int mrisr_op_linear(const mrisr_tensor* x_rows, const float* w_dev, const float* bias_dev, int n, int act,
                    int splitk, int tile, mrisr_tensor* y_rows, void* stream);
/* y = LayerNorm(x; gamma, beta, eps 1e-5) W^T + bias with the normalisation as a prologue of the row-panel GEMM kernel (bf16;
 * K = 320 or 640, n % 16 == 0) - the form the transformer blocks use for norm1/2/3 -> to_q|k|v / to_q / ff.net.0.proj */
int mrisr_op_ln_linear(const mrisr_tensor* x_rows, const float* gamma_dev, const float* beta_dev, const float* w_dev,
                       const float* bias_dev, int n, int act, mrisr_tensor* y_rows, void* stream);
/* y = [x +] FF2(GEGLU(FF1(LayerNorm(x)))) - the feed-forward of BasicTransformerBlock (diffusers attention.py: norm3 -> ff -> + hidden)
 * in ONE kernel at C = 320 (bf16): w1 = ff.net.0.proj.weight [2*hidden][320], w2 = ff.net.2.weight [320][hidden], f32 on the device */
int mrisr_op_mlp(const mrisr_tensor* x_rows, const float* gamma_dev, const float* beta_dev, const float* w1_dev, const float* b1_dev,
                 const float* w2_dev, const float* b2_dev, int hidden, int residual, mrisr_tensor* y_rows, void* stream);
/* the fp8 form of the same projection (BASELINE configs[4]): weights quantised to OCP e4m3 with one scale per output channel, the
 * rows with one scale per row inside the kernel, f32 accumulate; gamma_dev / beta_dev NULL = no LayerNorm prologue */
int mrisr_op_linear_fp8(const mrisr_tensor* x_rows, const float* gamma_dev, const float* beta_dev, const float* w_dev,
                        const float* bias_dev, int n, int act, mrisr_tensor* y_rows, void* stream);
int mrisr_op_groupnorm(const mrisr_tensor* x_nhwc, const mrisr_tensor* x2_nhwc, const float* gamma_dev,
                       const float* beta_dev, int groups, float eps, int silu, mrisr_tensor* y_nhwc, void* stream);
int mrisr_op_layernorm(const mrisr_tensor* x_rows, const float* gamma_dev, const float* beta_dev, float eps,
                       mrisr_tensor* y_rows, void* stream);
/* q,k,v rows [B,N,H*d] (k,v: [B,Nk,H*d]) -> out [B,N,H*d]; flash=1 uses the fused kernel (bf16) */
int mrisr_op_attention(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, int heads, int flash,
                       mrisr_tensor* out, void* stream);

/* gradients of mrisr_op_attention (bf16, flash path): dq [B,N,H*d], dk, dv [B,Nk,H*d] for an upstream dout [B,N,H*d] */
int mrisr_op_attention_bwd(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, const mrisr_tensor* dout,
                           int heads, mrisr_tensor* dq, mrisr_tensor* dk, mrisr_tensor* dv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRISR_H */
