// Model object behind the opaque mrisr_model handle (UNet2DConditionModel or ControlNetModel).
#pragma once
#include "runtime.h"

namespace mrisr {

struct ResW {
    NormW n1, n2;
    ConvW c1, c2;
    bool has_sc = false;
    LinW sc;
    int temb_off = 0, cin = 0, cout = 0;
};
struct XfW {
    int C = 0;
    NormW norm, ln1, ln2, ln3;
    LinW proj_in, proj_out, qkv, out1, q2, kv2, out2, ff1, ff2;
    void* ff2p = nullptr;  // ff2.w with K permuted for the fused feed-forward kernel (bf16, C = 320 only)
    void* kc = nullptr;   // cached cross-attention K   [B*H][ctx_pad][dpad]
    void* vtc = nullptr;  // cached cross-attention V^T [B*H][dpad][ctx_pad]
    // fused row-local middle of the block (xtail.hip; bf16, C = 320): attn2.to_q / attn2.to_out weights with K permuted to the
    // accumulator order, and the prompt K / V as per-(image, head pair) LDS images (planned with kc / vtc, packed by set_context)
    void* q2p = nullptr;
    void* proj_outp = nullptr;  // proj_out.w K-permuted: the continuation of the fused feed-forward kernel
    void* out2p = nullptr;
    void* kvp = nullptr;
};
struct Level {
    std::vector<ResW> res;
    std::vector<XfW> xf;
    bool has_down = false, has_up = false;
    ConvW down, up;
    void* up_sp = nullptr;  // bf16 inference: `up` as four 2 x 2 parity banks [4][cout][2][2][cin] (sub-pixel form, launch_pack_conv_subpix)
};
struct HeadBuf {  // head-major q / k / v^T staging for one (tokens, channels) geometry; pads stay zero
    int B = 0, H = 0, N = 0, hd = 0, npad = 0, dpad = 0;
    void* q = nullptr;
    void* k = nullptr;
    void* vt = nullptr;
};

const char* last_error_cstr();

struct Model {
    mrisr_unet_cfg cfg{};
    bool is_controlnet = false;
    std::map<std::string, RawParam> raw;
    float lora_scale = 1.0f;
    bool finalized = false;
    std::vector<std::unique_ptr<DevBuf>> packed;

    // packed modules
    ConvW conv_in, conv_out;
    NormW norm_out;
    LinW te1, te2, tproj;
    std::vector<std::string> temb_mods;
    int tproj_total = 0;
    std::vector<Level> down, up;
    ResW mid_r0, mid_r1;
    XfW mid_xf;
    std::vector<ConvW> ce;  // ControlNet condition embedding
    std::vector<int> ce_stride;
    std::vector<LinW> cn_down;
    LinW cn_mid;

    // workspace (planned per input geometry)
    Arena arena;
    DevBuf persist;
    std::string ws_key;
    unsigned long long ws_gen = 0;  // bumped whenever persist / arena are re-planned or reallocated: captured graphs that
                                    // baked the old addresses in must be re-captured (mrisr_sampler_run keys on it)
    int ws_B = 0, ws_h = 0, ws_w = 0, ctx_len = 0, ctx_pad = 0;
    std::vector<HeadBuf> heads;
    void* ctx_rows = nullptr;
    void* cond_emb = nullptr;
    size_t cond_emb_bytes = 0;
    bool ctx_valid = false, cond_valid = false;
    bool keep = false;  // training forward: never release arena temporaries (the backward reads them)
    // per-forward state
    float* tproj_out = nullptr;
    // fused sampler: the time embedding of EVERY step of the schedule computed once per run (they depend on the timestep only); the step then
    // copies row (*tproj_step - tproj_first) instead of running the sinusoid + three GEMVs (50 MB of weights per step).  Null outside a run.
    const float* tproj_table = nullptr;
    const int* tproj_step = nullptr;
    int tproj_first = 0;
    int build_tproj_table(const long long* ts_dev, int rows, float* scratch, float* table, hipStream_t st);
    float *te_s = nullptr, *te_y1 = nullptr, *te_emb = nullptr;  // the time-embedding MLP's activations of this forward (training reads them)
    float* d_tproj = nullptr;                                     // full-parameter training: d(loss)/d(tproj_out), [rows][tproj_total]
    int t_scalar = 1;

    int set_param(const char* key, const float* data, const int64_t* shape, int ndim, int is_device);
    int64_t num_params() const;
    const RawParam* find(const std::string& k) const;
    void* new_packed(size_t bytes, bool zero);
    int finalize(hipStream_t st);
    int num_skips() const;
    int skip_shape(int k, int B, int h, int w, int64_t shape[4]) const;
    const HeadBuf& head_buf(int N, int C) const;
    const HeadBuf& head_buf_for_C(int C) const;
    int ensure_workspace(int B, int h, int w, int L, hipStream_t st);
    int forward_unet(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                     const mrisr_tensor* down_res, int n_down, const mrisr_tensor* mid_res,
                     const mrisr_tensor* intrablock, int n_intra, mrisr_tensor* out, hipStream_t st);
    int forward_controlnet(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                           const mrisr_tensor* cond, float scale, mrisr_tensor* down_out, int n_down,
                           mrisr_tensor* mid_out, hipStream_t st);
    int set_context(const mrisr_tensor* ehs, int B, int h, int w, hipStream_t st);
    int set_cond(const mrisr_tensor* cond, int L, hipStream_t st);

    // ---- LoRA fine-tuning (train.hip) ----
    struct Trainable { std::string key; long long offset, numel; int rows, cols; };
    std::vector<Trainable> trainables;   // lora_A / lora_B tensors in flat-vector order
    long long n_trainable = 0;
    float* theta = nullptr;              // caller-owned flat f32 parameter vector (bound)
    float* grad = nullptr;               // caller-owned flat f32 gradient vector (bound)
    bool train_ready = false;
    std::vector<mrisr_tensor> d_intra;   // optional outputs: gradients w.r.t. the T2I-Adapter features of the next train_step
    // ControlNet residuals of the next train_step (inputs) and the tensors their gradients are written to (outputs): 12 + mid
    std::vector<mrisr_tensor> tr_down, d_tr_down;
    mrisr_tensor tr_mid{}, d_tr_mid{};
    bool has_tr_mid = false;
    std::string train_ws_key;
    // ---- full-parameter training (ControlNet: every raw tensor is trainable; train.hip "full gradient" blocks) ----
    std::map<std::string, long long> full_off;   // raw key -> offset in the caller's flat f32 vectors
    std::vector<Trainable> full_trainables;      // every raw tensor, in key order (rows = shape[0], cols = the rest)
    std::vector<std::string> full_unsupported;   // tensors whose gradient this build leaves at zero (reported to the host)
    long long n_full = 0;
    float* full_theta = nullptr;                 // caller-owned flat parameters / gradients (bound by full_train_bind)
    float* full_grad = nullptr;
    std::shared_ptr<void> full_tape;             // the recorded forward of full_train_forward (a Trainer<T>), consumed by full_train_backward
    int full_train_prepare(hipStream_t st);
    int full_train_bind(float* theta_dev, float* grad_dev, int init_from_model, hipStream_t st);
    int full_train_refresh(hipStream_t st);      // copy theta back into the f32 masters and re-pack every weight
    int repack(hipStream_t st);                  // finalize again INTO the existing packed buffers (same parameter set, new values)
    bool repacking = false;
    size_t repack_cursor = 0;
    int controlnet_train_forward(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs, const mrisr_tensor* cond,
                                 float scale, mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out, hipStream_t st);
    int controlnet_train_backward(const mrisr_tensor* d_down, int n_down, const mrisr_tensor* d_mid, float scale, hipStream_t st);
    std::vector<LinW*> lora_linears();   // every LinW that carries adapters, fixed order
    int train_prepare(hipStream_t st);   // dgrad weight copies + trainable layout (after finalize)
    int train_bind(float* theta_dev, float* grad_dev, hipStream_t st);
    int lora_refresh(hipStream_t st);    // re-pack the adapters from theta
    int train_step(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                   const mrisr_tensor* intrablock, int n_intra, const mrisr_tensor* target, float* loss_dev,
                   mrisr_tensor* pred_out, hipStream_t st);
};

}  // namespace mrisr
