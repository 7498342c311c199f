// C ABI (include/mrisr.h): thin extern "C" layer over Model, plus the sampler (per-step hipGraph), the
// T2I-Adapter runtime and the single-op entry points the parity tests call.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "model.h"

using namespace mrisr;

struct mrisr_model : public Model {};

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)
#define API_BEGIN try {
#define API_END                                              \
    }                                                        \
    catch (const std::exception& e) {                        \
        set_error(std::string("exception: ") + e.what());    \
        return 99;                                           \
    }

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" {

const char* mrisr_last_error(void) { return last_error_cstr(); }
const char* mrisr_version(void) { return "mrisr 0.1 (gfx950)"; }

static int create_model(const mrisr_unet_cfg* cfg, bool cn, mrisr_model** out) {
    API_BEGIN
    MRISR_REQUIRE(cfg && out, "null argument");
    MRISR_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 4, "num_levels 1..4");
    MRISR_REQUIRE(cfg->compute_dtype == MRISR_F32 || cfg->compute_dtype == MRISR_BF16, "compute dtype f32 or bf16");
    const int bk = cfg->compute_dtype == MRISR_F32 ? 32 : 64;
    for (int i = 0; i < cfg->num_levels; ++i) {
        MRISR_REQUIRE(cfg->block_out_channels[i] % bk == 0, "block_out_channels must be multiples of the 128-byte K tile");
        MRISR_REQUIRE(cfg->block_out_channels[i] % cfg->num_heads == 0 && (cfg->block_out_channels[i] / cfg->num_heads) % 4 == 0,
                      "head dim must be a multiple of 4");
        MRISR_REQUIRE(cfg->block_out_channels[i] % cfg->norm_num_groups == 0, "channels vs norm groups");
    }
    MRISR_REQUIRE(cfg->cross_attention_dim % bk == 0, "cross_attention_dim must be a multiple of the K tile");
    int dev_count = 0;
    MRISR_CHECK_HIP(hipGetDeviceCount(&dev_count));
    MRISR_REQUIRE(dev_count > 0, "no HIP device: libmrisr has no CPU fallback");
    auto* m = new mrisr_model();
    m->cfg = *cfg;
    m->is_controlnet = cn;
    *out = m;
    return 0;
    API_END
}
int mrisr_unet_create(const mrisr_unet_cfg* cfg, mrisr_model** out) { return create_model(cfg, false, out); }
int mrisr_controlnet_create(const mrisr_unet_cfg* cfg, mrisr_model** out) { return create_model(cfg, true, out); }
void mrisr_model_destroy(mrisr_model* m) { delete m; }

int mrisr_model_set_param(mrisr_model* m, const char* key, const float* data, const int64_t* shape, int ndim,
                          int is_device) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->set_param(key, data, shape, ndim, is_device);
    API_END
}
int mrisr_model_set_lora_scale(mrisr_model* m, float scale) {
    MRISR_REQUIRE(m, "null handle");
    m->lora_scale = scale;
    m->finalized = false;
    return 0;
}
int mrisr_model_finalize(mrisr_model* m, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    TRY(gemm_prepare());
    return m->finalize((hipStream_t)stream);
    API_END
}
int64_t mrisr_model_num_params(const mrisr_model* m) { return m ? m->num_params() : 0; }
int64_t mrisr_model_workspace_bytes(const mrisr_model* m) {
    return m ? (int64_t)(m->arena.buf.bytes + m->persist.bytes) : 0;
}
int mrisr_model_num_skips(const mrisr_model* m) { return m ? m->num_skips() : 0; }
int mrisr_model_skip_shape(const mrisr_model* m, int k, int B, int h, int w, int64_t shape[4]) {
    MRISR_REQUIRE(m, "null handle");
    return m->skip_shape(k, B, h, w, shape);
}

int mrisr_unet_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep,
                       const mrisr_tensor* ehs, const mrisr_tensor* down_res, int n_down_res,
                       const mrisr_tensor* mid_res, const mrisr_tensor* intrablock, int n_intrablock,
                       mrisr_tensor* out, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->forward_unet(sample, timestep, ehs, down_res, n_down_res, mid_res, intrablock, n_intrablock, out,
                           (hipStream_t)stream);
    API_END
}
int mrisr_model_set_context(mrisr_model* m, const mrisr_tensor* ehs, int latent_h, int latent_w, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m && ehs, "null argument");
    return m->set_context(ehs, (int)ehs->shape[0], latent_h, latent_w, (hipStream_t)stream);
    API_END
}
int mrisr_controlnet_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep,
                             const mrisr_tensor* ehs, const mrisr_tensor* cond, float conditioning_scale,
                             mrisr_tensor* down_out, int n_down_out, mrisr_tensor* mid_out, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->forward_controlnet(sample, timestep, ehs, cond, conditioning_scale, down_out, n_down_out, mid_out,
                                 (hipStream_t)stream);
    API_END
}
int mrisr_controlnet_set_cond(mrisr_model* m, const mrisr_tensor* cond, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m && cond, "null argument");
    MRISR_REQUIRE(m->ctx_len > 0, "set the context (or run one forward) before caching the condition image");
    return m->set_cond(cond, m->ctx_len, (hipStream_t)stream);
    API_END
}

int mrisr_resshift_forward(const mrisr_tensor* hr, const mrisr_tensor* lr, const mrisr_tensor* noise,
                           const float* alphas_cumprod_dev, const mrisr_tensor* timestep, mrisr_tensor* out,
                           void* stream) {
    API_BEGIN
    MRISR_REQUIRE(hr && lr && noise && out && timestep && alphas_cumprod_dev, "null argument");
    MRISR_REQUIRE(hr->dtype == MRISR_F32 && lr->dtype == MRISR_F32 && noise->dtype == MRISR_F32 && out->dtype == MRISR_F32,
                  "f32 latents");
    MRISR_REQUIRE(timestep->dtype == MRISR_I64, "int64 timesteps");
    const int B = (int)hr->shape[0];
    long long per = 1;
    for (int i = 1; i < hr->ndim; ++i) per *= hr->shape[i];
    const int scalar = timestep->ndim == 0 || timestep->shape[0] == 1;
    MRISR_REQUIRE(timestep->ndim <= 1 && (scalar || timestep->shape[0] == B), "timesteps: 0-dim, [1] or [B]");
    {
        long long nl = 1, nn = 1, no = 1;
        for (int i = 0; i < lr->ndim; ++i) nl *= lr->shape[i];
        for (int i = 0; i < noise->ndim; ++i) nn *= noise->shape[i];
        for (int i = 0; i < out->ndim; ++i) no *= out->shape[i];
        MRISR_REQUIRE(nl == per * B && nn == per * B && no == per * B, "hr / lr / noise / out must have the same shape");
    }
    return launch_resshift_forward((const float*)hr->data, (const float*)lr->data, (const float*)noise->data,
                                   alphas_cumprod_dev, (const long long*)timestep->data, scalar, (float*)out->data, B,
                                   per, (hipStream_t)stream);
    API_END
}

}  // extern "C"

// =================================================================================================
// sampler
// =================================================================================================
__global__ void load_t_kernel(long long* cur_t, const long long* table, const int* step) { *cur_t = table[*step]; }

static int g_temb_table = -1;  // test hook: -1 = MRISR_TEMB_TABLE (default 1), 0 off, 1 on
extern "C" void mrisr_debug_temb_table(int on) { g_temb_table = on; }
static bool temb_table_enabled() {
    static const int env = [] { const char* e = getenv("MRISR_TEMB_TABLE"); return e ? atoi(e) : 1; }();
    return g_temb_table < 0 ? env != 0 : g_temb_table != 0;
}
struct mrisr_sampler {
    mrisr_model* unet = nullptr;
    mrisr_model* cnet = nullptr;
    int kind = 0, n_steps = 0, first = 0, last = 0;
    float clip = 0.f;
    std::vector<float> sigma;  // host copy of each step's noise coefficient (which steps read a step_noise slab)
    DevBuf d_ts, d_coef, d_step, d_curt, d_eps;
    DevBuf tp_unet, tp_cnet;  // per-run time-embedding tables [scratch | n_steps x tproj_total] (f32)
    std::vector<std::unique_ptr<DevBuf>> res_bufs;   // ControlNet -> UNet residuals (NHWC, compute dtype)
    std::vector<std::unique_ptr<DevBuf>> intra_bufs;  // adapter features converted once
    hipGraphExec_t exec = nullptr;
    std::string graph_key;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    ~mrisr_sampler() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
    }
};

extern "C" {

int mrisr_sampler_create(mrisr_model* unet, mrisr_model* controlnet, int step_kind, const int64_t* timesteps,
                         int n_steps, const float* alphas_cumprod, int n_train, mrisr_sampler** out) {
    API_BEGIN
    MRISR_REQUIRE(unet && timesteps && alphas_cumprod && out && n_steps > 0, "bad argument");
    MRISR_REQUIRE(!controlnet || controlnet->cfg.compute_dtype == unet->cfg.compute_dtype, "UNet/ControlNet dtype mismatch");
    std::unique_ptr<mrisr_sampler> s(new mrisr_sampler());
    s->unet = unet;
    s->cnet = controlnet;
    s->kind = step_kind;
    s->n_steps = n_steps;
    s->first = 0;
    s->last = n_steps;
    std::vector<long long> ts(timesteps, timesteps + n_steps);
    MRISR_REQUIRE(step_kind >= MRISR_STEP_DDIM && step_kind <= MRISR_STEP_DDPM, "unknown step kind");
    std::vector<float> coef((size_t)n_steps * 8, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        const long long t = ts[i];
        MRISR_REQUIRE(t >= 0 && t < n_train, "timestep out of range");
        // zero-terminal-SNR tables (rescale_betas_zero_snr, nb ResDif c11:46) end in abar = 0 exactly and "trailing" spacing
        // samples that entry first: every step divides by sqrt(abar_t) (res_srdiff.py:86), so it is clamped to 2^-24 here
        // (the reference itself would produce inf)
        const double a_t = std::max((double)alphas_cumprod[t], 5.9604644775390625e-8);
        if (step_kind == MRISR_STEP_DDIM) {
            // t_prev = t - T/n; alpha_prev = alpha[t_prev] or alpha[0] (set_alpha_to_one=False)
            const long long tp = t - n_train / n_steps;
            const double a_p = tp >= 0 ? alphas_cumprod[tp] : alphas_cumprod[0];
            coef[2 * i] = (float)std::sqrt(a_p / a_t);
            coef[2 * i + 1] = (float)(std::sqrt(1.0 - a_p) - std::sqrt(a_p * (1.0 - a_t) / a_t));
        } else if (step_kind == MRISR_STEP_DDPM) {
            // diffusers DDPMScheduler.step, variance_type "fixed_small" (un-vendored; BASELINE config 1)
            const long long tp = t - n_train / n_steps;
            const double a_p = tp >= 0 ? alphas_cumprod[tp] : 1.0;
            const double al = a_t / a_p, be = 1.0 - al;
            coef[8 * i] = (float)(1.0 / std::sqrt(a_t));
            coef[8 * i + 1] = (float)(std::sqrt(1.0 - a_t) / std::sqrt(a_t));
            coef[8 * i + 2] = (float)(std::sqrt(a_p) * be / (1.0 - a_t));
            coef[8 * i + 3] = (float)(std::sqrt(al) * (1.0 - a_p) / (1.0 - a_t));
            coef[8 * i + 4] = t > 0 ? (float)std::sqrt(std::max((1.0 - a_p) / (1.0 - a_t) * be, 1e-20)) : 0.f;
        } else {
            // reference res_srdiff.py:83-96: prev_t = timesteps[i+1] or 0; last step uses alpha[0] and no noise
            const long long tp = i + 1 < n_steps ? ts[i + 1] : 0;
            const double a_p = alphas_cumprod[tp];
            coef[4 * i] = (float)std::sqrt(a_t);
            coef[4 * i + 1] = (float)std::sqrt(1.0 - a_t);
            coef[4 * i + 2] = (float)std::sqrt(a_p);
            coef[4 * i + 3] = tp > 0 ? (float)std::sqrt((1.0 - a_p) / (1.0 - a_t) * (1.0 - a_t / a_p)) : 0.f;
        }
    }
    s->sigma.assign(n_steps, 0.f);
    for (int i = 0; i < n_steps; ++i)
        s->sigma[i] = step_kind == MRISR_STEP_DDPM ? coef[8 * i + 4] : (step_kind == MRISR_STEP_RESSHIFT ? coef[4 * i + 3] : 0.f);
    TRY(s->d_ts.reserve(sizeof(long long) * n_steps, false));
    TRY(s->d_coef.reserve(sizeof(float) * coef.size(), false));
    TRY(s->d_step.reserve(16, true));
    TRY(s->d_curt.reserve(16, true));
    MRISR_CHECK_HIP(hipMemcpy(s->d_ts.p, ts.data(), sizeof(long long) * n_steps, hipMemcpyHostToDevice));
    MRISR_CHECK_HIP(hipMemcpy(s->d_coef.p, coef.data(), sizeof(float) * coef.size(), hipMemcpyHostToDevice));
    *out = s.release();
    return 0;
    API_END
}
void mrisr_sampler_destroy(mrisr_sampler* s) { delete s; }
int mrisr_sampler_set_range(mrisr_sampler* s, int first_step, int last_step) {
    MRISR_REQUIRE(s && first_step >= 0 && first_step <= last_step && last_step <= s->n_steps, "step range");
    s->first = first_step;
    s->last = last_step;
    return 0;
}

int mrisr_sampler_set_clip(mrisr_sampler* s, float clip_sample_range) {
    MRISR_REQUIRE(s, "null sampler");
    MRISR_REQUIRE(s->kind == MRISR_STEP_DDPM, "x0 clipping belongs to the DDPM step");
    if (s->clip != clip_sample_range && s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }  // baked into the graph
    s->clip = clip_sample_range;
    return 0;
}

int mrisr_sampler_run(mrisr_sampler* s, mrisr_tensor* latents, const mrisr_tensor* lr_latents,
                      const mrisr_tensor* step_noise, const mrisr_tensor* ehs, const mrisr_tensor* cond,
                      const mrisr_tensor* intrablock, int n_intrablock, int use_graph, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(s && latents && ehs, "null argument");
    MRISR_REQUIRE(latents->ndim == 4 && latents->dtype == MRISR_F32 && latents->layout == MRISR_NCHW, "latents: f32 NCHW");
    MRISR_REQUIRE(s->kind != MRISR_STEP_RESSHIFT || lr_latents, "Res-SRDiff needs the LR anchor latents");
    MRISR_REQUIRE(!s->cnet || cond, "ControlNet needs the condition image");
    hipStream_t user = (hipStream_t)stream;
    hipStream_t st = user;
    Model& U = *s->unet;
    const int B = (int)latents->shape[0], h = (int)latents->shape[2], w = (int)latents->shape[3];
    const int L = (int)ehs->shape[1];
    const long long n = (long long)B * latents->shape[1] * h * w;
    const int cdt = U.cfg.compute_dtype;
    const int esz = dtype_size(cdt);
    // every operand the step kernels index is checked against the latents here: the kernels themselves read
    // lr[i], noise[step * n + i] for i < n without bounds
    auto numel = [](const mrisr_tensor* t) { long long k = 1; for (int i = 0; i < t->ndim; ++i) k *= t->shape[i]; return k; };
    MRISR_REQUIRE(latents->shape[1] == U.cfg.in_channels, "latents channels vs the UNet's in_channels");
    MRISR_REQUIRE(ehs->ndim == 3 && ehs->shape[0] == B && ehs->shape[2] == U.cfg.cross_attention_dim,
                  "encoder_hidden_states must be [B, L, cross_attention_dim] with the latents' batch");
    if (lr_latents) MRISR_REQUIRE(lr_latents->dtype == MRISR_F32 && numel(lr_latents) == n, "lr_latents: f32, same shape as latents");
    if (cond) MRISR_REQUIRE(cond->ndim == 4 && cond->shape[0] == B && cond->shape[2] == 8 * h && cond->shape[3] == 8 * w,
                            "controlnet_cond must be [B, C, 8h, 8w] with the latents' batch");
    {
        int need = 0;  // slabs read: one per step, indexed by the step's position in the schedule, when its sigma != 0
        for (int i = s->first; i < s->last; ++i)
            if (s->sigma[i] != 0.f) need = i + 1;
        if (step_noise) {
            MRISR_REQUIRE(step_noise->dtype == MRISR_F32 && step_noise->ndim >= 1 && numel(step_noise) % n == 0,
                          "step_noise: f32, a stack of latents-shaped slabs");
            MRISR_REQUIRE(numel(step_noise) / n >= need, "step_noise has fewer slabs than the last stochastic step needs");
        }
    }
    for (int i = 0; i < n_intrablock; ++i)
        MRISR_REQUIRE(intrablock[i].ndim == 4 && intrablock[i].shape[0] == B, "adapter features must carry the latents' batch");

    if (use_graph && user == nullptr) {
        // the legacy default stream cannot be captured: run on an internal stream fenced by events
        if (!s->own_stream) {
            MRISR_CHECK_HIP(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
            MRISR_CHECK_HIP(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
            MRISR_CHECK_HIP(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
        }
        st = s->own_stream;
        MRISR_CHECK_HIP(hipEventRecord(s->ev_in, user));
        MRISR_CHECK_HIP(hipStreamWaitEvent(st, s->ev_in, 0));
    }

    // ---- one-time (per run) timestep-invariant work: context projections, condition embedding, features ----
    TRY(gemm_prepare());
    TRY(U.set_context(ehs, B, h, w, st));
    std::vector<mrisr_tensor> down_t;
    mrisr_tensor mid_t{};
    int ns = 0;
    if (s->cnet) {
        Model& C = *s->cnet;
        TRY(C.set_context(ehs, B, h, w, st));
        TRY(C.set_cond(cond, L, st));
        ns = C.num_skips();
        s->res_bufs.resize(ns + 1);
        down_t.resize(ns);
        for (int k = 0; k <= ns; ++k) {
            mrisr_tensor t{};
            t.ndim = 4; t.dtype = cdt; t.layout = MRISR_NHWC;
            C.skip_shape(k, B, h, w, t.shape);
            if (!s->res_bufs[k]) s->res_bufs[k].reset(new DevBuf());
            TRY(s->res_bufs[k]->reserve((size_t)t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3] * esz, false));
            t.data = s->res_bufs[k]->p;
            if (k < ns) down_t[k] = t; else mid_t = t;
        }
    }
    std::vector<mrisr_tensor> intra(n_intrablock);
    s->intra_bufs.resize(n_intrablock);
    for (int i = 0; i < n_intrablock; ++i) {
        const mrisr_tensor& f = intrablock[i];
        MRISR_REQUIRE(f.ndim == 4, "adapter feature rank");
        intra[i] = f;
        if (f.layout == MRISR_NHWC && f.dtype == cdt) continue;
        if (!s->intra_bufs[i]) s->intra_bufs[i].reset(new DevBuf());
        TRY(s->intra_bufs[i]->reserve((size_t)f.shape[0] * f.shape[1] * f.shape[2] * f.shape[3] * esz, false));
        if (cdt == MRISR_F32) TRY(launch_nchw_to_nhwc<float>(f.data, f.dtype, s->intra_bufs[i]->p, (int)f.shape[0], (int)f.shape[1], (int)f.shape[2], (int)f.shape[3], st));
        else TRY(launch_nchw_to_nhwc<bf16>(f.data, f.dtype, s->intra_bufs[i]->p, (int)f.shape[0], (int)f.shape[1], (int)f.shape[2], (int)f.shape[3], st));
        intra[i].data = s->intra_bufs[i]->p;
        intra[i].layout = MRISR_NHWC;
        intra[i].dtype = cdt;
    }
    TRY(s->d_eps.reserve((size_t)n * sizeof(float), false));
    // the time embedding of every step of this run, once (it depends on the timestep only): sinusoid -> MLP -> the 22 per-resnet projections for
    // all rows at once instead of three GEMVs over 50 MB of weights in every step.  MRISR_TEMB_TABLE=0 / mrisr_debug_temb_table(0): per step.
    struct TableGuard {  // the models must not keep pointing at this sampler's table after the run (plain forward calls compute their own)
        Model* a = nullptr; Model* b = nullptr;
        ~TableGuard() { if (a) a->tproj_table = nullptr; if (b) b->tproj_table = nullptr; }
    } tguard;
    const bool use_table = temb_table_enabled();
    if (use_table) {
        const int rows = s->last - s->first;
        auto build = [&](Model& M, DevBuf& buf) -> int {
            const size_t scratch = (size_t)64 * M.cfg.block_out_channels[0] * 9;
            TRY(buf.reserve((scratch + (size_t)rows * M.tproj_total) * sizeof(float), false));
            float* sc = static_cast<float*>(buf.p);
            TRY(M.build_tproj_table(static_cast<const long long*>(s->d_ts.p) + s->first, rows, sc, sc + scratch, st));
            M.tproj_table = sc + scratch; M.tproj_step = static_cast<const int*>(s->d_step.p); M.tproj_first = s->first;
            return 0;
        };
        TRY(build(U, s->tp_unet));
        tguard.a = &U;
        if (s->cnet) { TRY(build(*s->cnet, s->tp_cnet)); tguard.b = s->cnet; }
    }
    MRISR_CHECK_HIP(hipMemsetAsync(s->d_step.p, 0, 16, st));
    MRISR_CHECK_HIP(hipMemcpyAsync(s->d_step.p, &s->first, sizeof(int), hipMemcpyHostToDevice, st));

    mrisr_tensor tt{};
    tt.data = s->d_curt.p; tt.dtype = MRISR_I64; tt.ndim = 0;
    mrisr_tensor eps_t = *latents;
    eps_t.data = s->d_eps.p;
    const int* step = static_cast<const int*>(s->d_step.p);

    auto body = [&]() -> int {
        hipLaunchKernelGGL(load_t_kernel, dim3(1), dim3(1), 0, st, static_cast<long long*>(s->d_curt.p),
                           static_cast<const long long*>(s->d_ts.p), step);
        if (s->cnet)
            TRY(s->cnet->forward_controlnet(latents, &tt, nullptr, nullptr, 1.0f, down_t.data(), ns, &mid_t, st));
        TRY(U.forward_unet(latents, &tt, nullptr, s->cnet ? down_t.data() : nullptr, ns, s->cnet ? &mid_t : nullptr,
                           intra.data(), n_intrablock, &eps_t, st));
        if (s->kind == MRISR_STEP_DDIM)
            TRY(launch_ddim_step((float*)latents->data, (const float*)s->d_eps.p, (const float*)s->d_coef.p, step, n, st));
        else if (s->kind == MRISR_STEP_DDPM)
            TRY(launch_ddpm_step((float*)latents->data, (const float*)s->d_eps.p, step_noise ? (const float*)step_noise->data : nullptr,
                                 (const float*)s->d_coef.p, step, s->clip, n, st));
        else
            TRY(launch_resshift_step((float*)latents->data, (const float*)s->d_eps.p, (const float*)lr_latents->data,
                                     step_noise ? (const float*)step_noise->data : nullptr, (const float*)s->d_coef.p, step, n, st));
        return launch_advance_step(static_cast<int*>(s->d_step.p), st);
    };

    if (!use_graph) {
        for (int i = s->first; i < s->last; ++i) TRY(body());
    } else {
        // everything the captured launches bake in: geometry, caller pointers, the models' workspace generations (persist /
        // arena base addresses change when another geometry, a training step or a second sampler re-plans them), this
        // sampler's own buffers and the feature pointers
        std::string key;
        {
            char kb[256];
            snprintf(kb, sizeof(kb), "%d,%d,%d,%d,%p,%p,%p,%d,g%llu,%llu,e%p", B, h, w, L, latents->data, lr_latents ? lr_latents->data : nullptr,
                     step_noise ? step_noise->data : nullptr, n_intrablock, U.ws_gen, s->cnet ? s->cnet->ws_gen : 0ull, s->d_eps.p);
            key = kb;
            snprintf(kb, sizeof(kb), ",t%p,%p,%d", (const void*)U.tproj_table, s->cnet ? (const void*)s->cnet->tproj_table : nullptr, s->first);
            key += kb;
            for (auto& rb : s->res_bufs) { snprintf(kb, sizeof(kb), ",r%p", rb ? rb->p : nullptr); key += kb; }
            for (auto& f : intra) { snprintf(kb, sizeof(kb), ",f%p", f.data); key += kb; }
        }
        if (!s->exec || s->graph_key != key) {
            if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
            // workspaces are planned (set_context above), so the captured body performs launches only
            hipGraph_t graph = nullptr;
            MRISR_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = body();
            hipError_t e = hipStreamEndCapture(st, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            MRISR_CHECK_HIP(e);
            MRISR_CHECK_HIP(hipGraphInstantiate(&s->exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            s->graph_key = key;
        }
        for (int i = s->first; i < s->last; ++i) MRISR_CHECK_HIP(hipGraphLaunch(s->exec, st));
    }
    if (st != user) {
        MRISR_CHECK_HIP(hipEventRecord(s->ev_out, st));
        MRISR_CHECK_HIP(hipStreamWaitEvent(user, s->ev_out, 0));
    }
    return 0;
    API_END
}

}  // extern "C"

// =================================================================================================
// T2I-Adapter (reference src/adapters/modules.py:114-157, sk=True)
// =================================================================================================
struct AdBlock {
    bool down = false, has_in = false;
    ConvW down_w, in_w, b1, b2;
    int in_c = 0, out_c = 0;
};
struct AdRec {  // what one block's backward needs (activations stay in the arena until the next forward)
    Act x_in, x_down, x_pre, hmid, y;
};
struct AdTrainable {
    std::string key;
    long long offset = 0, numel = 0;
    std::vector<int64_t> shape;
};
struct mrisr_adapter {
    mrisr_adapter_cfg cfg{};
    std::map<std::string, RawParam> raw;
    std::vector<std::unique_ptr<DevBuf>> packed;
    ConvW conv_in;
    std::vector<AdBlock> blocks;
    Arena arena;
    bool finalized = false;
    // ---- training (the adapter runs and is differentiated every step) ----
    bool train_ready = false, recorded = false;
    std::vector<AdTrainable> trainables;
    long long n_trainable = 0;
    float* theta = nullptr;
    float* grad = nullptr;
    Act rec_u;
    std::vector<AdRec> recs;
    Act bwd_dcur;             // running gradient between the level-wise backward calls
    int bwd_next_level = -1;  // next level mrisr_adapter_backward_level expects (descending)
    std::vector<ConvW*> all_convs() {
        std::vector<ConvW*> v{&conv_in};
        for (auto& b : blocks) {
            if (b.down) v.push_back(&b.down_w);
            if (b.has_in) v.push_back(&b.in_w);
            v.push_back(&b.b1);
            v.push_back(&b.b2);
        }
        return v;
    }
};

template <typename T>
struct AdRunner {
    mrisr_adapter& a;
    hipStream_t st;
    bool dry;
    Act new_act(int B, int H, int W, int C) {
        Act x; x.B = B; x.H = H; x.W = W; x.C = C;
        x.p = a.arena.alloc(x.numel() * sizeof(T));
        if (!x.p) set_error("adapter workspace exhausted");
        return x;
    }
    int conv(const Act& x, const ConvW& cw, int stride, int act, const Act* resid, Act* out) {
        MRISR_REQUIRE(cw.cin == x.C, "adapter conv channel mismatch");
        const int Ho = (x.H - 1) / stride + 1, Wo = (x.W - 1) / stride + 1;
        *out = new_act(x.B, cw.ks == 3 ? Ho : x.H, cw.ks == 3 ? Wo : x.W, cw.cout);
        if (!out->p) return 7;
        GemmArgs g;
        g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
        g.w = cw.w; g.N = cw.cout; g.bias = cw.b; g.act = act;
        if (cw.ks == 3) {
            g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = Ho; g.Wout = Wo; g.stride = stride;
            g.M = x.B * Ho * Wo; g.K = 9 * x.C;
        } else {
            MRISR_REQUIRE(stride == 1, "1x1 conv stride");
            g.M = (int)x.rows(); g.K = x.C;
        }
        if (resid) { g.resid = resid->p; g.ldr = resid->C; }
        g.out = out->p; g.ldo = cw.cout;
        TRY(gemm_choose(g, sizeof(T) == 2));
        if (g.splitk > 1) {
            g.partial = static_cast<float*>(a.arena.alloc((size_t)g.splitk * g.M * g.N * sizeof(float)));
            if (!g.partial) return 7;
        }
        if (dry) return 0;
        return launch_gemm<T>(g, st);
    }
    // ---- backward building blocks ----
    void* alloc(size_t bytes) {
        void* p = a.arena.alloc(bytes);
        if (!p) set_error("adapter workspace exhausted");
        return p;
    }
    int gemm(GemmArgs& g) {
        TRY(gemm_choose(g, sizeof(T) == 2));
        if (g.splitk > 1) {
            g.partial = static_cast<float*>(alloc((size_t)g.splitk * g.M * g.N * sizeof(float)));
            if (!g.partial) return 7;
        }
        if (dry) return 0;
        return launch_gemm<T>(g, st);
    }
    // dX = dgrad(dY) (+ resid);  3x3: conv with the flipped / transposed bank, stride 2 through zero-stuffing;  1x1: dY W
    int conv_dgrad(const Act& dy, const ConvW& cw, int stride, const Act* resid, Act* dx) {
        MRISR_REQUIRE(cw.wd && dy.C == cw.cout, "adapter dgrad weights");
        const int up = stride == 2 ? 1 : 0;
        *dx = new_act(dy.B, dy.H << up, dy.W << up, cw.cin);
        if (!dx->p) return 7;
        GemmArgs g;
        g.a0 = dy.p; g.c0 = cw.cout; g.lda0 = cw.cout;
        g.w = cw.wd; g.N = cw.cin; g.out = dx->p; g.ldo = cw.cin;
        if (cw.ks == 3) {
            g.conv = 1; g.B = dy.B; g.Hin = dy.H; g.Win = dy.W; g.Hout = dy.H << up; g.Wout = dy.W << up; g.stride = 1; g.ups = up; g.zstuff = up;
            g.M = dy.B * g.Hout * g.Wout; g.K = 9 * cw.cout;
        } else {
            g.M = (int)dy.rows(); g.K = cw.cout;
        }
        if (resid) { g.resid = resid->p; g.ldr = resid->C; }
        return gemm(g);
    }
    // dW, db of y = conv(x): one GEMM per tap over the pixel index (operands transposed into pixel-major scratch)
    int conv_wgrad(const Act& x, const Act& dy, const ConvW& cw, int stride) {
        MRISR_REQUIRE(cw.offW >= 0 && a.grad, "adapter gradient vector not bound");
        const int M = (int)dy.rows(), Mpad = (M + 63) / 64 * 64, taps = cw.ks * cw.ks, pad = cw.ks / 2;
        const size_t mk = a.arena.mark();
        // ONE pixel-contraction GEMM per conv: dW[co][tap * Cin + ci] = sum_m dY^T[co][m] * im2col^T[tap * Cin + ci][m]
        // (one transposed im2col launch for all taps, one GEMM, one scatter-add into the PyTorch layout; per-tap GEMMs
        // meant 27 launches per conv)
        T* dyT = static_cast<T*>(alloc((size_t)cw.cout * Mpad * sizeof(T)));
        T* xT = static_cast<T*>(alloc((size_t)taps * cw.cin * Mpad * sizeof(T)));
        float* tmp = static_cast<float*>(alloc((size_t)cw.cout * taps * cw.cin * sizeof(float)));
        if (!dyT || !xT || !tmp) return 7;
        if (!dry) {
            if (Mpad != M) MRISR_CHECK_HIP(hipMemsetAsync(dyT, 0, (size_t)cw.cout * Mpad * sizeof(T), st));
            TRY(launch_transpose<T>(dy.p, dyT, M, cw.cout, cw.cout, Mpad, 0, 0, 1, M, st));
            if (cw.offB >= 0) TRY(launch_colsum<T>(dy.p, a.grad + cw.offB, M, cw.cout, st));
            TRY(launch_im2col_all_T<T>(x.p, xT, x.B, x.H, x.W, x.C, dy.H, dy.W, stride, pad, cw.ks, Mpad, st));
        }
        GemmArgs g;
        g.a0 = dyT; g.c0 = Mpad; g.lda0 = Mpad;
        g.w = xT; g.M = cw.cout; g.N = taps * cw.cin; g.K = Mpad;
        g.out_mode = OUT_F32; g.out = tmp; g.ldo = taps * cw.cin;
        TRY(gemm(g));
        if (!dry) TRY(launch_wgrad_accum_all(tmp, a.grad + cw.offW, cw.cout, cw.cin, taps, st));
        a.arena.release(mk);
        return 0;
    }
    // d_feats: gradients w.r.t. the four feature maps (what mrisr_train_step wrote through mrisr_train_set_intrablock_grads)
    // Blocks k_hi .. k_lo (descending) of the backward; the running gradient lives in a.bwd_dcur between calls, so that the host
    // can cut the pass at level boundaries and start the exchange of a level's finished weight gradients while the lower
    // levels are still being differentiated (mrisr_adapter_backward_level).
    int backward_blocks(const mrisr_tensor* d_feats, int n_feats, int k_hi, int k_lo, bool with_conv_in) {
        MRISR_REQUIRE(a.recorded || dry, "run the adapter forward first");
        MRISR_REQUIRE(n_feats * a.cfg.nums_rb == (int)a.blocks.size(), "one feature gradient per level");
        Act dcur = a.bwd_dcur;
        for (int k = k_hi; k >= k_lo; --k) {
            AdBlock& b = a.blocks[k];
            AdRec& r = a.recs[k];
            if ((k + 1) % a.cfg.nums_rb == 0) {
                const mrisr_tensor& f = d_feats[(k + 1) / a.cfg.nums_rb - 1];
                MRISR_REQUIRE(f.ndim == 4 && f.shape[0] == r.y.B && f.shape[1] == r.y.C && f.shape[2] == r.y.H && f.shape[3] == r.y.W,
                              "adapter feature gradient shape");
                Act gf = new_act(r.y.B, r.y.H, r.y.W, r.y.C);
                if (!gf.p) return 7;
                if (!dry) {
                    if (f.layout == MRISR_NHWC) {
                        MRISR_REQUIRE(f.dtype == a.cfg.compute_dtype, "NHWC feature gradients use the compute dtype");
                        MRISR_CHECK_HIP(hipMemcpyAsync(gf.p, f.data, gf.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
                    } else {
                        TRY(launch_nchw_to_nhwc<T>(f.data, f.dtype, gf.p, gf.B, gf.C, gf.H, gf.W, st));
                    }
                    if (dcur.p) TRY(launch_add_inplace<T>(gf.p, dcur.p, (long long)gf.numel(), st));
                }
                dcur = gf;
            }
            MRISR_REQUIRE(dcur.p, "no gradient reaches the last adapter block");
            // y = block2(relu(block1(x_pre))) + x_pre
            TRY(conv_wgrad(r.hmid, dcur, b.b2, 1));
            Act dh, dx;
            TRY(conv_dgrad(dcur, b.b2, 1, nullptr, &dh));
            if (!dry) TRY(launch_relu_bwd<T>(dh.p, r.hmid.p, dh.p, (long long)dh.numel(), st));
            TRY(conv_wgrad(r.x_pre, dh, b.b1, 1));
            TRY(conv_dgrad(dh, b.b1, 1, &dcur, &dx));
            if (b.has_in) {
                TRY(conv_wgrad(r.x_down, dx, b.in_w, 1));
                Act d2;
                TRY(conv_dgrad(dx, b.in_w, 1, nullptr, &d2));
                dx = d2;
            }
            if (b.down) {
                TRY(conv_wgrad(r.x_in, dx, b.down_w, 2));
                Act d2;
                TRY(conv_dgrad(dx, b.down_w, 2, nullptr, &d2));
                dx = d2;
            }
            dcur = dx;
        }
        a.bwd_dcur = dcur;
        if (with_conv_in) return conv_wgrad(a.rec_u, dcur, a.conv_in, 1);
        return 0;
    }
    int backward(const mrisr_tensor* d_feats, int n_feats) {
        a.bwd_dcur = Act();
        a.bwd_next_level = -1;
        return backward_blocks(d_feats, n_feats, (int)a.blocks.size() - 1, 0, true);
    }
    // one level (nums_rb blocks) of the backward, levels in descending order; level 0 also differentiates conv_in
    int backward_level(const mrisr_tensor* d_feats, int n_feats, int level) {
        const int nlev = (int)a.blocks.size() / a.cfg.nums_rb;
        MRISR_REQUIRE(level >= 0 && level < nlev, "adapter level");
        if (level == nlev - 1) { a.bwd_dcur = Act(); a.bwd_next_level = level; }
        MRISR_REQUIRE(a.bwd_next_level == level, "adapter backward levels must run in descending order, starting at the top level");
        a.bwd_next_level = level - 1;
        return backward_blocks(d_feats, n_feats, (level + 1) * a.cfg.nums_rb - 1, level * a.cfg.nums_rb, level == 0);
    }

    int forward(const mrisr_tensor& x, mrisr_tensor* feats, int n_feats) {
        a.arena.reset();
        const int B = (int)x.shape[0], C = (int)x.shape[1], H = (int)x.shape[2], W = (int)x.shape[3];
        MRISR_REQUIRE(H % 8 == 0 && W % 8 == 0 && C * 64 == a.cfg.cin, "adapter input must be [B, cin/64, 8h, 8w]");
        Act u = new_act(B, H / 8, W / 8, C * 64);
        if (!u.p) return 7;
        if (!dry) TRY(launch_pixel_unshuffle_nchw<T>(x.data, x.dtype, u.p, B, C, H, W, 8, st));
        Act cur;
        TRY(conv(u, a.conv_in, 1, ACT_NONE, nullptr, &cur));
        a.rec_u = u;
        a.recs.assign(a.blocks.size(), AdRec());
        int fi = 0;
        for (size_t k = 0; k < a.blocks.size(); ++k) {
            AdBlock& b = a.blocks[k];
            AdRec& rec = a.recs[k];
            Act y;
            rec.x_in = cur;
            if (b.down) { TRY(conv(cur, b.down_w, 2, ACT_NONE, nullptr, &y)); cur = y; }
            rec.x_down = cur;
            if (b.has_in) { TRY(conv(cur, b.in_w, 1, ACT_NONE, nullptr, &y)); cur = y; }
            rec.x_pre = cur;
            Act hmid;
            TRY(conv(cur, b.b1, 1, ACT_RELU, nullptr, &hmid));
            TRY(conv(hmid, b.b2, 1, ACT_NONE, &cur, &y));
            rec.hmid = hmid;
            rec.y = y;
            cur = y;
            if ((k + 1) % a.cfg.nums_rb == 0) {
                MRISR_REQUIRE(fi < n_feats, "too few feature outputs");
                const mrisr_tensor& f = feats[fi++];
                MRISR_REQUIRE(f.ndim == 4 && f.shape[0] == cur.B && f.shape[1] == cur.C && f.shape[2] == cur.H && f.shape[3] == cur.W,
                              "adapter feature output shape");
                if (!dry) {
                    if (f.layout == MRISR_NHWC) {
                        MRISR_REQUIRE(f.dtype == a.cfg.compute_dtype, "NHWC feature outputs use the compute dtype");
                        MRISR_CHECK_HIP(hipMemcpyAsync(f.data, cur.p, cur.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
                    } else {
                        TRY(launch_nhwc_to_nchw<T>(cur.p, f.data, f.dtype, cur.B, cur.C, cur.H, cur.W, 1.0f, st));
                    }
                }
            }
        }
        return 0;
    }
};

template <typename T>
static int adapter_finalize_t(mrisr_adapter& a, hipStream_t st) {
    a.packed.clear();
    a.blocks.clear();
    int err = 0;
    auto conv = [&](const std::string& name) {
        ConvW c;
        auto it = a.raw.find(name + ".weight");
        if (it == a.raw.end()) { set_error("missing parameter: " + name + ".weight"); err = 3; return c; }
        const RawParam& w = it->second;
        c.cout = (int)w.shape[0]; c.cin = (int)w.shape[1]; c.ks = (int)w.shape[2];
        c.name = name;
        a.packed.emplace_back(new DevBuf());
        if (a.packed.back()->reserve((size_t)w.numel() * sizeof(T), false)) { err = 4; return c; }
        c.w = a.packed.back()->p;
        if (launch_pack_conv3x3<T>(static_cast<const float*>(w.data->p), c.w, c.cout, c.cin, c.ks, st)) err = 5;
        auto ib = a.raw.find(name + ".bias");
        c.b = ib == a.raw.end() ? nullptr : static_cast<const float*>(ib->second.data->p);
        return c;
    };
    a.conv_in = conv("conv_in");
    int k = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < a.cfg.nums_rb; ++j, ++k) {
            AdBlock b;
            const std::string n = "body." + std::to_string(k);
            b.out_c = a.cfg.channels[i];
            b.in_c = (i > 0 && j == 0) ? a.cfg.channels[i - 1] : b.out_c;
            b.down = (i > 0 && j == 0);
            if (b.down) {
                if (!a.cfg.use_conv) { set_error("avg-pool downsample (use_conv=False) is not built"); return 8; }
                b.down_w = conv(n + ".down_opt.op");
            }
            if (a.raw.count(n + ".in_conv.weight")) { b.has_in = true; b.in_w = conv(n + ".in_conv"); }
            else if (b.in_c != b.out_c) { set_error("missing parameter: " + n + ".in_conv.weight"); return 3; }
            if (a.raw.count(n + ".skep.weight")) {
                // reference quirk: sk=False cannot run in the reference either
                set_error("Adapter_XL with sk=False is not runnable in the reference (channel mismatch in skep); use sk=True");
                return 8;
            }
            b.b1 = conv(n + ".block1");
            b.b2 = conv(n + ".block2");
            a.blocks.push_back(b);
        }
    if (err) return err;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    a.finalized = true;
    return 0;
}

// re-pack every conv of the adapter (forward bank, dgrad bank, bias pointer) from the bound trainable vector
template <typename T>
static int adapter_repack_t(mrisr_adapter& a, hipStream_t st) {
    for (ConvW* c : a.all_convs()) {
        const float* w = a.theta + c->offW;
        TRY(launch_pack_conv3x3<T>(w, c->w, c->cout, c->cin, c->ks, st));
        if (c->ks == 3) TRY(launch_pack_conv_dgrad<T>(w, c->wd, c->cout, c->cin, st));
        else TRY(launch_transpose<T>(c->w, c->wd, c->cout, c->cin, c->cin, c->cout, 0, 0, 1, c->cout, st));
        if (c->offB >= 0) c->b = a.theta + c->offB;
    }
    return 0;
}
extern "C" {

int mrisr_adapter_create(const mrisr_adapter_cfg* cfg, mrisr_adapter** out) {
    API_BEGIN
    MRISR_REQUIRE(cfg && out, "null argument");
    MRISR_REQUIRE(cfg->compute_dtype == MRISR_F32 || cfg->compute_dtype == MRISR_BF16, "compute dtype");
    MRISR_REQUIRE(cfg->ksize == 1 || cfg->ksize == 3, "ksize 1 or 3");
    auto* a = new mrisr_adapter();
    a->cfg = *cfg;
    *out = a;
    return 0;
    API_END
}
void mrisr_adapter_destroy(mrisr_adapter* a) { delete a; }
int mrisr_adapter_set_param(mrisr_adapter* a, const char* key, const float* data, const int64_t* shape, int ndim,
                            int is_device) {
    API_BEGIN
    MRISR_REQUIRE(a && key && data, "null argument");
    RawParam rp;
    rp.shape.assign(shape, shape + ndim);
    rp.data = std::make_shared<DevBuf>();
    TRY(rp.data->reserve((size_t)rp.numel() * sizeof(float), false));
    MRISR_CHECK_HIP(hipMemcpy(rp.data->p, data, (size_t)rp.numel() * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    a->raw[key] = rp;
    a->finalized = false;
    return 0;
    API_END
}
int mrisr_adapter_finalize(mrisr_adapter* a, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a, "null handle");
    TRY(gemm_prepare());
    if (a->cfg.compute_dtype == MRISR_F32) return adapter_finalize_t<float>(*a, (hipStream_t)stream);
    return adapter_finalize_t<bf16>(*a, (hipStream_t)stream);
    API_END
}
int mrisr_adapter_forward(mrisr_adapter* a, const mrisr_tensor* x, mrisr_tensor* feats, int n_feats, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a && x && feats && a->finalized, "adapter not finalized / null argument");
    MRISR_REQUIRE(x->ndim == 4 && x->layout == MRISR_NCHW, "adapter input: NCHW image");
    hipStream_t st = (hipStream_t)stream;
    // size the arena with a dry pass (exact), then run
    int rc;
    a->arena.dry = true; a->arena.reset(); a->arena.peak = 0;
    a->recorded = false;
    // training: the backward continues in the same arena (the activations must stay put), so size it for both now
    if (a->cfg.compute_dtype == MRISR_F32) { AdRunner<float> r{*a, st, true}; rc = r.forward(*x, feats, n_feats); if (!rc && a->train_ready) rc = r.backward(feats, n_feats); }
    else { AdRunner<bf16> r{*a, st, true}; rc = r.forward(*x, feats, n_feats); if (!rc && a->train_ready) rc = r.backward(feats, n_feats); }
    a->arena.dry = false;
    if (rc) return rc;
    if (a->arena.peak + 4096 > a->arena.buf.bytes) MRISR_CHECK_HIP(hipStreamSynchronize(st));  // the old buffer may still be in use
    TRY(a->arena.buf.reserve(a->arena.peak + 4096, false));
    if (a->cfg.compute_dtype == MRISR_F32) { AdRunner<float> r{*a, st, false}; rc = r.forward(*x, feats, n_feats); }
    else { AdRunner<bf16> r{*a, st, false}; rc = r.forward(*x, feats, n_feats); }
    a->recorded = rc == 0;
    return rc;
    API_END
}

// ---- T2I-Adapter training: flat f32 trainable / gradient vectors owned by the caller (as for the LoRA adapters) ----
int mrisr_adapter_train_prepare(mrisr_adapter* a, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a && a->finalized, "adapter not finalized");
    if (a->train_ready) return 0;
    a->trainables.clear();
    long long off = 0;
    const size_t es = a->cfg.compute_dtype == MRISR_F32 ? 4 : 2;
    for (ConvW* c : a->all_convs()) {
        MRISR_REQUIRE(c->ks == 1 || c->ks == 3, "adapter conv kernel size");
        c->offW = off;
        a->trainables.push_back({c->name + ".weight", off, (long long)c->cout * c->cin * c->ks * c->ks, {c->cout, c->cin, c->ks, c->ks}});
        off += (long long)c->cout * c->cin * c->ks * c->ks;
        if (c->b) {
            c->offB = off;
            a->trainables.push_back({c->name + ".bias", off, c->cout, {c->cout}});
            off += c->cout;
        }
        a->packed.emplace_back(new DevBuf());
        TRY(a->packed.back()->reserve((size_t)c->cout * c->cin * c->ks * c->ks * es, false));
        c->wd = a->packed.back()->p;
    }
    a->n_trainable = off;
    a->train_ready = true;
    (void)stream;
    return 0;
    API_END
}
int64_t mrisr_adapter_train_num_trainable(const mrisr_adapter* a) { return a && a->train_ready ? (int64_t)a->n_trainable : -1; }
int mrisr_adapter_train_num_tensors(const mrisr_adapter* a) { return a && a->train_ready ? (int)a->trainables.size() : -1; }
int mrisr_adapter_train_tensor_info(const mrisr_adapter* a, int i, const char** key, int64_t* offset, int64_t shape[4], int* ndim) {
    MRISR_REQUIRE(a && a->train_ready && i >= 0 && i < (int)a->trainables.size() && key && offset && shape && ndim, "adapter trainable index");
    const AdTrainable& t = a->trainables[i];
    *key = t.key.c_str();
    *offset = t.offset;
    *ndim = (int)t.shape.size();
    for (size_t k = 0; k < t.shape.size(); ++k) shape[k] = t.shape[k];
    return 0;
}
int mrisr_adapter_train_refresh(mrisr_adapter* a, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a && a->train_ready && a->theta, "bind the adapter's trainable vector first");
    return a->cfg.compute_dtype == MRISR_F32 ? adapter_repack_t<float>(*a, (hipStream_t)stream) : adapter_repack_t<bf16>(*a, (hipStream_t)stream);
    API_END
}
int mrisr_adapter_train_bind(mrisr_adapter* a, float* theta_dev, float* grad_dev, int init_from_model, void* stream) {
    API_BEGIN
    TRY(mrisr_adapter_train_prepare(a, stream));
    MRISR_REQUIRE(theta_dev && grad_dev, "theta / grad device buffers");
    a->theta = theta_dev;
    a->grad = grad_dev;
    if (init_from_model)
        for (auto& t : a->trainables) {
            auto it = a->raw.find(t.key);
            MRISR_REQUIRE(it != a->raw.end() && it->second.numel() == t.numel, "adapter tensor missing from the loaded parameters");
            MRISR_CHECK_HIP(hipMemcpyAsync(theta_dev + t.offset, it->second.data->p, (size_t)t.numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        }
    return mrisr_adapter_train_refresh(a, stream);
    API_END
}
int mrisr_adapter_backward(mrisr_adapter* a, const mrisr_tensor* d_feats, int n_feats, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a && d_feats && a->train_ready && a->grad, "bind the adapter's trainable vector first");
    hipStream_t st = (hipStream_t)stream;
    if (a->cfg.compute_dtype == MRISR_F32) { AdRunner<float> r{*a, st, false}; return r.backward(d_feats, n_feats); }
    AdRunner<bf16> r{*a, st, false};
    return r.backward(d_feats, n_feats);
    API_END
}
int mrisr_adapter_backward_level(mrisr_adapter* a, const mrisr_tensor* d_feats, int n_feats, int level, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(a && d_feats && a->train_ready && a->grad, "bind the adapter's trainable vector first");
    hipStream_t st = (hipStream_t)stream;
    if (a->cfg.compute_dtype == MRISR_F32) { AdRunner<float> r{*a, st, false}; return r.backward_level(d_feats, n_feats, level); }
    AdRunner<bf16> r{*a, st, false};
    return r.backward_level(d_feats, n_feats, level);
    API_END
}
int mrisr_adapter_train_level_range(const mrisr_adapter* ac, int level, int64_t* offset, int64_t* numel) {
    API_BEGIN
    mrisr_adapter* a = const_cast<mrisr_adapter*>(ac);
    MRISR_REQUIRE(a && a->train_ready && offset && numel, "adapter not prepared for training");
    const int rb = a->cfg.nums_rb, nlev = (int)a->blocks.size() / rb;
    MRISR_REQUIRE(level >= 0 && level < nlev, "adapter level");
    // all_convs() order = flat-vector order: conv_in, then per block (down, in_conv, block1, block2): a level is one contiguous range
    auto first_off = [&](AdBlock& b) { return b.down ? b.down_w.offW : (b.has_in ? b.in_w.offW : b.b1.offW); };
    const long long lo = level == 0 ? 0 : first_off(a->blocks[level * rb]);
    const long long hi = level + 1 < nlev ? first_off(a->blocks[(level + 1) * rb]) : a->n_trainable;
    *offset = lo;
    *numel = hi - lo;
    return 0;
    API_END
}

}  // extern "C"

// =================================================================================================
// single-op entry points (parity tests drive the very kernels the models launch)
// =================================================================================================
template <typename T>
__global__ void rows_to_heads_kernel(const T* x, T* dst, int B, int N, int H, int hd, int npad, int dpad, int tr) {
    const long long total = (long long)B * N * H * hd;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int dd = (int)(i % hd);
        const int h = (int)((i / hd) % H);
        const int tok = (int)((i / ((long long)hd * H)) % N);
        const int b = (int)(i / ((long long)hd * H * N));
        const size_t bh = (size_t)b * H + h;
        if (!tr) dst[(bh * npad + tok) * dpad + dd] = x[i];
        else dst[(bh * dpad + dd) * npad + tok] = x[i];
    }
}

template <typename T>
static int op_conv3x3_t(const mrisr_tensor* x, const mrisr_tensor* x2, const float* w, const float* bias, int cout,
                        int stride, int ups, int act, int splitk, mrisr_tensor* y, hipStream_t st) {
    const int B = (int)x->shape[0], C0 = (int)x->shape[1], H = (int)x->shape[2], W = (int)x->shape[3];
    const int C1 = x2 ? (int)x2->shape[1] : 0;
    const int Cin = C0 + C1;
    DevBuf wp, part;
    TRY(wp.reserve((size_t)cout * Cin * 9 * sizeof(T), false));
    TRY(launch_pack_conv3x3<T>(w, wp.p, cout, Cin, 3, st));
    if (!x2 && !ups && Cin % (128 / (int)sizeof(T)) != 0) {
        // fan-in below one K tile (conv_in: 4 channels): the direct kernels (matrix-core conv_in form for bf16, 4 channels)
        DirectConvArgs a;
        a.x = x->data; a.w = wp.p; a.bias = bias; a.y = y->data; a.B = B; a.Hin = H; a.Win = W; a.Cin = Cin;
        a.Hout = (H - 1) / stride + 1; a.Wout = (W - 1) / stride + 1; a.Cout = cout; a.ks = 3; a.stride = stride; a.pad = 1; a.act = act;
        MRISR_REQUIRE(y->shape[1] == cout && y->shape[2] == a.Hout && y->shape[3] == a.Wout, "conv output shape");
        TRY(launch_direct_conv<T>(a, st));
        MRISR_CHECK_HIP(hipStreamSynchronize(st));
        return 0;
    }
    if (ups == 2) {  // the sub-pixel form of `nearest x2 -> conv3x3` (runner.h::upsample_conv): four 2 x 2 parity convs + interleave
        MRISR_REQUIRE(sizeof(T) == 2 && !x2 && stride == 1 && act == ACT_NONE && (4 * Cin) % 64 == 0 && cout % 8 == 0, "sub-pixel upsample conv: bf16, single source");
        MRISR_REQUIRE(y->shape[1] == cout && y->shape[2] == 2 * H && y->shape[3] == 2 * W, "conv output shape");
        DevBuf sp, planes;
        TRY(sp.reserve((size_t)16 * cout * Cin * sizeof(T), false));
        TRY(planes.reserve((size_t)4 * B * H * W * cout * sizeof(T), false));
        TRY(launch_pack_conv_subpix<T>(w, sp.p, cout, Cin, st));
        GemmArgs g;
        g.a0 = x->data; g.c0 = C0; g.lda0 = C0;
        g.conv = 1; g.B = B; g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
        g.kw = 2; g.subpix = 1; g.batch = 4; g.w_bs = (long long)cout * 4 * Cin; g.o_bs = (long long)B * H * W * cout;
        g.w = sp.p; g.M = B * H * W; g.N = cout; g.K = 4 * Cin; g.bias = bias; g.out = planes.p; g.ldo = cout;
        TRY(gemm_choose(g, true));
        TRY(launch_gemm<T>(g, st));
        TRY(launch_subpix_shuffle<T>(planes.p, y->data, B, H, W, cout, st));
        MRISR_CHECK_HIP(hipStreamSynchronize(st));
        return 0;
    }
    GemmArgs g;
    g.a0 = x->data; g.c0 = C0; g.lda0 = C0;
    if (x2) { g.a1 = x2->data; g.c1 = C1; g.lda1 = C1; }
    const int Hc = H << ups, Wc = W << ups;
    g.conv = 1; g.B = B; g.Hin = H; g.Win = W; g.Hout = (Hc - 1) / stride + 1; g.Wout = (Wc - 1) / stride + 1;
    g.stride = stride; g.ups = ups;
    MRISR_REQUIRE(y->shape[1] == cout && y->shape[2] == g.Hout && y->shape[3] == g.Wout, "conv output shape");
    g.w = wp.p; g.M = B * g.Hout * g.Wout; g.N = cout; g.K = 9 * Cin; g.bias = bias; g.act = act;
    g.out = y->data; g.ldo = cout;
    g.splitk = splitk;
    if (splitk <= 0) { g.splitk = 1; TRY(gemm_choose(g, sizeof(T) == 2)); }
    if (g.splitk > 1) {
        TRY(part.reserve((size_t)g.splitk * g.M * g.N * sizeof(float), false));
        g.partial = static_cast<float*>(part.p);
    }
    TRY(launch_gemm<T>(g, st));
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" void mrisr_debug_force_tile(int t);

template <typename T>
static int op_attention_t(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, int H, int flash,
                          mrisr_tensor* out, hipStream_t st) {
    const int B = (int)q->shape[0], N = (int)q->shape[1], C = (int)q->shape[2], Nk = (int)k->shape[1];
    const int hd = C / H;
    constexpr int BK = 128 / (int)sizeof(T);
    const bool use_flash = flash && sizeof(T) == 2;
    const int dpad = round_up(hd, use_flash ? 32 : BK), npad = round_up(N, 64), nkpad = round_up(Nk, 64);
    DevBuf qb, kb, vb, sb, pb;
    TRY(qb.reserve((size_t)B * H * npad * dpad * sizeof(T), true));
    TRY(kb.reserve((size_t)B * H * nkpad * dpad * sizeof(T), true));
    TRY(vb.reserve((size_t)B * H * dpad * nkpad * sizeof(T), true));
    auto conv = [&](const mrisr_tensor* x, void* dst, int n, int np, int tr) {
        hipLaunchKernelGGL(rows_to_heads_kernel<T>, dim3(1024), dim3(256), 0, st, static_cast<const T*>(x->data),
                           static_cast<T*>(dst), B, n, H, hd, np, dpad, tr);
    };
    conv(q, qb.p, N, npad, 0);
    conv(k, kb.p, Nk, nkpad, 0);
    conv(v, vb.p, Nk, nkpad, 1);
    MRISR_CHECK_HIP(hipGetLastError());
    const float scale = 1.0f / sqrtf((float)hd);
    if (use_flash) {
        AttnArgs a;
        a.q = qb.p; a.k = kb.p; a.vt = vb.p; a.out = out->data;
        a.B = B; a.H = H; a.nq = N; a.nk = Nk; a.nkpad = nkpad; a.hd = hd; a.dpad = dpad; a.scale = scale;
        if (flash == 2) {  // fp8 (OCP e4m3) Q K^T and P V
            DevBuf k8, v8, sc;
            TRY(k8.reserve((size_t)B * H * nkpad * dpad, false));
            TRY(v8.reserve((size_t)B * H * nkpad * dpad, false));
            TRY(sc.reserve((size_t)B * H * 4 * sizeof(float), false));
            a.k8 = k8.p; a.vt8 = v8.p; a.f8_scales = static_cast<float*>(sc.p);
            TRY(launch_attention_fp8(a, st));
            MRISR_CHECK_HIP(hipStreamSynchronize(st));
        } else {
            TRY(launch_attention_bf16(a, st));
        }
    } else {
        const int BH = B * H;
        TRY(sb.reserve((size_t)BH * N * nkpad * sizeof(float), false));
        void* P = sb.p;
        if (sizeof(T) == 2) { TRY(pb.reserve((size_t)BH * N * nkpad * sizeof(T), false)); P = pb.p; }
        GemmArgs g;
        g.a0 = qb.p; g.c0 = dpad; g.lda0 = dpad; g.a_bs = (long long)npad * dpad;
        g.w = kb.p; g.w_bs = (long long)nkpad * dpad; g.M = N; g.N = nkpad; g.K = dpad; g.batch = BH; g.alpha = scale;
        g.out_mode = OUT_F32; g.out = sb.p; g.ldo = nkpad; g.o_bs = (long long)N * nkpad;
        TRY(launch_gemm<T>(g, st));
        TRY(launch_softmax_rows<T>(static_cast<const float*>(sb.p), nkpad, P, nkpad, (long long)BH * N, Nk, st));
        GemmArgs o;
        o.a0 = P; o.c0 = nkpad; o.lda0 = nkpad; o.a_bs = (long long)N * nkpad;
        o.w = vb.p; o.w_bs = (long long)dpad * nkpad; o.M = N; o.N = hd; o.K = nkpad; o.batch = BH;
        o.heads = H; o.o_bs = (long long)N * C; o.o_hs = hd; o.out = out->data; o.ldo = C;
        TRY(launch_gemm<T>(o, st));
    }
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}


extern "C" {

static int op_dtype_ok(const mrisr_tensor* x) {
    MRISR_REQUIRE(x && (x->dtype == MRISR_F32 || x->dtype == MRISR_BF16), "op tensors: f32 or bf16");
    return 0;
}

int mrisr_op_conv3x3(const mrisr_tensor* x, const mrisr_tensor* x2, const float* w_oihw_dev, const float* bias_dev,
                     int cout, int stride, int upsample, int act, int splitk, int tile, mrisr_tensor* y,
                     void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    TRY(gemm_prepare());
    MRISR_REQUIRE(x->layout == MRISR_NHWC && y && y->layout == MRISR_NHWC && y->dtype == x->dtype, "NHWC in/out, same dtype");
    mrisr_debug_force_tile(tile);
    int rc = x->dtype == MRISR_F32
                 ? op_conv3x3_t<float>(x, x2, w_oihw_dev, bias_dev, cout, stride, upsample, act, splitk, y, (hipStream_t)stream)
                 : op_conv3x3_t<bf16>(x, x2, w_oihw_dev, bias_dev, cout, stride, upsample, act, splitk, y, (hipStream_t)stream);
    mrisr_debug_force_tile(0);
    return rc;
    API_END
}

int mrisr_op_linear(const mrisr_tensor* x, const float* w_dev, const float* bias_dev, int n, int act, int splitk,
                    int tile, mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    TRY(gemm_prepare());
    MRISR_REQUIRE(x->ndim == 2 && y && y->ndim == 2 && y->dtype == x->dtype, "rows in/out");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)x->shape[0], K = (int)x->shape[1];
    const bool f32 = x->dtype == MRISR_F32;
    const int esz = f32 ? 4 : 2;
    DevBuf wp, part, bp;
    TRY(wp.reserve((size_t)n * K * esz, false));
    const bool geglu = act == ACT_GEGLU;
    if (f32) TRY(launch_pack_rows<float>(w_dev, n, K, wp.p, K, 0, 0, geglu ? 1 : 0, n / 2, 1.0f, st));
    else TRY(launch_pack_rows<bf16>(w_dev, n, K, wp.p, K, 0, 0, geglu ? 1 : 0, n / 2, 1.0f, st));
    const float* bias = bias_dev;
    if (geglu && bias_dev) {
        TRY(bp.reserve((size_t)n * sizeof(float), false));
        TRY(launch_pack_bias_geglu(bias_dev, static_cast<float*>(bp.p), n / 2, st));
        bias = static_cast<const float*>(bp.p);
    }
    GemmArgs g;
    g.a0 = x->data; g.c0 = K; g.lda0 = K; g.w = wp.p; g.M = M; g.N = n; g.K = K; g.bias = bias; g.act = act;
    g.out = y->data; g.ldo = (int)y->shape[1];
    g.splitk = splitk;
    if (splitk <= 0) { g.splitk = 1; mrisr_debug_force_tile(tile); TRY(gemm_choose(g, !f32)); }
    if (g.splitk > 1) {
        TRY(part.reserve((size_t)g.splitk * M * n * sizeof(float), false));
        g.partial = static_cast<float*>(part.p);
    }
    mrisr_debug_force_tile(tile);
    int rc = f32 ? launch_gemm<float>(g, st) : launch_gemm<bf16>(g, st);
    mrisr_debug_force_tile(0);
    if (rc) return rc;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
    API_END
}

int mrisr_op_ln_linear(const mrisr_tensor* x, const float* gamma_dev, const float* beta_dev, const float* w_dev, const float* bias_dev,
                       int n, int act, mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    TRY(gemm_prepare());
    MRISR_REQUIRE(x->ndim == 2 && y && y->ndim == 2 && y->dtype == x->dtype && x->dtype == MRISR_BF16, "bf16 rows in/out");
    MRISR_REQUIRE(gamma_dev && beta_dev && w_dev, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)x->shape[0], K = (int)x->shape[1];
    DevBuf wp, bp;
    TRY(wp.reserve((size_t)n * K * 2, false));
    const bool geglu = act == ACT_GEGLU;
    TRY(launch_pack_rows<bf16>(w_dev, n, K, wp.p, K, 0, 0, geglu ? 1 : 0, n / 2, 1.0f, st));
    const float* bias = bias_dev;
    if (geglu && bias_dev) {
        TRY(bp.reserve((size_t)n * sizeof(float), false));
        TRY(launch_pack_bias_geglu(bias_dev, static_cast<float*>(bp.p), n / 2, st));
        bias = static_cast<const float*>(bp.p);
    }
    GemmArgs g;
    g.a0 = x->data; g.c0 = K; g.lda0 = K; g.w = wp.p; g.M = M; g.N = n; g.K = K; g.bias = bias; g.act = act;
    g.out = y->data; g.ldo = (int)y->shape[1];
    MRISR_REQUIRE(gemm_rp_tile(g) != 0, "LayerNorm prologue: the row-panel kernel does not take this shape (K = 320 / 640, N % 16 == 0)");
    g.ln_gamma = gamma_dev; g.ln_beta = beta_dev; g.ln_eps = 1e-5f;
    TRY(gemm_choose(g, true));
    TRY(launch_gemm<bf16>(g, st));
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
    API_END
}

int mrisr_op_mlp(const mrisr_tensor* x, const float* gamma_dev, const float* beta_dev, const float* w1_dev, const float* b1_dev,
                 const float* w2_dev, const float* b2_dev, int hidden, int residual, mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    TRY(gemm_prepare());
    MRISR_REQUIRE(x->ndim == 2 && y && y->ndim == 2 && y->dtype == x->dtype && x->dtype == MRISR_BF16, "bf16 rows in/out");
    MRISR_REQUIRE(gamma_dev && beta_dev && w1_dev && w2_dev, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)x->shape[0], C = (int)x->shape[1], N2 = (int)y->shape[1];
    MRISR_REQUIRE(y->shape[0] == M && mlp_fused_ok(C, hidden, N2), "fused feed-forward: C = 320 rows in and out, hidden % 32 == 0");
    DevBuf w1p, b1p, w2b, w2p;
    TRY(w1p.reserve((size_t)2 * hidden * C * 2, false));
    TRY(w2b.reserve((size_t)N2 * hidden * 2, false));
    TRY(w2p.reserve((size_t)N2 * hidden * 2, false));
    TRY(launch_pack_rows<bf16>(w1_dev, 2 * hidden, C, w1p.p, C, 0, 0, 1, hidden, 1.0f, st));
    TRY(launch_pack_rows<bf16>(w2_dev, N2, hidden, w2b.p, hidden, 0, 0, 0, 0, 1.0f, st));
    TRY(launch_pack_mlp_w2(w2b.p, w2p.p, N2, hidden, st));
    const float* b1 = nullptr;
    if (b1_dev) {
        TRY(b1p.reserve((size_t)2 * hidden * sizeof(float), false));
        TRY(launch_pack_bias_geglu(b1_dev, static_cast<float*>(b1p.p), hidden, st));
        b1 = static_cast<const float*>(b1p.p);
    }
    MlpArgs a;
    a.x = x->data; a.ldx = C; a.M = M; a.ln_gamma = gamma_dev; a.ln_beta = beta_dev; a.ln_eps = 1e-5f;
    a.w1 = w1p.p; a.b1 = b1; a.w2p = w2p.p; a.b2 = b2_dev;
    a.resid = residual ? x->data : nullptr; a.ldr = C; a.out = y->data; a.ldo = N2; a.C = C; a.H = hidden; a.N2 = N2;
    TRY(launch_mlp_fused(a, st));
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
    API_END
}

int mrisr_op_linear_fp8(const mrisr_tensor* x, const float* gamma_dev, const float* beta_dev, const float* w_dev, const float* bias_dev,
                        int n, int act, mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    TRY(gemm_prepare());
    MRISR_REQUIRE(x->ndim == 2 && y && y->ndim == 2 && y->dtype == x->dtype && x->dtype == MRISR_BF16 && w_dev, "bf16 rows in/out");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)x->shape[0], K = (int)x->shape[1];
    DevBuf wp, w8, ws, bp;
    TRY(wp.reserve((size_t)n * K * 2, false));
    TRY(w8.reserve((size_t)n * K, false));
    TRY(ws.reserve((size_t)n * sizeof(float), false));
    const bool geglu = act == ACT_GEGLU;
    TRY(launch_pack_rows<bf16>(w_dev, n, K, wp.p, K, 0, 0, geglu ? 1 : 0, n / 2, 1.0f, st));
    TRY(launch_quant_rows_fp8(wp.p, n, K, w8.p, static_cast<float*>(ws.p), st));
    const float* bias = bias_dev;
    if (geglu && bias_dev) {
        TRY(bp.reserve((size_t)n * sizeof(float), false));
        TRY(launch_pack_bias_geglu(bias_dev, static_cast<float*>(bp.p), n / 2, st));
        bias = static_cast<const float*>(bp.p);
    }
    GemmArgs g;
    g.a0 = x->data; g.c0 = K; g.lda0 = K; g.w = wp.p; g.M = M; g.N = n; g.K = K; g.bias = bias; g.act = act;
    g.out = y->data; g.ldo = (int)y->shape[1];
    g.w8 = w8.p; g.w_scale = static_cast<const float*>(ws.p);
    MRISR_REQUIRE(gemm_rp_tile(g) != 0, "fp8 operands: the row-panel kernel does not take this shape (K = 320 / 640, N % 16 == 0)");
    if (gamma_dev) { g.ln_gamma = gamma_dev; g.ln_beta = beta_dev; g.ln_eps = 1e-5f; }
    TRY(gemm_choose(g, true));
    TRY(launch_gemm<bf16>(g, st));
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
    API_END
}

int mrisr_op_groupnorm(const mrisr_tensor* x, const mrisr_tensor* x2, const float* gamma_dev, const float* beta_dev,
                       int groups, float eps, int silu, mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    MRISR_REQUIRE(x->layout == MRISR_NHWC && y && y->dtype == x->dtype, "NHWC in/out, same dtype");
    hipStream_t st = (hipStream_t)stream;
    GroupNormArgs a;
    a.x0 = x->data; a.c0 = (int)x->shape[1];
    if (x2) { a.x1 = x2->data; a.c1 = (int)x2->shape[1]; }
    a.B = (int)x->shape[0]; a.HW = (int)(x->shape[2] * x->shape[3]); a.groups = groups; a.eps = eps;
    a.gamma = gamma_dev; a.beta = beta_dev; a.silu = silu; a.y = y->data;
    a.nsplit = groupnorm_nsplit(a.B, a.HW);
    DevBuf part;
    TRY(part.reserve((size_t)a.B * a.nsplit * groups * 2 * sizeof(float), false));
    a.partial = static_cast<float*>(part.p);
    int rc = x->dtype == MRISR_F32 ? launch_groupnorm<float>(a, st) : launch_groupnorm<bf16>(a, st);
    if (rc) return rc;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
    API_END
}

int mrisr_op_layernorm(const mrisr_tensor* x, const float* gamma_dev, const float* beta_dev, float eps,
                       mrisr_tensor* y, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(x));
    MRISR_REQUIRE(x->ndim == 2 && y && y->dtype == x->dtype, "rows in/out");
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)x->shape[0], C = (int)x->shape[1];
    return x->dtype == MRISR_F32 ? launch_layernorm<float>(x->data, y->data, gamma_dev, beta_dev, M, C, eps, st)
                                 : launch_layernorm<bf16>(x->data, y->data, gamma_dev, beta_dev, M, C, eps, st);
    API_END
}

int mrisr_op_attention(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, int heads, int flash,
                       mrisr_tensor* out, void* stream) {
    API_BEGIN
    TRY(op_dtype_ok(q));
    TRY(gemm_prepare());
    MRISR_REQUIRE(q->ndim == 3 && k && v && out && k->dtype == q->dtype && v->dtype == q->dtype && out->dtype == q->dtype,
                  "q,k,v,out: [B,N,C] same dtype");
    return q->dtype == MRISR_F32 ? op_attention_t<float>(q, k, v, heads, flash, out, (hipStream_t)stream)
                                 : op_attention_t<bf16>(q, k, v, heads, flash, out, (hipStream_t)stream);
    API_END
}

}  // extern "C"

// flash attention forward (with log-sum-exp) + backward on bf16 token rows: the kernels the fine-tuning step runs
static int op_attention_bwd_bf16(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, const mrisr_tensor* dout, int H,
                                 mrisr_tensor* dq, mrisr_tensor* dk, mrisr_tensor* dv, hipStream_t st) {
    typedef bf16 T;
    const int B = (int)q->shape[0], N = (int)q->shape[1], C = (int)q->shape[2], Nk = (int)k->shape[1];
    const int hd = C / H, BH = B * H;
    const int dpad = round_up(hd, 32), npad = round_up(N, 64), nkpad = round_up(Nk, 64);
    const size_t qsz = (size_t)BH * npad * dpad * sizeof(T), ksz = (size_t)BH * nkpad * dpad * sizeof(T);
    DevBuf qb, kb, vtb, vb, ktb, qtb, doh, doht, ob, lse, dsum;
    TRY(qb.reserve(qsz, true)); TRY(kb.reserve(ksz, true)); TRY(vtb.reserve(ksz, true)); TRY(vb.reserve(ksz, true));
    TRY(ktb.reserve(ksz, true)); TRY(qtb.reserve(qsz, true)); TRY(doh.reserve(qsz, true)); TRY(doht.reserve(qsz, true));
    TRY(ob.reserve((size_t)B * N * C * sizeof(T), false));
    TRY(lse.reserve((size_t)BH * npad * sizeof(float), true)); TRY(dsum.reserve((size_t)BH * npad * sizeof(float), true));
    auto conv = [&](const mrisr_tensor* x, void* dst, int n, int np, int tr) {
        hipLaunchKernelGGL(rows_to_heads_kernel<T>, dim3(1024), dim3(256), 0, st, static_cast<const T*>(x->data), static_cast<T*>(dst), B,
                           n, H, hd, np, dpad, tr);
    };
    conv(q, qb.p, N, npad, 0);
    conv(k, kb.p, Nk, nkpad, 0);
    conv(v, vtb.p, Nk, nkpad, 1);
    MRISR_CHECK_HIP(hipGetLastError());
    const float scale = 1.0f / sqrtf((float)hd);
    AttnArgs a;
    a.q = qb.p; a.k = kb.p; a.vt = vtb.p; a.out = ob.p;
    a.B = B; a.H = H; a.nq = N; a.nk = Nk; a.nkpad = nkpad; a.hd = hd; a.dpad = dpad; a.scale = scale;
    a.lse = static_cast<float*>(lse.p);
    TRY(launch_attention_bf16(a, st));
    TRY(launch_attention_bwd_prep(dout->data, ob.p, doh.p, static_cast<float*>(dsum.p), B, N, H, hd, npad, dpad, st));
    TRY(launch_transpose<T>(vtb.p, vb.p, dpad, nkpad, nkpad, dpad, (long long)dpad * nkpad, (long long)nkpad * dpad, BH, dpad, st));
    TRY(launch_transpose<T>(kb.p, ktb.p, nkpad, dpad, dpad, nkpad, (long long)nkpad * dpad, (long long)dpad * nkpad, BH, nkpad, st));
    TRY(launch_transpose<T>(qb.p, qtb.p, npad, dpad, dpad, npad, (long long)npad * dpad, (long long)dpad * npad, BH, npad, st));
    TRY(launch_transpose<T>(doh.p, doht.p, npad, dpad, dpad, npad, (long long)npad * dpad, (long long)dpad * npad, BH, npad, st));
    AttnBwdArgs g;
    g.q = qb.p; g.k = kb.p; g.v = vb.p; g.doh = doh.p; g.qt = qtb.p; g.kt = ktb.p; g.doht = doht.p;
    g.lse = static_cast<const float*>(lse.p); g.dsum = static_cast<const float*>(dsum.p);
    g.dq = dq->data; g.ldq = C; g.dk = dk->data; g.dv = dv->data; g.ldkv = C;
    g.B = B; g.H = H; g.nq = N; g.nk = Nk; g.npad = npad; g.nkpad = nkpad; g.hd = hd; g.dpad = dpad; g.scale = scale;
    TRY(launch_attention_bwd_bf16(g, st));
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int mrisr_op_attention_bwd(const mrisr_tensor* q, const mrisr_tensor* k, const mrisr_tensor* v, const mrisr_tensor* dout,
                                      int heads, mrisr_tensor* dq, mrisr_tensor* dk, mrisr_tensor* dv, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(q && k && v && dout && dq && dk && dv && q->ndim == 3 && q->dtype == MRISR_BF16 && k->dtype == MRISR_BF16 &&
                      v->dtype == MRISR_BF16 && dout->dtype == MRISR_BF16 && dq->dtype == MRISR_BF16 && dk->dtype == MRISR_BF16 &&
                      dv->dtype == MRISR_BF16,
                  "attention backward: bf16 [B,N,C] tensors");
    MRISR_REQUIRE(heads >= 1 && q->shape[2] % heads == 0 && (q->shape[2] / heads) % 4 == 0, "head dim must be a multiple of 4");
    return op_attention_bwd_bf16(q, k, v, dout, heads, dq, dk, dv, (hipStream_t)stream);
    API_END
}

// =================================================================================================
// GEMM micro-benchmark (tools/gemm_sweep.py): times one implicit-GEMM shape with a forced tile / split-K on
// pseudo-random operands (zero-filled operands would flatter the clock; guide rule 25).
// =================================================================================================
__global__ void fill_random_bf16_kernel(bf16* p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (bf16)(((float)(x & 0xFFFF) / 32768.0f - 1.0f) * 0.5f);
    }
}
extern "C" void mrisr_debug_force_tile(int t);
extern "C" int mrisr_bench_mlp(int M, int hidden, int iters, float* ms_out) {
    API_BEGIN
    TRY(gemm_prepare());
    hipStream_t st = nullptr;
    const int C = 320;
    DevBuf x, w1, w2, o, b1, b2, gb;
    TRY(x.reserve((size_t)M * C * 2, false));
    TRY(o.reserve((size_t)M * C * 2, false));
    TRY(w1.reserve((size_t)2 * hidden * C * 2, false));
    TRY(w2.reserve((size_t)C * hidden * 2, false));
    TRY(b1.reserve((size_t)2 * hidden * 4, true));
    TRY(b2.reserve((size_t)C * 4, true));
    TRY(gb.reserve((size_t)C * 4, true));
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)x.p, (long long)M * C, 1u);
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)w1.p, (long long)2 * hidden * C, 2u);
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)w2.p, (long long)C * hidden, 3u);
    MlpArgs a;
    a.x = x.p; a.ldx = C; a.M = M; a.ln_gamma = (const float*)gb.p; a.ln_beta = (const float*)gb.p;
    a.w1 = w1.p; a.b1 = (const float*)b1.p; a.w2p = w2.p; a.b2 = (const float*)b2.p;
    a.resid = x.p; a.ldr = C; a.out = o.p; a.ldo = C; a.C = C; a.H = hidden; a.N2 = C;
    for (int i = 0; i < 2; ++i) TRY(launch_mlp_fused(a, st));
    hipEvent_t e0, e1;
    MRISR_CHECK_HIP(hipEventCreate(&e0));
    MRISR_CHECK_HIP(hipEventCreate(&e1));
    MRISR_CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) (void)launch_mlp_fused(a, st);
    MRISR_CHECK_HIP(hipEventRecord(e1, st));
    MRISR_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MRISR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    API_END
}

extern "C" int mrisr_bench_gemm(int M, int N, int K, int conv, int B, int H, int W, int stride, int ups, int c1,
                                int tile, int splitk, int iters, float* ms_out) {
    API_BEGIN
    TRY(gemm_prepare());
    hipStream_t st = nullptr;
    GemmArgs g;
    const int Cin = conv ? K / 9 : K;
    const int c0 = Cin - c1;
    DevBuf a0, a1, wb, ob, part, bias;
    const long long a_rows = conv ? (long long)B * H * W : M;
    TRY(a0.reserve((size_t)a_rows * c0 * 2, false));
    if (c1) TRY(a1.reserve((size_t)a_rows * c1 * 2, false));
    TRY(wb.reserve((size_t)N * K * 2, false));
    TRY(ob.reserve((size_t)M * N * 2, false));
    TRY(bias.reserve((size_t)N * 4, true));
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)a0.p, a_rows * c0, 1u);
    if (c1) hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)a1.p, a_rows * c1, 2u);
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(2048), dim3(256), 0, st, (bf16*)wb.p, (long long)N * K, 3u);
    g.a0 = a0.p; g.c0 = c0; g.lda0 = c0;
    if (c1) { g.a1 = a1.p; g.c1 = c1; g.lda1 = c1; }
    if (conv) {
        const int Hc = H << ups, Wc = W << ups;
        g.conv = 1; g.B = B; g.Hin = H; g.Win = W; g.Hout = (Hc - 1) / stride + 1; g.Wout = (Wc - 1) / stride + 1;
        g.stride = stride; g.ups = ups;
        MRISR_REQUIRE(M == B * g.Hout * g.Wout, "bench conv M");
    }
    g.w = wb.p; g.M = M; g.N = N; g.K = K; g.bias = (const float*)bias.p; g.out = ob.p; g.ldo = N;
    if (const char* e = getenv("MRISR_BENCH_NOSTORE")) { if (e[0] == '1') g.out_mode = OUT_NONE; }  // experiment: epilogue without the store
    g.splitk = splitk;
    if (splitk <= 0) { g.splitk = 1; mrisr_debug_force_tile(tile); TRY(gemm_choose(g, true)); }
    if (g.splitk > 1) {
        TRY(part.reserve((size_t)g.splitk * M * N * 4, false));
        g.partial = (float*)part.p;
    }
    mrisr_debug_force_tile(tile);
    for (int i = 0; i < 2; ++i) { int rc = launch_gemm<bf16>(g, st); if (rc) { mrisr_debug_force_tile(0); return rc; } }
    hipEvent_t e0, e1;
    MRISR_CHECK_HIP(hipEventCreate(&e0));
    MRISR_CHECK_HIP(hipEventCreate(&e1));
    MRISR_CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) (void)launch_gemm<bf16>(g, st);
    MRISR_CHECK_HIP(hipEventRecord(e1, st));
    MRISR_CHECK_HIP(hipEventSynchronize(e1));
    mrisr_debug_force_tile(0);
    float ms = 0.f;
    MRISR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return g.splitk * 1000 == 0 ? 0 : 0;
    API_END
}

// ================================================================================================
// LoRA fine-tuning step (train.hip)
// ================================================================================================
extern "C" {

int mrisr_train_prepare(mrisr_model* m, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    TRY(gemm_prepare());
    return m->train_prepare((hipStream_t)stream);
    API_END
}
int64_t mrisr_train_num_trainable(const mrisr_model* m) { return m && m->train_ready ? (int64_t)m->n_trainable : -1; }
int mrisr_train_num_tensors(const mrisr_model* m) { return m && m->train_ready ? (int)m->trainables.size() : -1; }
int mrisr_train_tensor_info(const mrisr_model* m, int i, const char** key, int64_t* offset, int64_t shape[2]) {
    MRISR_REQUIRE(m && m->train_ready, "call mrisr_train_prepare first");
    MRISR_REQUIRE(i >= 0 && i < (int)m->trainables.size() && key && offset && shape, "trainable tensor index");
    const Model::Trainable& t = m->trainables[i];
    *key = t.key.c_str();
    *offset = t.offset;
    shape[0] = t.rows;
    shape[1] = t.cols;
    return 0;
}
int mrisr_train_bind(mrisr_model* m, float* theta_dev, float* grad_dev, int init_from_model, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    hipStream_t st = (hipStream_t)stream;
    TRY(m->train_bind(theta_dev, grad_dev, st));
    if (init_from_model) {
        for (auto& t : m->trainables) {
            const RawParam* r = m->find(t.key);
            MRISR_REQUIRE(r && r->numel() == t.numel, "adapter tensor missing from the loaded parameters");
            MRISR_CHECK_HIP(hipMemcpyAsync(theta_dev + t.offset, r->data->p, (size_t)t.numel * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    return m->lora_refresh(st);
    API_END
}
int mrisr_train_refresh(mrisr_model* m, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->lora_refresh((hipStream_t)stream);
    API_END
}
int mrisr_train_step(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                     const mrisr_tensor* intrablock, int n_intrablock, const mrisr_tensor* target, float* loss_dev,
                     mrisr_tensor* pred_out, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->train_step(sample, timestep, ehs, intrablock, n_intrablock, target, loss_dev, pred_out, (hipStream_t)stream);
    API_END
}
int mrisr_train_set_intrablock_grads(mrisr_model* m, const mrisr_tensor* grads, int n) {
    MRISR_REQUIRE(m && n >= 0 && n <= 4 && (n == 0 || grads), "feature-gradient outputs: 0..4 tensors");
    m->d_intra.assign(grads, grads + n);
    return 0;
}
// ---- full-parameter training of a ControlNet handle ----
int mrisr_controlnet_train_prepare(mrisr_model* m, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->full_train_prepare((hipStream_t)stream);
    API_END
}
int64_t mrisr_controlnet_train_num_trainable(const mrisr_model* m) { return m ? m->n_full : 0; }
int mrisr_controlnet_train_num_tensors(const mrisr_model* m) { return m ? (int)m->full_trainables.size() : 0; }
int mrisr_controlnet_train_tensor_info(const mrisr_model* m, int i, const char** key, int64_t* offset, int64_t* numel, int* differentiated) {
    MRISR_REQUIRE(m && i >= 0 && i < (int)m->full_trainables.size() && key && offset && numel, "tensor index");
    const auto& t = m->full_trainables[i];
    *key = t.key.c_str();
    *offset = t.offset;
    *numel = t.numel;
    if (differentiated) *differentiated = std::find(m->full_unsupported.begin(), m->full_unsupported.end(), t.key) == m->full_unsupported.end() ? 1 : 0;
    return 0;
}
int mrisr_controlnet_train_bind(mrisr_model* m, float* theta_dev, float* grad_dev, int init_from_model, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->full_train_bind(theta_dev, grad_dev, init_from_model, (hipStream_t)stream);
    API_END
}
int mrisr_controlnet_train_refresh(mrisr_model* m, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->full_train_refresh((hipStream_t)stream);
    API_END
}
int mrisr_controlnet_train_forward(mrisr_model* m, const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                                   const mrisr_tensor* cond, float conditioning_scale, mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out,
                                   void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->controlnet_train_forward(sample, timestep, ehs, cond, conditioning_scale, down_out, n_down, mid_out, (hipStream_t)stream);
    API_END
}
int mrisr_controlnet_train_backward(mrisr_model* m, const mrisr_tensor* d_down, int n_down, const mrisr_tensor* d_mid, float conditioning_scale,
                                    void* stream) {
    API_BEGIN
    MRISR_REQUIRE(m, "null handle");
    return m->controlnet_train_backward(d_down, n_down, d_mid, conditioning_scale, (hipStream_t)stream);
    API_END
}
int mrisr_train_set_controlnet_residuals(mrisr_model* m, const mrisr_tensor* down, const mrisr_tensor* d_down, int n_down,
                                         const mrisr_tensor* mid, const mrisr_tensor* d_mid) {
    MRISR_REQUIRE(m && n_down >= 0 && n_down <= 16 && (n_down == 0 || down), "ControlNet residuals of the training step: 0..16 tensors");
    m->tr_down.assign(down, down + n_down);
    if (d_down) m->d_tr_down.assign(d_down, d_down + n_down); else m->d_tr_down.clear();
    m->has_tr_mid = mid != nullptr;
    m->tr_mid = mid ? *mid : mrisr_tensor{};
    m->d_tr_mid = d_mid ? *d_mid : mrisr_tensor{};
    m->train_ws_key.clear();  // the training workspace is planned per residual configuration
    return 0;
}
int mrisr_optim_sumsq(const float* g_dev, int64_t n, float* out_dev, void* stream) {
    MRISR_REQUIRE(g_dev && out_dev && n >= 0, "sumsq arguments");
    return launch_sumsq(g_dev, (long long)n, out_dev, (hipStream_t)stream);
}
int mrisr_optim_ema(float* ema_dev, const float* theta_dev, int64_t n, float decay, void* stream) {
    MRISR_REQUIRE(ema_dev && theta_dev && n >= 0 && decay >= 0.f && decay <= 1.f, "ema arguments");
    return launch_ema(ema_dev, theta_dev, (long long)n, decay, (hipStream_t)stream);
}
int mrisr_optim_adamw(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, const float* sumsq_dev,
                      float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int step, void* stream) {
    MRISR_REQUIRE(p_dev && g_dev && m_dev && v_dev && n >= 0 && step >= 1, "adamw arguments");
    MRISR_REQUIRE(max_norm <= 0.f || sumsq_dev, "clipping needs the squared gradient norm");
    return launch_adamw(p_dev, g_dev, m_dev, v_dev, (long long)n, sumsq_dev, grad_scale, max_norm, lr, beta1, beta2, eps,
                        weight_decay, step, (hipStream_t)stream);
}

}  // extern "C"
