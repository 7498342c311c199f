// AutoencoderKL (SD-1.5 VAE) - the step either side of the sampling loop (reference call sites
// src/adapters/res_srdiff.py:49-50 `vae.encode(lr).latent_dist.sample() * scaling_factor` and :107-110
// `vae.decode(latents / scaling_factor).sample`).  The arithmetic is diffusers' (un-vendored); restated from the
// published architecture under its state-dict key names.  Same kernel classes as the UNet (implicit-GEMM convs, GroupNorm,
// one single-head attention per mid block), driven through the Runner helpers over this handle's own arena.
#include <cstring>

#include "model.h"
#include "runner.h"

using namespace mrisr;

#define API_BEGIN try {
#define API_END                                              \
    }                                                        \
    catch (const std::exception& e) {                        \
        set_error(std::string("exception: ") + e.what());    \
        return 99;                                           \
    }

namespace {

struct VRes {
    NormW n1, n2;
    ConvW c1, c2;
    bool has_sc = false;
    LinW sc;
    int cin = 0, cout = 0;
};
struct VAttn {
    NormW gn;
    LinW qkv, out;
    int C = 0;
};
struct VMid {
    VRes r0, r1;
    VAttn at;
};
struct VLevel {
    std::vector<VRes> res;
    bool has_resample = false;
    ConvW resample;  // encoder: stride-2 conv (asymmetric pad); decoder: conv after nearest x2
};

}  // namespace

struct mrisr_vae {
    mrisr_vae_cfg cfg{};
    Model ctx;  // parameter store + arena + GroupNorm configuration for the Runner helpers (holds no UNet)
    ConvW enc_in, enc_out, quant, post_quant, dec_in, dec_out;
    NormW enc_norm, dec_norm;
    std::vector<VLevel> enc, dec;
    VMid enc_mid, dec_mid;
    bool finalized = false;
    std::string ws_key;
};

namespace {

template <typename T>
struct VPacker {
    mrisr_vae& v;
    hipStream_t st;
    int err = 0;
    const RawParam* need(const std::string& k) {
        const RawParam* r = v.ctx.find(k);
        if (!r && !err) { set_error("missing parameter: " + k); err = 3; }
        return r;
    }
    NormW norm(const std::string& name) {
        NormW n;
        const RawParam *g = need(name + ".weight"), *b = need(name + ".bias");
        if (g && b) { n.g = static_cast<const float*>(g->data->p); n.b = static_cast<const float*>(b->data->p); n.c = (int)g->shape[0]; }
        return n;
    }
    // cout_pad: output channels rounded up (zero filters) so that the implicit GEMM's N % 4 == 0 holds (conv_out: 3 -> 4)
    ConvW conv(const std::string& name, int cout_pad = 0) {
        ConvW c;
        const RawParam* w = need(name + ".weight");
        if (!w) return c;
        const int cout = (int)w->shape[0];
        c.name = name; c.cin = (int)w->shape[1]; c.ks = (int)w->shape[2];
        c.cout = cout_pad > cout ? cout_pad : cout;
        const size_t per = (size_t)c.cin * c.ks * c.ks;
        c.w = v.ctx.new_packed((size_t)c.cout * per * sizeof(T), c.cout != cout);
        if (!c.w) { err = 4; return c; }
        if (launch_pack_conv3x3<T>(static_cast<const float*>(w->data->p), c.w, cout, c.cin, c.ks, st)) err = 5;
        if (const RawParam* b = v.ctx.find(name + ".bias")) {
            float* bp = static_cast<float*>(v.ctx.new_packed((size_t)c.cout * sizeof(float), true));
            if (!bp) { err = 4; return c; }
            if (hipMemcpyAsync(bp, b->data->p, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) err = 5;
            c.b = bp;
        }
        return c;
    }
    // Linear (or 1x1 conv) modules fused along N
    LinW linear(const std::vector<std::string>& mods) {
        LinW l;
        int ntot = 0, k = 0;
        for (auto& m : mods) {
            const RawParam* w = need(m + ".weight");
            if (!w) return l;
            ntot += (int)w->shape[0];
            k = (int)w->shape[1];
        }
        l.n = ntot; l.k = k;
        l.w = v.ctx.new_packed((size_t)ntot * k * sizeof(T), false);
        float* bias = static_cast<float*>(v.ctx.new_packed((size_t)ntot * sizeof(float), true));
        if (!l.w || !bias) { err = 4; return l; }
        int row = 0;
        for (auto& m : mods) {
            const RawParam* w = v.ctx.find(m + ".weight");
            const int n = (int)w->shape[0];
            if (launch_pack_rows<T>(static_cast<const float*>(w->data->p), n, k, l.w, k, row, 0, 0, 0, 1.0f, st)) err = 5;
            if (const RawParam* b = v.ctx.find(m + ".bias"))
                if (hipMemcpyAsync(bias + row, b->data->p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) err = 5;
            row += n;
        }
        l.b = bias;
        return l;
    }
    VRes resnet(const std::string& name) {
        VRes r;
        r.n1 = norm(name + ".norm1");
        r.c1 = conv(name + ".conv1");
        r.n2 = norm(name + ".norm2");
        r.c2 = conv(name + ".conv2");
        r.cin = r.c1.cin; r.cout = r.c1.cout;
        if (v.ctx.find(name + ".conv_shortcut.weight")) { r.has_sc = true; r.sc = linear({name + ".conv_shortcut"}); }
        return r;
    }
    VMid mid(const std::string& name) {
        VMid m;
        m.r0 = resnet(name + ".resnets.0");
        const std::string a = name + ".attentions.0";
        m.at.gn = norm(a + ".group_norm");
        m.at.C = m.at.gn.c;
        m.at.qkv = linear({a + ".to_q", a + ".to_k", a + ".to_v"});
        m.at.out = linear({a + ".to_out.0"});
        m.r1 = resnet(name + ".resnets.1");
        return m;
    }
};

template <typename T>
int vae_finalize_t(mrisr_vae& v, hipStream_t st) {
    VPacker<T> pk{v, st};
    const mrisr_vae_cfg& c = v.cfg;
    v.ctx.packed.clear();
    v.enc.clear();
    v.dec.clear();
    const int L = c.num_levels;
    v.enc_in = pk.conv("encoder.conv_in");
    for (int i = 0; i < L; ++i) {
        VLevel lv;
        for (int j = 0; j < c.layers_per_block; ++j) lv.res.push_back(pk.resnet("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
        if (i < L - 1) { lv.has_resample = true; lv.resample = pk.conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv"); }
        v.enc.push_back(std::move(lv));
    }
    v.enc_mid = pk.mid("encoder.mid_block");
    v.enc_norm = pk.norm("encoder.conv_norm_out");
    v.enc_out = pk.conv("encoder.conv_out");
    v.quant = pk.conv("quant_conv");
    v.post_quant = pk.conv("post_quant_conv");
    v.dec_in = pk.conv("decoder.conv_in");
    v.dec_mid = pk.mid("decoder.mid_block");
    for (int i = 0; i < L; ++i) {
        VLevel lv;
        for (int j = 0; j < c.layers_per_block + 1; ++j) lv.res.push_back(pk.resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
        if (i < L - 1) { lv.has_resample = true; lv.resample = pk.conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv"); }
        v.dec.push_back(std::move(lv));
    }
    v.dec_norm = pk.norm("decoder.conv_norm_out");
    v.dec_out = pk.conv("decoder.conv_out", (c.out_channels + 3) & ~3);
    if (pk.err) return pk.err;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    v.finalized = true;
    v.ws_key.clear();
    return 0;
}

template <typename T>
struct VaeRunner : Runner<T> {
    typedef Runner<T> R;
    using R::m;
    using R::st;
    using R::dry;
    using R::new_act;
    using R::alloc;
    mrisr_vae& v;
    VaeRunner(mrisr_vae& vv, hipStream_t s, bool d) : R(vv.ctx, s, d), v(vv) {}
    float eps() const { return 1e-6f; }

    // stride-2 3x3 conv with diffusers' asymmetric (0,1,0,1) zero padding
    int conv_down(const Act& x, const ConvW& cw, Act* out) {
        MRISR_REQUIRE(cw.cin == x.C && cw.ks == 3 && x.H % 2 == 0 && x.W % 2 == 0, "VAE downsample conv geometry");
        *out = new_act(x.B, x.H / 2, x.W / 2, cw.cout);
        if (!out->p) return 7;
        GemmArgs g;
        g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
        g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H / 2; g.Wout = x.W / 2; g.stride = 2; g.pad = 0;
        g.w = cw.w; g.M = x.B * g.Hout * g.Wout; g.N = cw.cout; g.K = 9 * x.C; g.bias = cw.b;
        g.out = out->p; g.ldo = cw.cout;
        return R::run_gemm(g);
    }

    int resnet(const VRes& r, const Act& x, Act* out) {
        Act o = new_act(x.B, x.H, x.W, r.cout);
        if (!o.p) return 7;
        const size_t mk = m.arena.mark();
        Act xn, h, hn;
        TRY(R::gn(x, nullptr, r.n1, true, eps(), &xn));
        TRY(R::conv3(xn, nullptr, r.c1, 1, 0, nullptr, 0, 1, nullptr, ACT_NONE, &h));
        TRY(R::gn(h, nullptr, r.n2, true, eps(), &hn));
        Act res = x;
        if (r.has_sc) {
            GemmArgs g;
            g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
            g.w = r.sc.w; g.M = (int)x.rows(); g.N = r.cout; g.K = r.cin; g.bias = r.sc.b; g.out = o.p; g.ldo = r.cout;
            TRY(R::run_gemm(g));
            res = o;
        }
        GemmArgs g;
        g.a0 = hn.p; g.c0 = hn.C; g.lda0 = hn.C;
        g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W;
        g.w = r.c2.w; g.M = (int)x.rows(); g.N = r.cout; g.K = 9 * r.cout; g.bias = r.c2.b;
        g.resid = res.p; g.ldr = r.cout; g.out = o.p; g.ldo = r.cout;
        TRY(R::run_gemm(g));
        m.arena.release(mk);
        *out = o;
        return 0;
    }

    // single-head self-attention over the feature map (diffusers Attention, residual_connection=True)
    int attention(const VAttn& a, const Act& x, Act* out) {
        const int C = a.C, N = x.H * x.W, M = (int)x.rows();
        MRISR_REQUIRE(N % 64 == 0, "VAE attention: the latent must have a multiple of 64 positions");
        Act o = new_act(x.B, x.H, x.W, C);
        if (!o.p) return 7;
        const size_t mk = m.arena.mark();
        Act xn;
        TRY(R::gn(x, nullptr, a.gn, false, eps(), &xn));
        HeadBuf hb;
        hb.B = x.B; hb.H = 1; hb.N = N; hb.hd = C; hb.npad = N; hb.dpad = C;
        const size_t hsz = (size_t)M * C * sizeof(T);
        hb.q = alloc(hsz); hb.k = alloc(hsz); hb.vt = alloc(hsz);
        T* ao = static_cast<T*>(alloc(hsz));
        if (!hb.q || !hb.k || !hb.vt || !ao) return 7;
        {
            GemmArgs g;
            R::heads_args(&g, hb, hb.q, 0, hb.k, 0, hb.vt, 1, N, N);
            TRY(R::linear(xn.p, M, C, a.qkv, ACT_NONE, nullptr, 0, &g, nullptr, 0));
        }
        TRY(R::attention(hb, hb.k, hb.vt, N, N, ao));
        TRY(R::linear(ao, M, C, a.out, ACT_NONE, x.p, C, nullptr, o.p, C));
        m.arena.release(mk);
        *out = o;
        return 0;
    }
    int mid(const VMid& md, Act x, Act* out) {
        Act y;
        TRY(resnet(md.r0, x, &y)); x = y;
        TRY(attention(md.at, x, &y)); x = y;
        TRY(resnet(md.r1, x, &y));
        *out = y;
        return 0;
    }

    int encode(const mrisr_tensor& img, const mrisr_tensor& moments) {
        m.arena.reset();
        Act s, x, y;
        TRY(R::import_act(img, &s, false));
        TRY(R::direct(s, v.enc_in, 1, ACT_NONE, nullptr, &x));
        for (auto& lv : v.enc) {
            for (auto& r : lv.res) { TRY(resnet(r, x, &y)); x = y; }
            if (lv.has_resample) { TRY(conv_down(x, lv.resample, &y)); x = y; }
        }
        TRY(mid(v.enc_mid, x, &x));
        Act xn;
        TRY(R::gn(x, nullptr, v.enc_norm, true, eps(), &xn));
        TRY(R::conv3(xn, nullptr, v.enc_out, 1, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
        Act q;
        TRY(R::direct(y, v.quant, 1, ACT_NONE, nullptr, &q));
        return R::export_act(q, moments, 1.0f);
    }

    int decode(const mrisr_tensor& z, const mrisr_tensor& out) {
        m.arena.reset();
        Act s, x, y;
        TRY(R::import_act(z, &s, false));
        TRY(R::direct(s, v.post_quant, 1, ACT_NONE, nullptr, &x));
        TRY(R::direct(x, v.dec_in, 1, ACT_NONE, nullptr, &y));
        x = y;
        TRY(mid(v.dec_mid, x, &x));
        for (auto& lv : v.dec) {
            for (auto& r : lv.res) { TRY(resnet(r, x, &y)); x = y; }
            if (lv.has_resample) { TRY(R::conv3(x, nullptr, lv.resample, 1, 1, nullptr, 0, 1, nullptr, ACT_NONE, &y)); x = y; }
        }
        Act xn;
        TRY(R::gn(x, nullptr, v.dec_norm, true, eps(), &xn));
        TRY(R::conv3(xn, nullptr, v.dec_out, 1, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));  // Cout padded to 4
        // [B][H][W][4] -> NCHW planes, then the first out_channels planes of every sample to the caller
        const int Co = v.cfg.out_channels;
        MRISR_REQUIRE(out.ndim == 4 && out.layout == MRISR_NCHW && out.shape[0] == y.B && out.shape[1] == Co && out.shape[2] == y.H &&
                          out.shape[3] == y.W,
                      "decode output: NCHW [B, out_channels, 8h, 8w]");
        const size_t es = dtype_size(out.dtype), plane = (size_t)y.H * y.W * es;
        void* tmp = alloc((size_t)y.B * y.C * plane);
        if (!tmp) return 7;
        if (dry) return 0;
        TRY(launch_nhwc_to_nchw<T>(y.p, tmp, out.dtype, y.B, y.C, y.H, y.W, 1.0f, st));
        MRISR_CHECK_HIP(hipMemcpy2DAsync(out.data, Co * plane, tmp, y.C * plane, Co * plane, y.B, hipMemcpyDeviceToDevice, st));
        return 0;
    }
};

template <typename T>
int vae_run_t(mrisr_vae& v, bool enc, const mrisr_tensor& in, const mrisr_tensor& out, hipStream_t st) {
    char key[96];
    snprintf(key, sizeof(key), "%c,%d,%d,%d,%d", enc ? 'e' : 'd', (int)in.shape[0], (int)in.shape[2], (int)in.shape[3], (int)out.dtype);
    if (v.ws_key != key) {
        v.ctx.arena.dry = true;
        v.ctx.arena.reset();
        v.ctx.arena.peak = 0;
        int rc;
        {
            VaeRunner<T> r(v, st, true);
            rc = enc ? r.encode(in, out) : r.decode(in, out);
        }
        v.ctx.arena.dry = false;
        if (rc) return rc;
        MRISR_CHECK_HIP(hipStreamSynchronize(st));
        TRY(v.ctx.arena.buf.reserve(v.ctx.arena.peak + 4096, false));
        v.ctx.arena.reset();
        v.ws_key = key;
    }
    VaeRunner<T> r(v, st, false);
    return enc ? r.encode(in, out) : r.decode(in, out);
}

}  // namespace

extern "C" {

int mrisr_vae_create(const mrisr_vae_cfg* cfg, mrisr_vae** out) {
    API_BEGIN
    MRISR_REQUIRE(cfg && out, "null argument");
    MRISR_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 4 && cfg->layers_per_block >= 1, "VAE levels");
    MRISR_REQUIRE(cfg->compute_dtype == MRISR_F32 || cfg->compute_dtype == MRISR_BF16, "compute dtype f32 or bf16");
    const int bk = cfg->compute_dtype == MRISR_F32 ? 32 : 64;
    for (int i = 0; i < cfg->num_levels; ++i)
        MRISR_REQUIRE(cfg->block_out_channels[i] % bk == 0 && cfg->block_out_channels[i] % cfg->norm_num_groups == 0,
                      "block_out_channels must be multiples of the 128-byte K tile and of the norm groups");
    MRISR_REQUIRE(cfg->latent_channels % 4 == 0, "latent_channels must be a multiple of 4");
    int dev_count = 0;
    MRISR_CHECK_HIP(hipGetDeviceCount(&dev_count));
    MRISR_REQUIRE(dev_count > 0, "no HIP device: libmrisr has no CPU fallback");
    auto* v = new mrisr_vae();
    v->cfg = *cfg;
    v->ctx.cfg.norm_num_groups = cfg->norm_num_groups;
    v->ctx.cfg.compute_dtype = cfg->compute_dtype;
    v->ctx.cfg.flash_attention = 0;
    *out = v;
    return 0;
    API_END
}
void mrisr_vae_destroy(mrisr_vae* v) { delete v; }
int mrisr_vae_set_param(mrisr_vae* v, const char* key, const float* data, const int64_t* shape, int ndim, int is_device) {
    API_BEGIN
    MRISR_REQUIRE(v, "null handle");
    v->finalized = false;
    return v->ctx.set_param(key, data, shape, ndim, is_device);
    API_END
}
int64_t mrisr_vae_num_params(const mrisr_vae* v) { return v ? v->ctx.num_params() : 0; }
int mrisr_vae_finalize(mrisr_vae* v, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(v, "null handle");
    TRY(gemm_prepare());
    return v->cfg.compute_dtype == MRISR_F32 ? vae_finalize_t<float>(*v, (hipStream_t)stream) : vae_finalize_t<bf16>(*v, (hipStream_t)stream);
    API_END
}
int mrisr_vae_encode(mrisr_vae* v, const mrisr_tensor* image, mrisr_tensor* moments, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(v && v->finalized, "call mrisr_vae_finalize first");
    const int f = 1 << (v->cfg.num_levels - 1);
    MRISR_REQUIRE(image && image->ndim == 4 && image->shape[1] == v->cfg.in_channels && image->shape[2] % f == 0 && image->shape[3] % f == 0,
                  "image must be [B, in_channels, H, W] with H, W divisible by 2^(levels-1)");
    MRISR_REQUIRE(moments && moments->ndim == 4 && moments->shape[0] == image->shape[0] && moments->shape[1] == 2 * v->cfg.latent_channels &&
                      moments->shape[2] == image->shape[2] / f && moments->shape[3] == image->shape[3] / f,
                  "moments must be [B, 2*latent_channels, H/f, W/f]");
    return v->cfg.compute_dtype == MRISR_F32 ? vae_run_t<float>(*v, true, *image, *moments, (hipStream_t)stream)
                                             : vae_run_t<bf16>(*v, true, *image, *moments, (hipStream_t)stream);
    API_END
}
int mrisr_vae_decode(mrisr_vae* v, const mrisr_tensor* latents, mrisr_tensor* image, void* stream) {
    API_BEGIN
    MRISR_REQUIRE(v && v->finalized, "call mrisr_vae_finalize first");
    const int f = 1 << (v->cfg.num_levels - 1);
    MRISR_REQUIRE(latents && latents->ndim == 4 && latents->shape[1] == v->cfg.latent_channels, "latents must be [B, latent_channels, h, w]");
    MRISR_REQUIRE(image && image->ndim == 4 && image->shape[0] == latents->shape[0] && image->shape[2] == latents->shape[2] * f &&
                      image->shape[3] == latents->shape[3] * f,
                  "image must be [B, out_channels, f*h, f*w]");
    return v->cfg.compute_dtype == MRISR_F32 ? vae_run_t<float>(*v, false, *latents, *image, (hipStream_t)stream)
                                             : vae_run_t<bf16>(*v, false, *latents, *image, (hipStream_t)stream);
    API_END
}

}  // extern "C"
