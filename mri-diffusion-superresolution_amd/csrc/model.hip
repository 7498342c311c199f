// Model runtime: weights keyed by diffusers state-dict names, load-time re-layout for the kernels, and the
// UNet2DConditionModel / ControlNetModel forward as a fixed sequence of kernel launches over a bump arena
// (deterministic addresses -> the whole step is hipGraph-capturable).  Spec: SURVEY.md App. A.1-A.6;
// reference call sites src/adapters/res_srdiff.py:65-78.
#include "model.h"

#include <climits>
#include <cmath>
#include <cstring>

namespace mrisr {

thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error_cstr() { return g_err.c_str(); }

#define TRY(expr)              \
    do {                       \
        int _rc = (expr);      \
        if (_rc) return _rc;   \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ================================================================================================
// parameters
// ================================================================================================
int Model::set_param(const char* key, const float* data, const int64_t* shape, int ndim, int is_device) {
    MRISR_REQUIRE(key && data && ndim >= 1 && ndim <= 4, "bad parameter");
    std::string k(key);
    // peft wraps a LoRA-targeted Linear: <mod>.base_layer.weight
    const std::string bl = ".base_layer.";
    size_t pos = k.find(bl);
    if (pos != std::string::npos) k = k.substr(0, pos) + "." + k.substr(pos + bl.size());
    RawParam rp;
    rp.shape.assign(shape, shape + ndim);
    const size_t bytes = (size_t)rp.numel() * sizeof(float);
    rp.data = std::make_shared<DevBuf>();
    TRY(rp.data->reserve(bytes, false));
    MRISR_CHECK_HIP(hipMemcpy(rp.data->p, data, bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    raw[k] = rp;
    finalized = false;
    return 0;
}

int64_t Model::num_params() const {
    int64_t n = 0;
    for (auto& kv : raw) n += kv.second.numel();
    return n;
}

const RawParam* Model::find(const std::string& k) const {
    auto it = raw.find(k);
    return it == raw.end() ? nullptr : &it->second;
}

void* Model::new_packed(size_t bytes, bool zero) {
    packed.emplace_back(new DevBuf());
    if (packed.back()->reserve(bytes ? bytes : 16, zero)) return nullptr;
    return packed.back()->p;
}

template <typename T>
struct Packer {
    Model& m;
    hipStream_t st;
    int err = 0;
    explicit Packer(Model& mm, hipStream_t s) : m(mm), st(s) {}

    const RawParam* need(const std::string& k) {
        const RawParam* r = m.find(k);
        if (!r && !err) {  // keep the FIRST missing key as the message
            set_error("missing parameter: " + k);
            err = 3;
        }
        return r;
    }
    const float* f32(const std::string& k) {  // bias / gamma / beta stay f32: use the raw copy directly
        const RawParam* r = need(k);
        return r ? static_cast<const float*>(r->data->p) : nullptr;
    }
    NormW norm(const std::string& name) {
        NormW n;
        n.g = f32(name + ".weight");
        n.b = f32(name + ".bias");
        if (n.g) n.c = (int)m.find(name + ".weight")->shape[0];
        return n;
    }
    ConvW conv(const std::string& name) {
        ConvW c;
        const RawParam* w = need(name + ".weight");
        if (!w) return c;
        c.cout = (int)w->shape[0];
        c.cin = (int)w->shape[1];
        c.ks = (int)w->shape[2];
        c.w = m.new_packed((size_t)w->numel() * sizeof(T), false);
        if (!c.w) { err = 4; return c; }
        if (launch_pack_conv3x3<T>(static_cast<const float*>(w->data->p), c.w, c.cout, c.cin, c.ks, st)) err = 5;
        c.b = m.find(name + ".bias") ? f32(name + ".bias") : nullptr;
        return c;
    }
    // Linear / 1x1 conv, optionally several modules fused along N (QKV, KV), optional LoRA, optional GEGLU.
    LinW linear(const std::vector<std::string>& mods, bool geglu = false) {
        LinW l;
        const bool fused_lora = m.cfg.lora_rank > 0 && m.cfg.lora_fused;
        int ntot = 0, k = 0;
        bool any_lora = false;
        int rsum = 0, rmod = 0, nmod0 = 0;
        for (auto& mod : mods) {
            const RawParam* w = need(mod + ".weight");
            if (!w) return l;
            ntot += (int)w->shape[0];
            k = (int)w->shape[1];
            const RawParam* la = m.find(mod + ".lora_A.default.weight");
            if (la) { any_lora = true; rsum += (int)la->shape[0]; rmod = (int)la->shape[0]; }
            if (!nmod0) nmod0 = (int)w->shape[0];
        }
        l.n = ntot;
        l.k = k;
        const int ktot = k;
        float* lbuf = nullptr;
        if (any_lora && fused_lora) {
            // one rank-r adapter slot per fused module (modules without an adapter keep zero A rows / B rows)
            l.r = rmod;
            l.R = (int)mods.size() * rmod;
            l.secN = nmod0;
            l.loraA = m.new_packed((size_t)l.R * k * sizeof(T), true);
            lbuf = static_cast<float*>(m.new_packed((size_t)ntot * rmod * sizeof(float), true));
            if (!l.loraA || !lbuf) { err = 4; return l; }
            l.loraB = lbuf;
        }
        l.w = m.new_packed((size_t)ntot * ktot * sizeof(T), false);
        if (!l.w) { err = 4; return l; }
        bool has_bias = false;
        for (auto& mod : mods) has_bias |= m.find(mod + ".bias") != nullptr;
        float* bias = nullptr;
        if (has_bias) {
            bias = static_cast<float*>(m.new_packed((size_t)ntot * sizeof(float), true));
            if (!bias) { err = 4; return l; }
        }
        int row = 0, rcol = 0;
        for (auto& mod : mods) {
            const RawParam* w = m.find(mod + ".weight");
            const int n = (int)w->shape[0];
            const float* wsrc = static_cast<const float*>(w->data->p);
            const RawParam* la = m.find(mod + ".lora_A.default.weight");
            const RawParam* lb = m.find(mod + ".lora_B.default.weight");
            std::unique_ptr<DevBuf> merged;
            if (la && lb && !fused_lora) {
                // merged mode: W' = W + s * B A, computed in f32 on the device by the f32 GEMM itself
                merged.reset(new DevBuf());
                if (merged->reserve((size_t)n * k * sizeof(float), false)) { err = 4; return l; }
                if (merge_lora(wsrc, static_cast<const float*>(la->data->p), static_cast<const float*>(lb->data->p),
                               static_cast<float*>(merged->p), n, k, (int)la->shape[0]))
                    err = 6;
                wsrc = static_cast<const float*>(merged->p);
            }
            if (launch_pack_rows<T>(wsrc, n, k, l.w, ktot, row, 0, geglu ? 1 : 0, n / 2, 1.0f, st)) err = 5;
            if (la && lb && fused_lora) {
                const int r = (int)la->shape[0];
                if (r != l.r || n != l.secN) { set_error("fused LoRA needs the same rank / width for every fused module: " + mod); err = 6; }
                // A rows -> loraA[rcol .. rcol+r);  (alpha/r) * B -> f32 [n][r] rows of this module
                if (launch_pack_rows<T>(static_cast<const float*>(la->data->p), r, k, l.loraA, k, rcol, 0, 0, 0, 1.0f, st)) err = 5;
                if (launch_pack_rows<float>(static_cast<const float*>(lb->data->p), n, r, lbuf, r, row, 0, 0, 0, m.lora_scale, st)) err = 5;
            }
            if (fused_lora && any_lora) rcol += l.r;
            if (const RawParam* b = m.find(mod + ".bias")) {
                const float* bsrc = static_cast<const float*>(b->data->p);
                if (geglu) {
                    if (launch_pack_bias_geglu(bsrc, bias + row, n / 2, st)) err = 5;
                } else if (hipMemcpyAsync(bias + row, bsrc, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) err = 5;
            }
            if (merged) (void)hipStreamSynchronize(st);  // merged buffer dies at scope end
            row += n;
        }
        l.b = bias;
        return l;
    }
    // host-side helper for merged mode (load time only): W' = W + s*B*A via a tiny kernel
    int merge_lora(const float* w, const float* A, const float* B, float* out, int n, int k, int r);

    ResW resnet(const std::string& name) {
        ResW r;
        r.n1 = norm(name + ".norm1");
        r.c1 = conv(name + ".conv1");
        r.n2 = norm(name + ".norm2");
        r.c2 = conv(name + ".conv2");
        r.cin = r.c1.cin;
        r.cout = r.c1.cout;
        if (m.find(name + ".conv_shortcut.weight")) {
            r.has_sc = true;
            r.sc = linear({name + ".conv_shortcut"});
        }
        m.temb_mods.push_back(name + ".time_emb_proj");
        r.temb_off = m.tproj_total;
        m.tproj_total += r.cout;
        return r;
    }
    XfW transformer(const std::string& name) {
        XfW x;
        x.norm = norm(name + ".norm");
        x.C = x.norm.c;
        x.proj_in = linear({name + ".proj_in"});
        x.proj_out = linear({name + ".proj_out"});
        const std::string b = name + ".transformer_blocks.0";
        x.ln1 = norm(b + ".norm1");
        x.ln2 = norm(b + ".norm2");
        x.ln3 = norm(b + ".norm3");
        x.qkv = linear({b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"});
        x.out1 = linear({b + ".attn1.to_out.0"});
        x.q2 = linear({b + ".attn2.to_q"});
        x.kv2 = linear({b + ".attn2.to_k", b + ".attn2.to_v"});
        x.out2 = linear({b + ".attn2.to_out.0"});
        x.ff1 = linear({b + ".ff.net.0.proj"}, true);
        x.ff2 = linear({b + ".ff.net.2"});
        return x;
    }
};

__global__ void merge_lora_kernel(const float* w, const float* A, const float* B, float* out, int n, int k, int r,
                                  float s) {
    const long long total = (long long)n * k;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int row = (int)(i / k), col = (int)(i - (long long)row * k);
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += B[(size_t)row * r + j] * A[(size_t)j * k + col];
        out[i] = w[i] + s * acc;
    }
}
template <typename T>
int Packer<T>::merge_lora(const float* w, const float* A, const float* B, float* out, int n, int k, int r) {
    long long blocks = ((long long)n * k + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(merge_lora_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, A, B, out, n, k, r, m.lora_scale);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
static int finalize_t(Model& m, hipStream_t st) {
    Packer<T> pk(m, st);
    const mrisr_unet_cfg& c = m.cfg;
    m.packed.clear();
    m.temb_mods.clear();
    m.tproj_total = 0;
    m.down.clear();
    m.up.clear();
    m.conv_in = pk.conv("conv_in");
    m.te1 = pk.linear({"time_embedding.linear_1"});
    m.te2 = pk.linear({"time_embedding.linear_2"});
    const int L = c.num_levels;
    for (int i = 0; i < L; ++i) {
        Level lv;
        for (int j = 0; j < c.layers_per_block; ++j) lv.res.push_back(pk.resnet("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
        if (c.attn_levels[i])
            for (int j = 0; j < c.layers_per_block; ++j) lv.xf.push_back(pk.transformer("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j)));
        if (i < L - 1) {
            lv.has_down = true;
            lv.down = pk.conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv");
        }
        m.down.push_back(std::move(lv));
    }
    m.mid_r0 = pk.resnet("mid_block.resnets.0");
    m.mid_xf = pk.transformer("mid_block.attentions.0");
    m.mid_r1 = pk.resnet("mid_block.resnets.1");
    if (!m.is_controlnet) {
        for (int i = 0; i < L; ++i) {
            const int lvl = L - 1 - i;
            Level lv;
            for (int j = 0; j < c.layers_per_block + 1; ++j) lv.res.push_back(pk.resnet("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
            if (c.attn_levels[lvl])
                for (int j = 0; j < c.layers_per_block + 1; ++j) lv.xf.push_back(pk.transformer("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j)));
            if (i < L - 1) {
                lv.has_up = true;
                lv.up = pk.conv("up_blocks." + std::to_string(i) + ".upsamplers.0.conv");
            }
            m.up.push_back(std::move(lv));
        }
        m.norm_out = pk.norm("conv_norm_out");
        m.conv_out = pk.conv("conv_out");
    } else {
        m.ce.clear();
        m.ce_stride.clear();
        m.ce.push_back(pk.conv("controlnet_cond_embedding.conv_in"));
        m.ce_stride.push_back(1);
        int nb = 0;
        while (m.find("controlnet_cond_embedding.blocks." + std::to_string(nb) + ".weight")) ++nb;
        for (int k = 0; k < nb; ++k) {
            m.ce.push_back(pk.conv("controlnet_cond_embedding.blocks." + std::to_string(k)));
            m.ce_stride.push_back(k % 2 ? 2 : 1);
        }
        m.ce.push_back(pk.conv("controlnet_cond_embedding.conv_out"));
        m.ce_stride.push_back(1);
        m.cn_down.clear();
        for (int k = 0; k < m.num_skips(); ++k) m.cn_down.push_back(pk.linear({"controlnet_down_blocks." + std::to_string(k)}));
        m.cn_mid = pk.linear({"controlnet_mid_block"});
    }
    if (pk.err) return pk.err;  // e.g. "missing parameter: <key>" (message already set)
    // all time_emb_proj linears as ONE [sum C][temb] matrix: a single weight-streaming launch per step
    {
        const int temb = 4 * c.block_out_channels[0];
        LinW tp;
        tp.n = m.tproj_total;
        tp.k = temb;
        tp.w = m.new_packed((size_t)tp.n * temb * sizeof(T), false);
        float* bias = static_cast<float*>(m.new_packed((size_t)tp.n * sizeof(float), true));
        if (!tp.w || !bias) return 4;
        int row = 0;
        for (auto& mod : m.temb_mods) {
            const RawParam* w = pk.need(mod + ".weight");
            const RawParam* b = pk.need(mod + ".bias");
            if (!w || !b) break;
            const int n = (int)w->shape[0];
            if (launch_pack_rows<T>(static_cast<const float*>(w->data->p), n, temb, tp.w, temb, row, 0, 0, 0, 1.0f, st)) pk.err = 5;
            MRISR_CHECK_HIP(hipMemcpyAsync(bias + row, b->data->p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
            row += n;
        }
        tp.b = bias;
        m.tproj = tp;
    }
    if (pk.err) return pk.err;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    m.finalized = true;
    m.ws_key = "";  // force workspace re-plan (cross-attention caches depend on weights)
    m.ctx_valid = false;
    m.cond_valid = false;
    return 0;
}

int Model::finalize(hipStream_t st) {
    TRY(init_zero_page());
    if (cfg.compute_dtype == MRISR_F32) return finalize_t<float>(*this, st);
    return finalize_t<bf16>(*this, st);
}

int Model::num_skips() const {
    int n = 1;
    for (int i = 0; i < cfg.num_levels; ++i) n += cfg.layers_per_block + (i < cfg.num_levels - 1 ? 1 : 0);
    return n;
}
int Model::skip_shape(int k, int B, int h, int w, int64_t shape[4]) const {
    std::vector<std::pair<int, int>> s;  // (channels, level)
    s.push_back({cfg.block_out_channels[0], 0});
    for (int i = 0; i < cfg.num_levels; ++i) {
        for (int j = 0; j < cfg.layers_per_block; ++j) s.push_back({cfg.block_out_channels[i], i});
        if (i < cfg.num_levels - 1) s.push_back({cfg.block_out_channels[i], i + 1});
    }
    s.push_back({cfg.block_out_channels[cfg.num_levels - 1], cfg.num_levels - 1});  // mid
    MRISR_REQUIRE(k >= 0 && k < (int)s.size(), "skip index");
    shape[0] = B;
    shape[1] = s[k].first;
    shape[2] = h >> s[k].second;
    shape[3] = w >> s[k].second;
    return 0;
}

// ================================================================================================
// forward
// ================================================================================================
template <typename T>
struct Runner {
    Model& m;
    hipStream_t st;
    bool dry;
    Runner(Model& mm, hipStream_t s, bool d) : m(mm), st(s), dry(d) {}

    void* alloc(size_t bytes) {
        void* p = m.arena.alloc(bytes);
        if (!p) set_error("workspace arena exhausted");
        return p;
    }
    Act new_act(int B, int H, int W, int C) {
        Act a;
        a.B = B; a.H = H; a.W = W; a.C = C;
        a.p = alloc(a.numel() * sizeof(T));
        return a;
    }

    int gn(const Act& x0, const Act* x1, const NormW& nw, bool silu, float eps, Act* out) {
        const int C = x0.C + (x1 ? x1->C : 0);
        MRISR_REQUIRE(nw.c == C, "GroupNorm channel mismatch");
        *out = new_act(x0.B, x0.H, x0.W, C);
        GroupNormArgs a;
        a.x0 = x0.p; a.c0 = x0.C;
        a.x1 = x1 ? x1->p : nullptr; a.c1 = x1 ? x1->C : 0;
        a.B = x0.B; a.HW = x0.H * x0.W; a.groups = m.cfg.norm_num_groups; a.eps = eps;
        a.gamma = nw.g; a.beta = nw.b; a.silu = silu ? 1 : 0; a.y = out->p;
        a.nsplit = groupnorm_nsplit(a.B, a.HW);
        a.partial = static_cast<float*>(alloc((size_t)a.B * a.nsplit * a.groups * 2 * sizeof(float)));
        if (!out->p || !a.partial) return 7;
        if (dry) return 0;
        return launch_groupnorm<T>(a, st);
    }

    int run_gemm(GemmArgs& g) {
        TRY(gemm_choose(g, sizeof(T) == 2));
        if (g.splitk > 1) {
            g.partial = static_cast<float*>(alloc((size_t)g.splitk * g.batch * g.M * g.N * sizeof(float)));
            if (!g.partial) return 7;
        }
        if (dry) return 0;
        return launch_gemm<T>(g, st);
    }

    // 3x3 conv (pad 1) over NHWC, optional second concat source, stride, nearest-x2 upsample of the input
    int conv3(const Act& x, const Act* x1, const ConvW& cw, int stride, int ups, const float* rowvec, int rowvec_ld,
              int rowvec_div, const Act* resid, int act, Act* out) {
        const int Cin = x.C + (x1 ? x1->C : 0);
        MRISR_REQUIRE(cw.cin == Cin && cw.ks == 3, "conv3x3 weight mismatch");
        const int Hc = x.H << ups, Wc = x.W << ups;
        const int Ho = (Hc - 1) / stride + 1, Wo = (Wc - 1) / stride + 1;
        *out = new_act(x.B, Ho, Wo, cw.cout);
        if (!out->p) return 7;
        GemmArgs g;
        g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
        if (x1) { g.a1 = x1->p; g.c1 = x1->C; g.lda1 = x1->C; }
        g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = Ho; g.Wout = Wo; g.stride = stride; g.ups = ups;
        g.w = cw.w; g.M = x.B * Ho * Wo; g.N = cw.cout; g.K = 9 * Cin;
        g.bias = cw.b; g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rowvec_div = rowvec_div; g.act = act;
        if (resid) { g.resid = resid->p; g.ldr = resid->C; }
        g.out = out->p; g.ldo = cw.cout;
        return run_gemm(g);
    }

    // y[M][n] = x[M][k] W^T (+LoRA tail) + bias ...; x given as raw rows
    int linear(const void* x, int M, int lda, const LinW& lw, int act, const void* resid, int ldr, GemmArgs* custom,
               void* out, int ldo) {
        GemmArgs g = custom ? *custom : GemmArgs();
        const size_t mk = m.arena.mark();
        g.a0 = x; g.c0 = lw.k; g.lda0 = lda;
        if (lw.R) {
            // LoRA: z = x A^T (f32 [M][R], one bandwidth-bound pass over x); the rank-r up-projection (alpha/r) B z is
            // accumulated in the projection GEMM's epilogue
            float* z = static_cast<float*>(alloc((size_t)M * lw.R * sizeof(float)));
            if (!z) return 7;
            if (!dry) TRY(launch_lora_down<T>(x, lda, lw.loraA, z, M, lw.k, lw.R, st));
            g.lora_z = z; g.lora_zld = lw.R; g.lora_b = lw.loraB; g.lora_r = lw.r; g.lora_secN = lw.secN;
        }
        g.w = lw.w; g.M = M; g.N = lw.n; g.K = lw.k;
        g.alg_flops = 2.0 * M * (double)lw.n * (lw.k + lw.r);
        g.bias = lw.b; g.act = act; g.resid = resid; g.ldr = ldr;
        if (g.out_mode != OUT_HEADS) { g.out = out; g.ldo = ldo; }
        TRY(run_gemm(g));
        m.arena.release(mk);
        return 0;
    }

    int layernorm(const void* x, const NormW& nw, int M, int C, void* y) {
        if (dry) return 0;
        return launch_layernorm<T>(x, y, nw.g, nw.b, M, C, 1e-5f, st);
    }

    // ---- ResnetBlock2D (App. A.3) ----
    int resnet(const ResW& r, const Act& x, const Act* x1, Act* out) {
        Act o = new_act(x.B, x.H, x.W, r.cout);  // allocated first: survives the temporaries below
        if (!o.p) return 7;
        const size_t mk = m.arena.mark();
        Act xn, h, hn;
        TRY(gn(x, x1, r.n1, true, m.cfg.norm_eps, &xn));
        const int div = m.t_scalar ? INT_MAX : x.H * x.W;
        TRY(conv3(xn, nullptr, r.c1, 1, 0, m.tproj_out + r.temb_off, m.tproj_total, div, nullptr, ACT_NONE, &h));
        TRY(gn(h, nullptr, r.n2, true, m.cfg.norm_eps, &hn));
        Act res = x;
        if (r.has_sc) {
            // 1x1 shortcut on the (concatenated) raw input, written straight into the output buffer
            GemmArgs g;
            if (x1) { g.a1 = x1->p; g.c1 = x1->C; g.lda1 = x1->C; }
            g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
            g.w = r.sc.w; g.M = (int)x.rows(); g.N = r.cout; g.K = r.cin; g.bias = r.sc.b; g.out = o.p; g.ldo = r.cout;
            MRISR_REQUIRE(r.sc.R == 0, "shortcut has no LoRA");
            TRY(run_gemm(g));
            res = o;
        } else {
            MRISR_REQUIRE(!x1, "concat input requires a shortcut conv");
        }
        // conv2 + bias + residual -> o (in place when res == o: each element is read then written by one lane)
        {
            GemmArgs g;
            g.a0 = hn.p; g.c0 = hn.C; g.lda0 = hn.C;
            g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W;
            g.w = r.c2.w; g.M = (int)x.rows(); g.N = r.cout; g.K = 9 * r.cout; g.bias = r.c2.b;
            g.resid = res.p; g.ldr = r.cout; g.out = o.p; g.ldo = r.cout;
            TRY(run_gemm(g));
        }
        m.arena.release(mk);
        *out = o;
        return 0;
    }

    // ---- attention core on head-major operands ----
    int attention(const HeadBuf& hb, const void* k, const void* vt, int nk, int nkpad, void* out_rows) {
        const int BH = hb.B * hb.H;
        const float scale = 1.0f / sqrtf((float)hb.hd);
        if (m.cfg.flash_attention && sizeof(T) == 2) {
            AttnArgs a;
            a.q = hb.q; a.k = k; a.vt = vt; a.out = out_rows;
            a.B = hb.B; a.H = hb.H; a.nq = hb.N; a.nk = nk; a.nkpad = nkpad; a.hd = hb.hd; a.dpad = hb.dpad;
            a.scale = scale;
            if (dry) return 0;
            return launch_attention_bf16(a, st);
        }
        const size_t mk = m.arena.mark();
        float* S = static_cast<float*>(alloc((size_t)BH * hb.N * nkpad * sizeof(float)));
        if (!S) return 7;
        void* P = S;
        if (sizeof(T) == 2) {
            P = alloc((size_t)BH * hb.N * nkpad * sizeof(T));
            if (!P) return 7;
        }
        GemmArgs g;
        g.a0 = hb.q; g.c0 = hb.dpad; g.lda0 = hb.dpad; g.a_bs = (long long)hb.npad * hb.dpad;
        g.w = k; g.w_bs = (long long)nkpad * hb.dpad;
        g.M = hb.N; g.N = nkpad; g.K = hb.dpad; g.batch = BH; g.alpha = scale;
        g.out_mode = OUT_F32; g.out = S; g.ldo = nkpad; g.o_bs = (long long)hb.N * nkpad;
        TRY(run_gemm(g));
        if (!dry) TRY(launch_softmax_rows<T>(S, nkpad, P, nkpad, (long long)BH * hb.N, nk, st));
        GemmArgs o;
        o.a0 = P; o.c0 = nkpad; o.lda0 = nkpad; o.a_bs = (long long)hb.N * nkpad;
        o.w = vt; o.w_bs = (long long)hb.dpad * nkpad;
        o.M = hb.N; o.N = hb.hd; o.K = nkpad; o.batch = BH;
        o.heads = hb.H; o.o_bs = (long long)hb.N * hb.H * hb.hd; o.o_hs = hb.hd;
        o.out = out_rows; o.ldo = hb.H * hb.hd;
        TRY(run_gemm(o));
        m.arena.release(mk);
        return 0;
    }

    void heads_args(GemmArgs* g, const HeadBuf& hb, void* s0, int tr0, void* s1, int tr1, void* s2, int tr2, int ntok,
                    int npad) {
        g->out_mode = OUT_HEADS;
        g->sec_ptr[0] = s0; g->sec_ptr[1] = s1; g->sec_ptr[2] = s2;
        g->sec_tr[0] = tr0; g->sec_tr[1] = tr1; g->sec_tr[2] = tr2;
        g->secC = hb.H * hb.hd; g->hd = hb.hd; g->dpad = hb.dpad; g->ntok = ntok; g->npad = npad; g->nheads = hb.H;
    }

    // cross-attention K / V^T of the prompt embedding: timestep-invariant, cached per transformer block
    int project_context(XfW& x, const HeadBuf& hb) {
        GemmArgs g;
        heads_args(&g, hb, x.kc, 0, x.vtc, 1, nullptr, 0, m.ctx_len, m.ctx_pad);
        return linear(m.ctx_rows, m.ws_B * m.ctx_len, m.cfg.cross_attention_dim, x.kv2, ACT_NONE, nullptr, 0, &g,
                      nullptr, 0);
    }

    // ---- Transformer2DModel with one BasicTransformerBlock (App. A.4) ----
    int transformer(XfW& xw, const Act& x, Act* out) {
        const int C = xw.C, M = (int)x.rows();
        const HeadBuf& hb = m.head_buf(x.H * x.W, C);
        Act o = new_act(x.B, x.H, x.W, C);
        if (!o.p) return 7;
        const size_t mk = m.arena.mark();
        Act xn;
        TRY(gn(x, nullptr, xw.norm, false, 1e-6f, &xn));
        T* t = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        T* nrm = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        T* ao = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        if (!t || !nrm || !ao) return 7;
        TRY(linear(xn.p, M, C, xw.proj_in, ACT_NONE, nullptr, 0, nullptr, t, C));
        // self-attention
        TRY(layernorm(t, xw.ln1, M, C, nrm));
        {
            GemmArgs g;
            heads_args(&g, hb, hb.q, 0, hb.k, 0, hb.vt, 1, hb.N, hb.npad);
            TRY(linear(nrm, M, C, xw.qkv, ACT_NONE, nullptr, 0, &g, nullptr, 0));
        }
        TRY(attention(hb, hb.k, hb.vt, hb.N, hb.npad, ao));
        TRY(linear(ao, M, C, xw.out1, ACT_NONE, t, C, nullptr, t, C));
        // cross-attention (K/V cached by set_context)
        TRY(layernorm(t, xw.ln2, M, C, nrm));
        {
            GemmArgs g;
            heads_args(&g, hb, hb.q, 0, nullptr, 0, nullptr, 0, hb.N, hb.npad);
            TRY(linear(nrm, M, C, xw.q2, ACT_NONE, nullptr, 0, &g, nullptr, 0));
        }
        TRY(attention(hb, xw.kc, xw.vtc, m.ctx_len, m.ctx_pad, ao));
        TRY(linear(ao, M, C, xw.out2, ACT_NONE, t, C, nullptr, t, C));
        // GEGLU feed-forward
        TRY(layernorm(t, xw.ln3, M, C, nrm));
        {
            T* ff = static_cast<T*>(alloc((size_t)M * 4 * C * sizeof(T)));
            if (!ff) return 7;
            TRY(linear(nrm, M, C, xw.ff1, ACT_GEGLU, nullptr, 0, nullptr, ff, 4 * C));
            TRY(linear(ff, M, 4 * C, xw.ff2, ACT_NONE, t, C, nullptr, t, C));
        }
        TRY(linear(t, M, C, xw.proj_out, ACT_NONE, x.p, C, nullptr, o.p, C));
        m.arena.release(mk);
        *out = o;
        return 0;
    }

    // ---- time embedding: sinusoid -> MLP -> all 22 per-resnet projections in one launch (App. A.2) ----
    int time_embed(const long long* t_dev, int t_scalar, int B) {
        const int rows = t_scalar ? 1 : B;
        const int c0 = m.cfg.block_out_channels[0], temb = 4 * c0;
        float* s = static_cast<float*>(alloc((size_t)rows * c0 * sizeof(float)));
        float* y1 = static_cast<float*>(alloc((size_t)rows * temb * sizeof(float)));
        float* emb = static_cast<float*>(alloc((size_t)rows * temb * sizeof(float)));
        m.tproj_out = static_cast<float*>(alloc((size_t)rows * m.tproj_total * sizeof(float)));
        if (!s || !y1 || !emb || !m.tproj_out) return 7;
        m.t_scalar = t_scalar;
        if (dry) return 0;
        TRY(launch_timestep_embedding(t_dev, t_scalar, s, rows, c0, st));
        TRY(launch_gemv_rows<T>(s, c0, m.te1.w, m.te1.b, y1, temb, rows, temb, c0, 0, st));
        TRY(launch_gemv_rows<T>(y1, temb, m.te2.w, m.te2.b, emb, temb, rows, temb, temb, 1, st));
        TRY(launch_gemv_rows<T>(emb, temb, m.tproj.w, m.tproj.b, m.tproj_out, m.tproj_total, rows, m.tproj_total, temb, 1, st));
        return 0;
    }

    int direct(const Act& x, const ConvW& cw, int stride, int act, const Act* add, Act* out) {
        const int pad = cw.ks / 2;
        const int Ho = (x.H + 2 * pad - cw.ks) / stride + 1, Wo = (x.W + 2 * pad - cw.ks) / stride + 1;
        *out = new_act(x.B, Ho, Wo, cw.cout);
        if (!out->p) return 7;
        MRISR_REQUIRE(cw.cin == x.C, "direct conv channel mismatch");
        DirectConvArgs a;
        a.x = x.p; a.w = cw.w; a.bias = cw.b; a.y = out->p; a.B = x.B; a.Hin = x.H; a.Win = x.W; a.Cin = x.C;
        a.Hout = Ho; a.Wout = Wo; a.Cout = cw.cout; a.ks = cw.ks; a.stride = stride; a.pad = pad; a.act = act;
        a.add = add ? add->p : nullptr;
        if (dry) return 0;
        return launch_direct_conv<T>(a, st);
    }

    // external tensor (NCHW any dtype, or NHWC compute dtype) -> NHWC T activation
    int import_act(const mrisr_tensor& t, Act* out, bool copy_if_nhwc) {
        MRISR_REQUIRE(t.ndim == 4, "expected a 4-D tensor");
        const int B = (int)t.shape[0], C = (int)t.shape[1], H = (int)t.shape[2], W = (int)t.shape[3];
        if (t.layout == MRISR_NHWC) {
            MRISR_REQUIRE(t.dtype == m.cfg.compute_dtype, "NHWC tensors must be in the compute dtype");
            if (!copy_if_nhwc) {
                out->p = t.data; out->B = B; out->H = H; out->W = W; out->C = C;
                return 0;
            }
        }
        *out = new_act(B, H, W, C);
        if (!out->p) return 7;
        if (dry) return 0;
        if (t.layout == MRISR_NHWC) {
            MRISR_CHECK_HIP(hipMemcpyAsync(out->p, t.data, out->numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
            return 0;
        }
        return launch_nchw_to_nhwc<T>(t.data, t.dtype, out->p, B, C, H, W, st);
    }
    int export_act(const Act& a, const mrisr_tensor& t, float scale) {
        MRISR_REQUIRE(t.ndim == 4 && t.shape[0] == a.B && t.shape[1] == a.C && t.shape[2] == a.H && t.shape[3] == a.W,
                      "output tensor shape mismatch");
        if (dry) return 0;
        if (t.layout == MRISR_NHWC) {
            MRISR_REQUIRE(t.dtype == m.cfg.compute_dtype && scale == 1.0f, "NHWC outputs: compute dtype, scale 1");
            MRISR_CHECK_HIP(hipMemcpyAsync(t.data, a.p, a.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
            return 0;
        }
        return launch_nhwc_to_nchw<T>(a.p, t.data, t.dtype, a.B, a.C, a.H, a.W, scale, st);
    }
    int add_external(Act& x, const mrisr_tensor& t) {
        const size_t mk = m.arena.mark();
        Act r;
        TRY(import_act(t, &r, false));
        MRISR_REQUIRE(r.B == x.B && r.C == x.C && r.H == x.H && r.W == x.W, "residual shape mismatch");
        if (!dry) TRY(launch_add_inplace<T>(x.p, r.p, (long long)x.numel(), st));
        m.arena.release(mk);
        return 0;
    }

    // ---- encoder shared by UNet and ControlNet: down blocks (+skips) ----
    int encoder(Act x, const mrisr_tensor* intrablock, int n_intra, std::vector<Act>* skips, Act* out) {
        skips->push_back(x);
        int ib = 0;
        for (int i = 0; i < m.cfg.num_levels; ++i) {
            Level& lv = m.down[i];
            const bool has_attn = !lv.xf.empty();
            for (size_t j = 0; j < lv.res.size(); ++j) {
                Act y;
                TRY(resnet(lv.res[j], x, nullptr, &y));
                x = y;
                if (has_attn) {
                    TRY(transformer(lv.xf[j], x, &y));
                    x = y;
                    if (j + 1 == lv.res.size() && ib < n_intra) TRY(add_external(x, intrablock[ib++]));
                }
                skips->push_back(x);
            }
            if (lv.has_down) {
                Act y;
                TRY(conv3(x, nullptr, lv.down, 2, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
                x = y;
                skips->push_back(x);
            }
            if (!has_attn && ib < n_intra) {
                // attention-free block: the adapter feature is added after the block returned, so the skips pushed
                // above must not see it -> work on a copy
                Act y = new_act(x.B, x.H, x.W, x.C);
                if (!y.p) return 7;
                if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(y.p, x.p, x.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
                x = y;
                TRY(add_external(x, intrablock[ib++]));
            }
        }
        *out = x;
        return 0;
    }
    int mid(Act x, Act* out) {
        Act y;
        TRY(resnet(m.mid_r0, x, nullptr, &y));
        x = y;
        TRY(transformer(m.mid_xf, x, &y));
        x = y;
        TRY(resnet(m.mid_r1, x, nullptr, &y));
        *out = y;
        return 0;
    }

    int set_context(const mrisr_tensor& ehs) {
        MRISR_REQUIRE(ehs.ndim == 3 && ehs.shape[2] == m.cfg.cross_attention_dim, "encoder_hidden_states shape");
        MRISR_REQUIRE((int)ehs.shape[0] == m.ws_B && (int)ehs.shape[1] == m.ctx_len, "context shape vs planned workspace");
        // [B, L, D] rows -> compute dtype (a [B*L, D, 1, 1] "image" through the boundary converter)
        if (!dry) TRY(launch_nchw_to_nhwc<T>(ehs.data, ehs.dtype, m.ctx_rows, m.ws_B * m.ctx_len, m.cfg.cross_attention_dim, 1, 1, st));
        auto each = [&](XfW& x) -> int { return project_context(x, m.head_buf_for_C(x.C)); };
        for (auto& lv : m.down) for (auto& x : lv.xf) TRY(each(x));
        TRY(each(m.mid_xf));
        for (auto& lv : m.up) for (auto& x : lv.xf) TRY(each(x));
        m.ctx_valid = true;
        return 0;
    }

    int unet_forward(const mrisr_tensor& sample, const long long* t_dev, int t_scalar, const mrisr_tensor* ehs,
                     const mrisr_tensor* down_res, int n_down, const mrisr_tensor* mid_res,
                     const mrisr_tensor* intrablock, int n_intra, const mrisr_tensor& out) {
        m.arena.reset();
        const int B = (int)sample.shape[0];
        if (ehs) TRY(set_context(*ehs));
        MRISR_REQUIRE(dry || m.ctx_valid, "no encoder_hidden_states given and none cached");
        TRY(time_embed(t_dev, t_scalar, B));
        Act s, x;
        TRY(import_act(sample, &s, false));
        TRY(direct(s, m.conv_in, 1, ACT_NONE, nullptr, &x));
        std::vector<Act> skips;
        TRY(encoder(x, intrablock, n_intra, &skips, &x));
        if (n_down > 0) {
            MRISR_REQUIRE(n_down == (int)skips.size(), "down_block_additional_residuals count");
            for (int k = 0; k < n_down; ++k) {
                // the encoder's own tensors also feed later encoder blocks, but those have all run: add in place.
                // Exception: the last skip IS the mid block's input, which must not see the residual -> copy it.
                if (skips[k].p == x.p) {
                    Act c = new_act(x.B, x.H, x.W, x.C);
                    if (!c.p) return 7;
                    if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(c.p, x.p, x.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
                    skips[k] = c;
                }
                TRY(add_external(skips[k], down_res[k]));
            }
        }
        TRY(mid(x, &x));
        if (mid_res) TRY(add_external(x, *mid_res));
        for (int i = 0; i < m.cfg.num_levels; ++i) {
            Level& lv = m.up[i];
            for (size_t j = 0; j < lv.res.size(); ++j) {
                Act sk = skips.back();
                skips.pop_back();
                Act y;
                TRY(resnet(lv.res[j], x, &sk, &y));
                x = y;
                if (!lv.xf.empty()) {
                    TRY(transformer(lv.xf[j], x, &y));
                    x = y;
                }
            }
            if (lv.has_up) {
                Act y;
                TRY(conv3(x, nullptr, lv.up, 1, 1, nullptr, 0, 1, nullptr, ACT_NONE, &y));
                x = y;
            }
        }
        Act xn, y;
        TRY(gn(x, nullptr, m.norm_out, true, m.cfg.norm_eps, &xn));
        // conv_out (Cout = 4): through the implicit GEMM (rows beyond N read zeros), 20x faster than the direct kernel
        if (m.conv_out.cout % 4 == 0) TRY(conv3(xn, nullptr, m.conv_out, 1, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
        else TRY(direct(xn, m.conv_out, 1, ACT_NONE, nullptr, &y));  // e.g. the 1-channel MNIST-plumbing config
        return export_act(y, out, 1.0f);
    }

    int set_cond(const mrisr_tensor& cond) {
        // ControlNet condition embedding (App. A.6): timestep-invariant; cached in m.cond_emb
        const size_t mk = m.arena.mark();
        Act e, y;
        TRY(import_act(cond, &e, false));
        for (size_t k = 0; k < m.ce.size(); ++k) {
            const bool last = k + 1 == m.ce.size();
            TRY(direct(e, m.ce[k], m.ce_stride[k], last ? ACT_NONE : ACT_SILU, nullptr, &y));
            e = y;
        }
        MRISR_REQUIRE(e.numel() * sizeof(T) <= m.cond_emb_bytes, "condition embedding larger than planned");
        if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(m.cond_emb, e.p, e.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
        m.arena.release(mk);
        m.cond_valid = true;
        return 0;
    }

    int controlnet_forward(const mrisr_tensor& sample, const long long* t_dev, int t_scalar, const mrisr_tensor* ehs,
                           const mrisr_tensor* cond, float scale, mrisr_tensor* down_out, int n_down,
                           mrisr_tensor* mid_out) {
        m.arena.reset();
        const int B = (int)sample.shape[0];
        if (ehs) TRY(set_context(*ehs));
        if (cond) TRY(set_cond(*cond));
        MRISR_REQUIRE(dry || (m.ctx_valid && m.cond_valid), "context / condition image neither given nor cached");
        TRY(time_embed(t_dev, t_scalar, B));
        Act s, x;
        TRY(import_act(sample, &s, false));
        Act ce;
        ce.p = m.cond_emb; ce.B = B; ce.H = s.H; ce.W = s.W; ce.C = m.cfg.block_out_channels[0];
        TRY(direct(s, m.conv_in, 1, ACT_NONE, &ce, &x));
        std::vector<Act> skips;
        TRY(encoder(x, nullptr, 0, &skips, &x));
        TRY(mid(x, &x));
        MRISR_REQUIRE(n_down == (int)skips.size(), "ControlNet output count");
        for (int k = 0; k < n_down; ++k) {
            const size_t mk = m.arena.mark();
            Act o = new_act(skips[k].B, skips[k].H, skips[k].W, skips[k].C);
            if (!o.p) return 7;
            TRY(linear(skips[k].p, (int)skips[k].rows(), skips[k].C, m.cn_down[k], ACT_NONE, nullptr, 0, nullptr, o.p, o.C));
            TRY(export_act(o, down_out[k], scale));
            m.arena.release(mk);
        }
        Act o = new_act(x.B, x.H, x.W, x.C);
        if (!o.p) return 7;
        TRY(linear(x.p, (int)x.rows(), x.C, m.cn_mid, ACT_NONE, nullptr, 0, nullptr, o.p, o.C));
        return export_act(o, *mid_out, scale);
    }
};

// ================================================================================================
// workspace planning
// ================================================================================================
const HeadBuf& Model::head_buf(int N, int C) const {
    for (auto& h : heads)
        if (h.N == N && h.H * h.hd == C) return h;
    return heads.front();  // unreachable after planning
}
const HeadBuf& Model::head_buf_for_C(int C) const {
    for (auto& h : heads)
        if (h.H * h.hd == C) return h;
    return heads.front();
}

template <typename T>
static int plan_t(Model& m, int B, int h, int w, int L, hipStream_t st) {
    const mrisr_unet_cfg& c = m.cfg;
    constexpr int BK = 128 / (int)sizeof(T);
    const bool flash = c.flash_attention && sizeof(T) == 2;
    m.heads.clear();
    m.ws_B = B; m.ws_h = h; m.ws_w = w; m.ctx_len = L;
    m.ctx_pad = round_up(L, 64);
    size_t persist = 0;
    auto carve = [&](size_t bytes) { size_t o = persist; persist = (persist + bytes + 255) & ~(size_t)255; return o; };
    // one (q, k, v^T) head-buffer set per distinct (tokens, channels): levels with attention + the mid block
    std::vector<std::pair<int, int>> keys;
    for (int i = 0; i < c.num_levels; ++i)
        if (c.attn_levels[i]) keys.push_back({(h >> i) * (w >> i), c.block_out_channels[i]});
    keys.push_back({(h >> (c.num_levels - 1)) * (w >> (c.num_levels - 1)), c.block_out_channels[c.num_levels - 1]});
    std::vector<size_t> offs;
    for (auto& kc : keys) {
        bool dup = false;
        for (auto& hb : m.heads) dup |= (hb.N == kc.first && hb.H * hb.hd == kc.second);
        if (dup) continue;
        HeadBuf hb;
        hb.B = B; hb.H = c.num_heads; hb.N = kc.first; hb.hd = kc.second / c.num_heads;
        hb.npad = round_up(hb.N, 64);
        hb.dpad = round_up(hb.hd, flash ? 32 : BK);
        const size_t qk = (size_t)B * hb.H * hb.npad * hb.dpad * sizeof(T);
        offs.push_back(carve(qk)); offs.push_back(carve(qk)); offs.push_back(carve(qk));
        m.heads.push_back(hb);
    }
    // cross-attention caches per transformer block
    std::vector<XfW*> xfs;
    for (auto& lv : m.down) for (auto& x : lv.xf) xfs.push_back(&x);
    xfs.push_back(&m.mid_xf);
    for (auto& lv : m.up) for (auto& x : lv.xf) xfs.push_back(&x);
    std::vector<size_t> xoffs;
    for (XfW* x : xfs) {
        const HeadBuf& hb = m.head_buf_for_C(x->C);
        const size_t kb = (size_t)B * hb.H * m.ctx_pad * hb.dpad * sizeof(T);
        xoffs.push_back(carve(kb)); xoffs.push_back(carve(kb));
    }
    const size_t ctx_off = carve((size_t)B * L * c.cross_attention_dim * sizeof(T));
    m.cond_emb_bytes = (size_t)B * h * w * c.block_out_channels[0] * sizeof(T);
    const size_t cond_off = carve(m.cond_emb_bytes);
    TRY(m.persist.reserve(persist, false));
    MRISR_CHECK_HIP(hipMemsetAsync(m.persist.p, 0, persist, st));  // zero pads of the head buffers
    char* base = static_cast<char*>(m.persist.p);
    for (size_t i = 0; i < m.heads.size(); ++i) {
        m.heads[i].q = base + offs[3 * i];
        m.heads[i].k = base + offs[3 * i + 1];
        m.heads[i].vt = base + offs[3 * i + 2];
    }
    for (size_t i = 0; i < xfs.size(); ++i) {
        xfs[i]->kc = base + xoffs[2 * i];
        xfs[i]->vtc = base + xoffs[2 * i + 1];
    }
    m.ctx_rows = base + ctx_off;
    m.cond_emb = base + cond_off;
    m.ctx_valid = false;
    m.cond_valid = false;
    // dry pass -> exact arena size
    m.arena.dry = true;
    m.arena.reset();
    m.arena.peak = 0;
    Runner<T> r(m, st, true);
    mrisr_tensor sample{}; sample.ndim = 4; sample.dtype = MRISR_F32;
    sample.shape[0] = B; sample.shape[1] = c.in_channels; sample.shape[2] = h; sample.shape[3] = w;
    mrisr_tensor ehs{}; ehs.ndim = 3; ehs.dtype = MRISR_F32; ehs.shape[0] = B; ehs.shape[1] = L; ehs.shape[2] = c.cross_attention_dim;
    int rc;
    const int ns = m.num_skips();
    std::vector<mrisr_tensor> res(ns + 1);
    for (int k = 0; k <= ns; ++k) {
        res[k] = mrisr_tensor{};
        res[k].ndim = 4; res[k].dtype = MRISR_F32; res[k].layout = MRISR_NCHW;
        m.skip_shape(k, B, h, w, res[k].shape);
    }
    if (!m.is_controlnet) {
        mrisr_tensor out = sample; out.shape[1] = c.out_channels;
        // worst case: ControlNet residuals + adapter features present
        std::vector<mrisr_tensor> intra(c.num_levels);
        for (int i = 0; i < c.num_levels; ++i) {
            intra[i] = res[0];
            intra[i].shape[1] = c.block_out_channels[i]; intra[i].shape[2] = h >> i; intra[i].shape[3] = w >> i;
        }
        rc = r.unet_forward(sample, nullptr, 0, &ehs, res.data(), ns, &res[ns], intra.data(), c.num_levels, out);
        if (!rc) rc = r.unet_forward(sample, nullptr, 1, &ehs, res.data(), ns, &res[ns], intra.data(), c.num_levels, out);
    } else {
        mrisr_tensor cond = sample; cond.shape[1] = c.cond_channels; cond.shape[2] = 8 * h; cond.shape[3] = 8 * w;
        rc = r.controlnet_forward(sample, nullptr, 0, &ehs, &cond, 1.0f, res.data(), ns, &res[ns]);
    }
    m.arena.dry = false;
    m.ctx_valid = false;
    m.cond_valid = false;
    if (rc) return rc;
    TRY(m.arena.buf.reserve(m.arena.peak + 4096, false));
    m.arena.reset();
    return 0;
}

int Model::ensure_workspace(int B, int h, int w, int L, hipStream_t st) {
    MRISR_REQUIRE(finalized, "call mrisr_model_finalize first");
    const int div = 1 << (cfg.num_levels - 1);
    MRISR_REQUIRE(h % div == 0 && w % div == 0, "latent size must be divisible by 2^(levels-1)");
    char key[96];
    snprintf(key, sizeof(key), "%d,%d,%d,%d", B, h, w, L);
    if (ws_key == key) return 0;
    int rc = cfg.compute_dtype == MRISR_F32 ? plan_t<float>(*this, B, h, w, L, st) : plan_t<bf16>(*this, B, h, w, L, st);
    if (rc) return rc;
    ws_key = key;
    return 0;
}

static int check_sample(const Model& m, const mrisr_tensor* s) {
    MRISR_REQUIRE(s && s->ndim == 4 && s->shape[1] == m.cfg.in_channels, "sample must be [B, in_channels, h, w]");
    return 0;
}

int Model::forward_unet(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                        const mrisr_tensor* down_res, int n_down, const mrisr_tensor* mid_res,
                        const mrisr_tensor* intrablock, int n_intra, mrisr_tensor* out, hipStream_t st) {
    MRISR_REQUIRE(!is_controlnet, "not a UNet handle");
    TRY(check_sample(*this, sample));
    MRISR_REQUIRE(timestep && timestep->dtype == MRISR_I64 && timestep->ndim <= 1, "timestep: device int64, 0-dim or [B]");
    MRISR_REQUIRE(out, "out tensor");
    const int B = (int)sample->shape[0], h = (int)sample->shape[2], w = (int)sample->shape[3];
    const int L = ehs ? (int)ehs->shape[1] : ctx_len;
    MRISR_REQUIRE(L > 0, "no encoder_hidden_states given and none cached");
    TRY(ensure_workspace(B, h, w, L, st));
    const int t_scalar = timestep->ndim == 0 || timestep->shape[0] == 1;
    MRISR_REQUIRE(t_scalar || timestep->shape[0] == B, "timestep length");
    const long long* t = static_cast<const long long*>(timestep->data);
    if (cfg.compute_dtype == MRISR_F32) {
        Runner<float> r(*this, st, false);
        return r.unet_forward(*sample, t, t_scalar, ehs, down_res, n_down, mid_res, intrablock, n_intra, *out);
    }
    Runner<bf16> r(*this, st, false);
    return r.unet_forward(*sample, t, t_scalar, ehs, down_res, n_down, mid_res, intrablock, n_intra, *out);
}

int Model::forward_controlnet(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs,
                              const mrisr_tensor* cond, float scale, mrisr_tensor* down_out, int n_down,
                              mrisr_tensor* mid_out, hipStream_t st) {
    MRISR_REQUIRE(is_controlnet, "not a ControlNet handle");
    TRY(check_sample(*this, sample));
    MRISR_REQUIRE(timestep && timestep->dtype == MRISR_I64 && timestep->ndim <= 1, "timestep: device int64, 0-dim or [B]");
    const int B = (int)sample->shape[0], h = (int)sample->shape[2], w = (int)sample->shape[3];
    const int L = ehs ? (int)ehs->shape[1] : ctx_len;
    MRISR_REQUIRE(L > 0, "no encoder_hidden_states given and none cached");
    if (cond) MRISR_REQUIRE(cond->ndim == 4 && cond->shape[2] == 8 * h && cond->shape[3] == 8 * w, "controlnet_cond must be [B,3,8h,8w]");
    TRY(ensure_workspace(B, h, w, L, st));
    const int t_scalar = timestep->ndim == 0 || timestep->shape[0] == 1;
    const long long* t = static_cast<const long long*>(timestep->data);
    if (cfg.compute_dtype == MRISR_F32) {
        Runner<float> r(*this, st, false);
        return r.controlnet_forward(*sample, t, t_scalar, ehs, cond, scale, down_out, n_down, mid_out);
    }
    Runner<bf16> r(*this, st, false);
    return r.controlnet_forward(*sample, t, t_scalar, ehs, cond, scale, down_out, n_down, mid_out);
}

int Model::set_context(const mrisr_tensor* ehs, int B, int h, int w, hipStream_t st) {
    MRISR_REQUIRE(ehs && ehs->ndim == 3, "encoder_hidden_states must be [B, L, D]");
    TRY(ensure_workspace(B, h, w, (int)ehs->shape[1], st));
    arena.reset();
    if (cfg.compute_dtype == MRISR_F32) {
        Runner<float> r(*this, st, false);
        return r.set_context(*ehs);
    }
    Runner<bf16> r(*this, st, false);
    return r.set_context(*ehs);
}

int Model::set_cond(const mrisr_tensor* cond, int L, hipStream_t st) {
    MRISR_REQUIRE(is_controlnet && cond && cond->ndim == 4, "controlnet_cond must be [B,3,8h,8w]");
    TRY(ensure_workspace((int)cond->shape[0], (int)cond->shape[2] / 8, (int)cond->shape[3] / 8, L, st));
    arena.reset();
    if (cfg.compute_dtype == MRISR_F32) {
        Runner<float> r(*this, st, false);
        return r.set_cond(*cond);
    }
    Runner<bf16> r(*this, st, false);
    return r.set_cond(*cond);
}

}  // namespace mrisr
