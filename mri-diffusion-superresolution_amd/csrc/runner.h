// Runner<T>: the forward pass of UNet2DConditionModel / ControlNetModel as a launch sequence over the model's arena
// (shared by model.hip - inference - and train.hip - the LoRA fine-tuning step, which records what its backward needs).
#pragma once
#include <climits>
#include <cmath>
#include <cstdlib>

#include "model.h"

namespace mrisr {

extern int g_subpix_override;  // test hook (mrisr_debug_subpix): -1 = MRISR_SUBPIX / default, 0 = off, n = minimum low-resolution rows

#ifndef TRY
#define TRY(expr)              \
    do {                       \
        int _rc = (expr);      \
        if (_rc) return _rc;   \
    } while (0)
#endif

// MRISR_LORA_INKERNEL=0 restores the separate down-projection pass (A/B measurements)
// MRISR_FUSE_LN=0: LayerNorm always as its own launch (A/B measurements)
inline bool fuse_ln() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRISR_FUSE_LN"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
inline bool lora_in_kernel() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRISR_LORA_INKERNEL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
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

    int gn(const Act& x0, const Act* x1, const NormW& nw, bool silu, float eps, Act* out, float** partial_out = nullptr,
           int* nsplit_out = nullptr) {
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
        if (partial_out) *partial_out = a.partial;
        if (nsplit_out) *nsplit_out = a.nsplit;
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

    // conv3x3 whose only consumer is a GroupNorm (+ SiLU): when the autotuner splits K for it, the f32 slabs are summed by the GroupNorm
    // kernel itself (launch_groupnorm_slabs: the reduce's operations in its order, then the normalisation from registers) - one launch
    // instead of splitk_reduce + GroupNorm.  raw (optional): also store the un-normalised output (something else reads it).
    int conv3_gn(const Act& x, const Act* x1, const ConvW& cw, const float* rowvec, int rowvec_ld, int rowvec_div, const Act* resid,
                 const NormW& nw, bool silu, float eps, Act* raw, Act* out) {
        const int Cin = x.C + (x1 ? x1->C : 0);
        MRISR_REQUIRE(cw.cin == Cin && cw.ks == 3 && nw.c == cw.cout, "conv3x3 + GroupNorm weight mismatch");
        GemmArgs g;
        g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
        if (x1) { g.a1 = x1->p; g.c1 = x1->C; g.lda1 = x1->C; }
        g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W; g.stride = 1; g.ups = 0;
        g.w = cw.w; g.M = x.B * x.H * x.W; g.N = cw.cout; g.K = 9 * Cin;
        g.bias = cw.b; g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rowvec_div = rowvec_div; g.act = ACT_NONE;
        if (resid) { g.resid = resid->p; g.ldr = resid->C; }
        g.ldo = cw.cout;
        bool fuse = sizeof(T) == 2 && !m.keep && groupnorm_slabs_ok(cw.cout, m.cfg.norm_num_groups, x.H * x.W);
        if (fuse) {
            g.out = reinterpret_cast<void*>(0x1000);  // (never written: the slabs are the output) - the planner only looks at the signature
            TRY(gemm_choose(g, true));
            fuse = g.splitk > 1;
        }
        if (!fuse) {
            Act h;
            TRY(conv3(x, x1, cw, 1, 0, rowvec, rowvec_ld, rowvec_div, resid, ACT_NONE, &h));
            if (raw) *raw = h;
            return gn(h, nullptr, nw, silu, eps, out);
        }
        Act h;
        h.B = x.B; h.H = x.H; h.W = x.W; h.C = cw.cout; h.p = nullptr;
        if (raw) { h = new_act(x.B, x.H, x.W, cw.cout); if (!h.p) return 7; *raw = h; }
        *out = new_act(x.B, x.H, x.W, cw.cout);
        if (!out->p) return 7;
        GroupNormArgs a;
        a.x0 = nullptr; a.c0 = cw.cout; a.B = x.B; a.HW = x.H * x.W; a.groups = m.cfg.norm_num_groups; a.eps = eps;
        a.gamma = nw.g; a.beta = nw.b; a.silu = silu ? 1 : 0; a.y = out->p;
        a.nsplit = groupnorm_nsplit(a.B, a.HW);
        a.partial = static_cast<float*>(alloc((size_t)a.B * a.nsplit * a.groups * 2 * sizeof(float)));
        g.partial = static_cast<float*>(alloc((size_t)g.splitk * g.M * g.N * sizeof(float)));
        if (!a.partial || !g.partial) return 7;
        if (dry) return 0;
        g.defer_reduce = 1;
        TRY(launch_gemm<T>(g, st));
        GnSlabSrc ss;
        ss.partial = g.partial; ss.splitk = g.splitk; ss.alpha = g.alpha; ss.bias = g.bias; ss.rowvec = g.rowvec; ss.rowvec_ld = g.rowvec_ld;
        ss.rowvec_div = g.rowvec_div; ss.resid = g.resid; ss.ldr = g.ldr; ss.raw_out = h.p;
        return launch_groupnorm_slabs(a, ss, st);
    }

    // Upsample2D: nearest x2, then conv3x3 (diffusers).  Every output pixel (2y + py, 2x + px) reads only a
    // 2 x 2 window of the LOW-resolution input (the three up-sampled rows it touches are two input rows), so the layer is four 2 x 2
    // convs - one per output parity, filter taps pre-summed at load (launch_pack_conv_subpix) - at 4/9 of the MACs: one batched implicit
    // GEMM (z = parity: its own bank, its own window origin; M = B h w rows each) into four parity planes, then one interleaving pass.
    // bf16 inference only (the f32 parity engine and the training graph keep the literal form); MRISR_SUBPIX=0 / a minimum row count.
    static int subpix_min_rows() {
        static const int v = [] { const char* e = getenv("MRISR_SUBPIX"); return e ? atoi(e) : 2048; }();  // 0: off; n: from n low-resolution rows on
        return g_subpix_override >= 0 ? g_subpix_override : v;
    }
    int upsample_conv(const Act& x, const Level& lv, Act* out) {
        const int minr = subpix_min_rows();
        if (!(sizeof(T) == 2 && lv.up_sp && !m.keep && minr > 0 && (long long)x.rows() >= minr))
            return conv3(x, nullptr, lv.up, 1, 1, nullptr, 0, 1, nullptr, ACT_NONE, out);
        const ConvW& cw = lv.up;
        MRISR_REQUIRE(cw.cin == x.C, "upsampler weight mismatch");
        *out = new_act(x.B, 2 * x.H, 2 * x.W, cw.cout);
        if (!out->p) return 7;
        const size_t mk = m.arena.mark();
        const long long plane = (long long)x.rows() * cw.cout;
        T* planes = static_cast<T*>(alloc((size_t)4 * plane * sizeof(T)));
        if (!planes) return 7;
        GemmArgs g;
        g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
        g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W; g.stride = 1; g.ups = 0;
        g.kw = 2; g.subpix = 1; g.batch = 4; g.a_bs = 0; g.w_bs = (long long)cw.cout * 4 * cw.cin; g.o_bs = plane;
        g.w = lv.up_sp; g.M = (int)x.rows(); g.N = cw.cout; g.K = 4 * x.C; g.bias = cw.b;
        g.out = planes; g.ldo = cw.cout;
        g.alg_flops = 2.0 * 4.0 * g.M * (double)g.N * g.K;
        TRY(run_gemm(g));
        if (!dry) TRY(launch_subpix_shuffle<T>(planes, out->p, x.B, x.H, x.W, cw.cout, st));
        m.arena.release(mk);
        return 0;
    }

    // y[M][n] = x[M][k] W^T (+LoRA tail) + bias ...; x given as raw rows
    //   ln / ln_scratch: y = LayerNorm(x) W^T ...  The normalisation runs as a prologue of the row-panel kernel on its
    //   register-resident rows when that kernel takes the shape (bf16, K = 320 / 640); otherwise as its own launch into
    //   ln_scratch ([M][k] of T), which then feeds the GEMM.
    int linear(const void* x, int M, int lda, const LinW& lw, int act, const void* resid, int ldr, GemmArgs* custom,
               void* out, int ldo, float** z_out = nullptr, const NormW* ln = nullptr, void* ln_scratch = nullptr) {
        GemmArgs g = custom ? *custom : GemmArgs();
        const size_t mk = m.arena.mark();
        g.a0 = x; g.c0 = lw.k; g.lda0 = lda;
        g.w = lw.w; g.M = M; g.N = lw.n; g.K = lw.k;
        g.alg_flops = 2.0 * M * (double)lw.n * (lw.k + lw.r);
        g.bias = lw.b; g.act = act; g.resid = resid; g.ldr = ldr;
        if (g.out_mode != OUT_HEADS) { g.out = out; g.ldo = ldo; }
        // the row-panel kernel fuses LoRA only in the in-kernel rank-4 form
        if (lw.R && !(lw.r == 4 && lw.R <= 16 && lora_in_kernel())) g.no_rp = 1;
        // fp8 operands (cfg.fp8_linears; the training forward only with cfg.fp8_train): when the row-panel fp8 kernel takes this launch
        if (lw.w8 && sizeof(T) == 2 && (!m.keep || m.cfg.fp8_train) && !g.no_rp) {
            GemmArgs probe = g;
            probe.w8 = lw.w8; probe.w_scale = lw.w_scale;
            if (lw.R) { probe.lora_a = lw.loraA; probe.lora_b = lw.loraB; probe.lora_r = lw.r; probe.lora_R = lw.R; probe.lora_a8 = lw.loraA8; probe.lora_a_scale = lw.loraA_scale; }
            if (gemm_rp_tile(probe)) { g.w8 = lw.w8; g.w_scale = lw.w_scale; g.lora_a8 = lw.loraA8; g.lora_a_scale = lw.loraA_scale; }
        }
        if (ln) {
            MRISR_REQUIRE(ln->c == lw.k && lda == lw.k, "LayerNorm width vs the projection's K");
            GemmArgs probe = g;  // eligibility as the launch will see it (the in-kernel LoRA form, rank 4, is part of it)
            if (lw.R) { probe.lora_a = lw.loraA; probe.lora_b = lw.loraB; probe.lora_r = lw.r; probe.lora_R = lw.R; }
            if (sizeof(T) == 2 && !m.keep && fuse_ln() && (!lw.R || (lw.R <= 16 && lora_in_kernel())) && gemm_rp_tile(probe)) {
                g.ln_gamma = ln->g; g.ln_beta = ln->b; g.ln_eps = 1e-5f;
            } else {
                MRISR_REQUIRE(ln_scratch, "LayerNorm scratch rows");
                TRY(layernorm(x, *ln, M, lw.k, ln_scratch));
                g.a0 = x = ln_scratch;
            }
        }
        if (!lw.R) {
            TRY(run_gemm(g));
            if (!m.keep) m.arena.release(mk);
            return 0;
        }
        // LoRA: the rank-r up-projection (alpha/r) B z is accumulated in the projection GEMM's epilogue.  z = x A^T is
        // computed inside the same kernel (bf16 buffer-addressed tiles, un-split); otherwise by one bandwidth-bound
        // pass over x (launch_lora_down) that hands z over through HBM.
        g.lora_zld = lw.R; g.lora_b = lw.loraB; g.lora_r = lw.r; g.lora_secN = lw.secN;
        TRY(gemm_choose(g, sizeof(T) == 2));
        float* z = static_cast<float*>(alloc((size_t)M * lw.R * sizeof(float)));
        if (!z) return 7;
        if (z_out) *z_out = z;
        const bool in_kernel = sizeof(T) == 2 && lw.R <= 16 && g.splitk == 1 && g.tile >= 14 && lora_in_kernel() && (g.tile < 60 || lw.r == 4);
        MRISR_REQUIRE(in_kernel || !g.ln_gamma, "fused LayerNorm needs the in-kernel LoRA form");
        if (in_kernel) {
            g.lora_a = lw.loraA; g.lora_R = lw.R; g.lora_zout = z_out ? z : nullptr;
        } else {
            if (!dry) TRY(launch_lora_down<T>(x, lda, lw.loraA, z, M, lw.k, lw.R, st));
            g.lora_z = z;
        }
        if (g.splitk > 1) {
            g.partial = static_cast<float*>(alloc((size_t)g.splitk * g.batch * g.M * g.N * sizeof(float)));
            if (!g.partial) return 7;
        }
        if (!dry) TRY(launch_gemm<T>(g, st));
        if (!m.keep) m.arena.release(mk);
        return 0;
    }

    int layernorm(const void* x, const NormW& nw, int M, int C, void* y) {
        if (dry) return 0;
        return launch_layernorm<T>(x, y, nw.g, nw.b, M, C, 1e-5f, st);
    }

    // ---- ResnetBlock2D (App. A.3) ----
    // next / next_out: the GroupNorm of the block that consumes this resnet's output and nothing else in between (the transformer's norm).
    // When conv2 is K-split, its slabs are summed by that GroupNorm's kernel (launch_groupnorm_slabs: + bias + residual, the raw output
    // stored as well) and next_out receives the normalised tensor; otherwise next_out->p stays null and the consumer normalises itself.
    int resnet(const ResW& r, const Act& x, const Act* x1, Act* out, const NormW* next = nullptr, float next_eps = 0.f, Act* next_out = nullptr) {
        Act o = new_act(x.B, x.H, x.W, r.cout);  // allocated first: survives the temporaries below
        if (!o.p) return 7;
        if (next_out) next_out->p = nullptr;
        GemmArgs g2;  // conv2, planned before anything else is allocated: whether the fused form applies decides what must outlive this block
        bool fuse2 = false;
        GroupNormArgs a2;
        if (next && next_out && sizeof(T) == 2 && !m.keep && next->c == r.cout && groupnorm_slabs_ok(r.cout, m.cfg.norm_num_groups, x.H * x.W)) {
            g2.c0 = r.cout; g2.lda0 = r.cout;
            g2.conv = 1; g2.B = x.B; g2.Hin = x.H; g2.Win = x.W; g2.Hout = x.H; g2.Wout = x.W;
            g2.w = r.c2.w; g2.M = (int)x.rows(); g2.N = r.cout; g2.K = 9 * r.cout; g2.bias = r.c2.b;
            g2.a0 = reinterpret_cast<void*>(0x1000); g2.resid = reinterpret_cast<void*>(0x1000); g2.ldr = r.cout;
            g2.out = reinterpret_cast<void*>(0x1000); g2.ldo = r.cout;
            TRY(gemm_choose(g2, true));
            fuse2 = g2.splitk > 1;
            if (fuse2) {
                *next_out = new_act(x.B, x.H, x.W, r.cout);  // outside the mark / release scope below: the consumer reads it
                a2.nsplit = groupnorm_nsplit(x.B, x.H * x.W);
                a2.partial = static_cast<float*>(alloc((size_t)x.B * a2.nsplit * m.cfg.norm_num_groups * 2 * sizeof(float)));
                if (!next_out->p || !a2.partial) return 7;
            }
        }
        const size_t mk = m.arena.mark();
        Act xn, h, hn;
        TRY(gn(x, x1, r.n1, true, m.cfg.norm_eps, &xn));
        const int div = m.t_scalar ? INT_MAX : x.H * x.W;
        TRY(conv3_gn(xn, nullptr, r.c1, m.tproj_out + r.temb_off, m.tproj_total, div, nullptr, r.n2, true, m.cfg.norm_eps, nullptr, &hn));
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
        if (fuse2) {
            g2.a0 = hn.p; g2.resid = res.p; g2.out = o.p;
            g2.partial = static_cast<float*>(alloc((size_t)g2.splitk * g2.M * g2.N * sizeof(float)));
            if (!g2.partial) return 7;
            if (!dry) {
                g2.defer_reduce = 1;
                TRY(launch_gemm<T>(g2, st));
                a2.x0 = nullptr; a2.c0 = r.cout; a2.B = x.B; a2.HW = x.H * x.W; a2.groups = m.cfg.norm_num_groups; a2.eps = next_eps;
                a2.gamma = next->g; a2.beta = next->b; a2.silu = 0; a2.y = next_out->p;
                GnSlabSrc ss;
                ss.partial = g2.partial; ss.splitk = g2.splitk; ss.alpha = g2.alpha; ss.bias = g2.bias; ss.resid = res.p; ss.ldr = r.cout; ss.raw_out = o.p;
                TRY(launch_groupnorm_slabs(a2, ss, st));
            }
        } else {
            GemmArgs g;
            g.a0 = hn.p; g.c0 = hn.C; g.lda0 = hn.C;
            g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W;
            g.w = r.c2.w; g.M = (int)x.rows(); g.N = r.cout; g.K = 9 * r.cout; g.bias = r.c2.b;
            g.resid = res.p; g.ldr = r.cout; g.out = o.p; g.ldo = r.cout;
            TRY(run_gemm(g));
        }
        if (!m.keep) m.arena.release(mk);
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
            if (m.cfg.fp8_attention) {  // BASELINE configs[4]: e4m3 Q K^T and P V; scratch for the quantised K / V^T + per-head scales
                const size_t mk8 = m.arena.mark();
                a.k8 = alloc((size_t)BH * nkpad * hb.dpad);
                a.vt8 = alloc((size_t)BH * nkpad * hb.dpad);
                a.f8_scales = static_cast<float*>(alloc((size_t)BH * 4 * sizeof(float)));
                if (!a.k8 || !a.vt8 || !a.f8_scales) return 7;
                const int rc = dry ? 0 : launch_attention_fp8(a, st);
                if (!m.keep) m.arena.release(mk8);
                return rc;
            }
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
        if (!m.keep) m.arena.release(mk);
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
        TRY(linear(m.ctx_rows, m.ws_B * m.ctx_len, m.cfg.cross_attention_dim, x.kv2, ACT_NONE, nullptr, 0, &g,
                   nullptr, 0));
        // the same K / V once more as the LDS images of the fused middle (xtail.hip); cheap, so packed whether or not that path is on
        if (x.kvp && !dry && m.ctx_len <= 80 && hb.hd == 40 && sizeof(T) == 2)
            TRY(launch_pack_xattn_kv(x.kc, x.vtc, x.kvp, m.ws_B, m.ctx_pad, hb.dpad, m.ctx_len, st));
        return 0;
    }

    // attn1.to_out + residual, LayerNorm2, attn2.to_q, cross-attention, attn2.to_out + residual as ONE launch (xtail.hip) when the
    // block has the geometry that kernel is written for; false: the four launches below
    bool fused_middle(const XfW& xw, const HeadBuf& hb, int M) const {
        if (sizeof(T) != 2 || m.keep || !xw.q2p || !xw.kvp || !xattn_tail_enabled() || m.cfg.fp8_attention || !m.cfg.flash_attention) return false;
        if (!xattn_tail_ok(xw.C, hb.H, M, hb.N, m.ctx_len) || hb.hd != 40) return false;
        const int R = xw.out1.R;
        if (xw.q2.R != R || xw.out2.R != R) return false;
        if (R && !(R == 4 && xw.out1.r == 4 && xw.q2.r == 4 && xw.out2.r == 4 && lora_in_kernel())) return false;
        if (xw.out1.w8 || xw.q2.w8 || xw.out2.w8) return false;
        return true;
    }

    // ---- Transformer2DModel with one BasicTransformerBlock (App. A.4) ----
    int transformer(XfW& xw, const Act& x, Act* out, const Act* pre_normed = nullptr) {
        const int C = xw.C, M = (int)x.rows();
        const HeadBuf& hb = m.head_buf(x.H * x.W, C);
        Act o = new_act(x.B, x.H, x.W, C);
        if (!o.p) return 7;
        const size_t mk = m.arena.mark();
        Act xn;
        if (pre_normed && pre_normed->p) xn = *pre_normed;  // (the producing resnet's conv2 epilogue already normalised it: resnet())
        else TRY(gn(x, nullptr, xw.norm, false, 1e-6f, &xn));
        T* t = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        T* nrm = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        T* ao = static_cast<T*>(alloc((size_t)M * C * sizeof(T)));
        if (!t || !nrm || !ao) return 7;
        TRY(linear(xn.p, M, C, xw.proj_in, ACT_NONE, nullptr, 0, nullptr, t, C));
        // self-attention
        {
            GemmArgs g;
            heads_args(&g, hb, hb.q, 0, hb.k, 0, hb.vt, 1, hb.N, hb.npad);
            TRY(linear(t, M, C, xw.qkv, ACT_NONE, nullptr, 0, &g, nullptr, 0, nullptr, &xw.ln1, nrm));
        }
        TRY(attention(hb, hb.k, hb.vt, hb.N, hb.npad, ao));
        if (fused_middle(xw, hb, M)) {
            XTailArgs a;
            a.ao = ao; a.ldao = C; a.t = t; a.ldt = C; a.M = M; a.ntok = hb.N;
            a.w1 = xw.out1.w; a.b1 = xw.out1.b; a.a1 = xw.out1.loraA; a.lb1 = xw.out1.loraB;
            a.ln_g = xw.ln2.g; a.ln_b = xw.ln2.b; a.ln_eps = 1e-5f;
            a.wq = xw.q2p; a.bq = xw.q2.b; a.aq = xw.q2.loraA; a.lbq = xw.q2.loraB;
            a.kvp = xw.kvp; a.nk = m.ctx_len; a.scale = 1.0f / sqrtf((float)hb.hd);
            a.w2 = xw.out2p; a.b2 = xw.out2.b; a.a2 = xw.out2.loraA; a.lb2 = xw.out2.loraB;
            a.lora_r = xw.out1.R ? 4 : 0;
            if (!dry) TRY(launch_xattn_tail(a, st));
        } else {
        TRY(linear(ao, M, C, xw.out1, ACT_NONE, t, C, nullptr, t, C));
        // cross-attention (K/V cached by set_context)
        {
            GemmArgs g;
            heads_args(&g, hb, hb.q, 0, nullptr, 0, nullptr, 0, hb.N, hb.npad);
            TRY(linear(t, M, C, xw.q2, ACT_NONE, nullptr, 0, &g, nullptr, 0, nullptr, &xw.ln2, nrm));
        }
        TRY(attention(hb, xw.kc, xw.vtc, m.ctx_len, m.ctx_pad, ao));
        TRY(linear(ao, M, C, xw.out2, ACT_NONE, t, C, nullptr, t, C));
        }
        // GEGLU feed-forward
        bool proj_done = false;
        if (xw.ff2p && !m.keep && mlp_fused_ok(C, 4 * C, C)) {
            MlpArgs a;
            a.x = t; a.ldx = C; a.M = M; a.ln_gamma = xw.ln3.g; a.ln_beta = xw.ln3.b; a.ln_eps = 1e-5f;
            a.w1 = xw.ff1.w; a.b1 = xw.ff1.b; a.w2p = xw.ff2p; a.b2 = xw.ff2.b;
            a.resid = t; a.ldr = C; a.out = t; a.ldo = C; a.C = C; a.H = 4 * C; a.N2 = C;
            if (xw.proj_outp && M % 128 == 0 && mlp_proj_enabled()) {  // proj_out + outer residual inside the same kernel
                a.wp = xw.proj_outp; a.bp = xw.proj_out.b; a.xres = x.p; a.ldxr = C; a.out2 = o.p; a.ldo2 = C;
                proj_done = true;
            }
            if (!dry) TRY(launch_mlp_fused(a, st));
        } else {
            T* ff = static_cast<T*>(alloc((size_t)M * 4 * C * sizeof(T)));
            if (!ff) return 7;
            TRY(linear(t, M, C, xw.ff1, ACT_GEGLU, nullptr, 0, nullptr, ff, 4 * C, nullptr, &xw.ln3, nrm));
            TRY(linear(ff, M, 4 * C, xw.ff2, ACT_NONE, t, C, nullptr, t, C));
        }
        if (!proj_done) TRY(linear(t, M, C, xw.proj_out, ACT_NONE, x.p, C, nullptr, o.p, C));
        if (!m.keep) m.arena.release(mk);
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
        m.te_s = s; m.te_y1 = y1; m.te_emb = emb;
        if (dry) return 0;
        if (m.tproj_table && t_scalar && !m.keep)  // fused sampler: this step's row of the per-run table (Model::build_tproj_table)
            return launch_select_row(m.tproj_table, m.tproj_step, m.tproj_first, m.tproj_out, m.tproj_total, st);
        TRY(launch_timestep_embedding(t_dev, t_scalar, s, rows, c0, st));
        TRY(launch_gemv_rows<T>(s, c0, m.te1.w, m.te1.b, y1, temb, rows, temb, c0, 0, st));
        TRY(launch_gemv_rows<T>(y1, temb, m.te2.w, m.te2.b, emb, temb, rows, temb, temb, 1, st));
        TRY(launch_gemv_rows<T>(emb, temb, m.tproj.w, m.tproj.b, m.tproj_out, m.tproj_total, rows, m.tproj_total, temb, 1, st));
        return 0;
    }

    // the same for `rows` timesteps at once (rows of the table): sinusoid -> MLP -> projections, weights read once per 64 rows
    int time_embed_table(const long long* ts_dev, int rows, float* scratch, float* table) {
        const int c0 = m.cfg.block_out_channels[0], temb = 4 * c0;
        for (int r0 = 0; r0 < rows; r0 += 64) {
            const int nr = std::min(64, rows - r0);
            float* s = scratch;
            float* y1 = s + (size_t)64 * c0;
            float* emb = y1 + (size_t)64 * temb;
            TRY(launch_timestep_embedding(ts_dev + r0, 0, s, nr, c0, st));
            // (many-row matrix-core form: inputs rounded to bf16 - the generic f32-input kernel re-reads the rows for every output column and took
            // ~1.3 ms per run for 50 x 19,840 outputs, more than the table saves; the bf16 engine's latents move by the same ~2e-3 either way)
            TRY(launch_gemv_rows<T>(s, c0, m.te1.w, m.te1.b, y1, temb, nr, temb, c0, 0, st));
            TRY(launch_gemv_rows<T>(y1, temb, m.te2.w, m.te2.b, emb, temb, nr, temb, temb, 1, st));
            TRY(launch_gemv_rows<T>(emb, temb, m.tproj.w, m.tproj.b, table + (size_t)r0 * m.tproj_total, m.tproj_total, nr, m.tproj_total, temb, 1, st));
        }
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
        if (!m.keep) m.arena.release(mk);
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
                Act y, xn_next;
                xn_next.p = nullptr;
                if (has_attn) TRY(resnet(lv.res[j], x, nullptr, &y, &lv.xf[j].norm, 1e-6f, &xn_next));
                else TRY(resnet(lv.res[j], x, nullptr, &y));
                x = y;
                if (has_attn) {
                    TRY(transformer(lv.xf[j], x, &y, &xn_next));
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
                // attention-free block: diffusers adds the adapter feature AFTER the block returned, but IN PLACE
                // (`sample += down_intrablock_additional_residuals.pop(0)`, unet_2d_condition.py forward) on the very
                // tensor DownBlock2D also returned as res_samples[-1] - so the LAST skip pushed above carries the
                // feature (as in the original TencentARC loop, which adds before hs.append(h)); the earlier skips of
                // the block do not.  x shares its buffer with skips->back(): add in place.
                TRY(add_external(x, intrablock[ib++]));
            }
        }
        *out = x;
        return 0;
    }
    int mid(Act x, Act* out) {
        Act y, xn_next;
        xn_next.p = nullptr;
        TRY(resnet(m.mid_r0, x, nullptr, &y, &m.mid_xf.norm, 1e-6f, &xn_next));
        x = y;
        TRY(transformer(m.mid_xf, x, &y, &xn_next));
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
                Act y, xn_next;
                xn_next.p = nullptr;
                if (!lv.xf.empty()) TRY(resnet(lv.res[j], x, &sk, &y, &lv.xf[j].norm, 1e-6f, &xn_next));
                else TRY(resnet(lv.res[j], x, &sk, &y));
                x = y;
                if (!lv.xf.empty()) {
                    TRY(transformer(lv.xf[j], x, &y, &xn_next));
                    x = y;
                }
            }
            if (lv.has_up) {
                Act y;
                TRY(upsample_conv(x, lv, &y));
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
            const int act = last ? ACT_NONE : ACT_SILU;
            constexpr int KQ = 128 / (int)sizeof(T);
            if (e.C % KQ == 0 && m.ce[k].cout % 4 == 0) TRY(conv3(e, nullptr, m.ce[k], m.ce_stride[k], 0, nullptr, 0, 1, nullptr, act, &y));
            else TRY(direct(e, m.ce[k], m.ce_stride[k], act, nullptr, &y));  // the first layer (3 input channels)
            e = y;
        }
        MRISR_REQUIRE(e.numel() * sizeof(T) <= m.cond_emb_bytes, "condition embedding larger than planned");
        if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(m.cond_emb, e.p, e.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (!m.keep) m.arena.release(mk);
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
            if (!m.keep) m.arena.release(mk);
        }
        Act o = new_act(x.B, x.H, x.W, x.C);
        if (!o.p) return 7;
        TRY(linear(x.p, (int)x.rows(), x.C, m.cn_mid, ACT_NONE, nullptr, 0, nullptr, o.p, o.C));
        return export_act(o, *mid_out, scale);
    }
};

}  // namespace mrisr
