// LoRA fine-tuning step of the denoiser (reference: the training cell of the ResDif notebook,
// nb:ResDif c11:14-41 - loss = mse(eps_hat, eps), backward for the adapters only, clip 1.0, AdamW).
//
// Forward: the same launch sequence as inference (runner.h), run with Model::keep so nothing is recycled, with P of
// every attention materialised and every block recording what its backward needs.  Backward: the recorded blocks are
// replayed in reverse.  Base weights are frozen, so the backward is dX everywhere (dgrad GEMMs on transposed / tap-flipped
// weight copies packed once by train_prepare) plus the rank-r wgrads of the adapters, accumulated with float atomics
// into the caller's flat f32 gradient vector (which the host all-reduces over RCCL before the optimiser step).
#include <algorithm>
#include <functional>
#include <set>

#include "model.h"
#include "runner.h"

namespace mrisr {

// ================================================================================================
// load-time packing for the backward
// ================================================================================================
// wd[ci][ky][kx][co] = w[co][ci][2-ky][2-kx]: the dgrad of a 3x3 conv is a 3x3 conv of dY with this bank
template <typename T>
__global__ void pack_conv_dgrad_kernel(const float* __restrict__ w, T* __restrict__ wd, int Cout, int Cin) {
    const long long total = (long long)Cout * Cin * 9;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const int t = (int)((i / Cout) % 9);
        const int ci = (int)(i / (9ll * Cout));
        const int ky = t / 3, kx = t - ky * 3;
        wd[i] = from_f32<T>(w[(((size_t)co * Cin + ci) * 3 + (2 - ky)) * 3 + (2 - kx)]);
    }
}

// adapter of fused-module slot `slot` (theta: A [r][k], B [n][r]) -> the four device views the kernels read:
//   loraA [R][k] T (forward down-projection), loraAT [k][R] f32 (dgrad epilogue),
//   loraB [ntot][r] f32 = s*B (forward epilogue), loraBT [R][ntot] T = s*B^T (backward down-projection)
template <typename T>
__global__ void lora_refresh_kernel(const float* __restrict__ A, const float* __restrict__ B, T* loraA, float* loraAT, float* loraB,
                                    T* loraBT, int r, int R, int k, int n, int ntot, int slot, int row0, float s) {
    const long long na = (long long)r * k, nb = (long long)n * r;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < na + nb; i += (long long)gridDim.x * 256) {
        if (i < na) {
            const int q = (int)(i / k), c = (int)(i - (long long)q * k);
            const float v = A[i];
            loraA[(size_t)(slot * r + q) * k + c] = from_f32<T>(v);
            loraAT[(size_t)c * R + slot * r + q] = v;
        } else {
            const long long j = i - na;
            const int c = (int)(j / r), q = (int)(j - (long long)c * r);
            const float v = s * B[j];
            loraB[(size_t)(row0 + c) * r + q] = v;
            loraBT[(size_t)(slot * r + q) * ntot + row0 + c] = from_f32<T>(v);
        }
    }
}

template <typename F>
static void for_each_xf(Model& m, F f) {
    for (auto& lv : m.down) for (auto& x : lv.xf) f(x);
    f(m.mid_xf);
    for (auto& lv : m.up) for (auto& x : lv.xf) f(x);
}
template <typename F>
static void for_each_res(Model& m, F f) {
    for (auto& lv : m.down) for (auto& r : lv.res) f(r);
    f(m.mid_r0);
    f(m.mid_r1);
    for (auto& lv : m.up) for (auto& r : lv.res) f(r);
}

std::vector<LinW*> Model::lora_linears() {
    std::vector<LinW*> v;
    for_each_xf(*this, [&](XfW& x) {
        for (LinW* l : {&x.proj_in, &x.qkv, &x.out1, &x.q2, &x.kv2, &x.out2, &x.ff2, &x.proj_out})
            if (l->R) v.push_back(l);
    });
    return v;
}

template <typename T>
static int train_prepare_t(Model& m, hipStream_t st) {
    int err = 0;
    auto conv_dgrad = [&](ConvW& c) {
        if (!c.w || c.ks != 3 || c.wd) return;
        const RawParam* w = m.find(c.name + ".weight");
        if (!w) { err = 3; set_error("missing parameter: " + c.name + ".weight"); return; }
        c.wd = m.new_packed((size_t)w->numel() * sizeof(T), false);
        if (!c.wd) { err = 4; return; }
        long long blocks = (w->numel() + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(pack_conv_dgrad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float*>(w->data->p),
                           static_cast<T*>(c.wd), c.cout, c.cin);
    };
    auto lin_t = [&](LinW& l) {
        if (!l.w || l.wT) return;
        l.wT = m.new_packed((size_t)l.n * l.k * sizeof(T), false);
        if (!l.wT) { err = 4; return; }
        if (launch_transpose<T>(l.w, l.wT, l.n, l.k, l.k, l.n, 0, 0, 1, l.n, st)) err = 5;
    };
    for_each_res(m, [&](ResW& r) {
        conv_dgrad(r.c1);
        conv_dgrad(r.c2);
        if (r.has_sc) lin_t(r.sc);
    });
    for (auto& lv : m.down) if (lv.has_down) conv_dgrad(lv.down);
    for (auto& lv : m.up) if (lv.has_up) conv_dgrad(lv.up);
    conv_dgrad(m.conv_out);
    for_each_xf(m, [&](XfW& x) {
        for (LinW* l : {&x.proj_in, &x.qkv, &x.out1, &x.q2, &x.out2, &x.ff1, &x.ff2, &x.proj_out}) lin_t(*l);
        if (x.ff1.R) { err = 6; set_error("LoRA on the GEGLU projection is not supported by the fine-tuning step"); }
    });
    if (err) return err;
    // flat trainable vector: [A_0 | B_0 | A_1 | B_1 ...] in lora_linears() x fused-module order
    m.trainables.clear();
    long long off = 0;
    for (LinW* l : m.lora_linears()) {
        const int nmod = (int)l->mod_names.size();
        l->offA.assign(nmod, -1);
        l->offB.assign(nmod, -1);
        for (int j = 0; j < nmod; ++j) {
            if (!l->mod_lora[j]) continue;
            l->offA[j] = off;
            m.trainables.push_back({l->mod_names[j] + ".lora_A.default.weight", off, (long long)l->r * l->k, l->r, l->k});
            off += (long long)l->r * l->k;
            l->offB[j] = off;
            m.trainables.push_back({l->mod_names[j] + ".lora_B.default.weight", off, (long long)l->secN * l->r, l->secN, l->r});
            off += (long long)l->secN * l->r;
        }
        if (!l->loraBT) {
            l->loraBT = m.new_packed((size_t)l->R * l->n * sizeof(T), true);
            l->loraAT = static_cast<float*>(m.new_packed((size_t)l->k * l->R * sizeof(float), true));
            if (!l->loraBT || !l->loraAT) return 4;
        }
    }
    m.n_trainable = off;
    MRISR_CHECK_HIP(hipStreamSynchronize(st));
    m.train_ready = true;
    return 0;
}

int Model::train_prepare(hipStream_t st) {
    MRISR_REQUIRE(finalized, "call mrisr_model_finalize first");
    MRISR_REQUIRE(!is_controlnet, "the fine-tuning step is implemented for the UNet handle");
    // lora_rank == 0: a FROZEN UNet - the step then only differentiates with respect to its inputs (ControlNet residuals,
    // T2I-Adapter features): no trainable vector, dX only
    MRISR_REQUIRE(cfg.lora_rank == 0 || cfg.lora_fused, "fine-tuning needs explicit (un-merged) LoRA adapters");
    if (train_ready) return 0;
    return cfg.compute_dtype == MRISR_F32 ? train_prepare_t<float>(*this, st) : train_prepare_t<bf16>(*this, st);
}

int Model::train_bind(float* theta_dev, float* grad_dev, hipStream_t st) {
    TRY(train_prepare(st));
    MRISR_REQUIRE(n_trainable == 0 || (theta_dev && grad_dev), "theta / grad device buffers");
    theta = theta_dev;
    grad = grad_dev;
    return 0;
}

int Model::lora_refresh(hipStream_t st) {
    MRISR_REQUIRE(train_ready && (theta || n_trainable == 0), "bind the trainable vector first");
    for (LinW* l : lora_linears()) {
        int row0 = 0;
        for (size_t j = 0; j < l->mod_names.size(); ++j, row0 += l->secN) {
            if (!l->mod_lora[j]) continue;
            const float* A = theta + l->offA[j];
            const float* B = theta + l->offB[j];
            const long long total = (long long)l->r * l->k + (long long)l->secN * l->r;
            const unsigned blocks = (unsigned)((total + 255) / 256);
            if (cfg.compute_dtype == MRISR_F32)
                hipLaunchKernelGGL(lora_refresh_kernel<float>, dim3(blocks), dim3(256), 0, st, A, B, static_cast<float*>(l->loraA), l->loraAT,
                                   l->loraB_rw, static_cast<float*>(l->loraBT), l->r, l->R, l->k, l->secN, l->n, (int)j, row0, lora_scale);
            else
                hipLaunchKernelGGL(lora_refresh_kernel<bf16>, dim3(blocks), dim3(256), 0, st, A, B, static_cast<bf16*>(l->loraA), l->loraAT,
                                   l->loraB_rw, static_cast<bf16*>(l->loraBT), l->r, l->R, l->k, l->secN, l->n, (int)j, row0, lora_scale);
        }
        // the fp8 copy of the adapters' A rows (inference through the fp8 projections after training steps)
        if (l->loraA8 && cfg.compute_dtype != MRISR_F32) TRY(launch_quant_rows_fp8(l->loraA, l->R, l->k, l->loraA8, l->loraA_scale, st));
    }
    MRISR_CHECK_HIP(hipGetLastError());
    ctx_valid = false;  // cached cross-attention K / V depend on attn2.to_k / to_v adapters
    return 0;
}

// ================================================================================================
// the step
// ================================================================================================
template <typename T>
struct Trainer : Runner<T> {
    typedef Runner<T> R;
    using R::m;
    using R::st;
    using R::dry;
    using R::alloc;
    using R::new_act;
    static constexpr int BK = 128 / (int)sizeof(T);

    struct Slot { void* g = nullptr; bool written = false; };
    std::map<const void*, Slot> slots;   // activation -> its gradient buffer
    std::set<const void*> live;          // activations downstream of a trainable parameter
    std::vector<std::function<int()>> tape;
    std::map<const XfW*, float*> z_kv;   // adapter down-projections of the cached context K / V

    Trainer(Model& mm, hipStream_t s, bool d) : R(mm, s, d) {}

    Slot& slot(const Act& a) {
        auto it = slots.find(a.p);
        if (it == slots.end()) {
            Slot s;
            s.g = alloc(a.numel() * sizeof(T));
            it = slots.insert({a.p, s}).first;
        }
        return it->second;
    }
    bool is_live(const Act& a) const { return live.count(a.p) != 0; }
    int zero(void* p, size_t bytes) {
        if (!p) return 7;
        if (dry) return 0;
        MRISR_CHECK_HIP(hipMemsetAsync(p, 0, bytes, st));
        return 0;
    }
    static bool has_lora(const XfW& x) {
        return x.proj_in.R || x.qkv.R || x.out1.R || x.q2.R || x.kv2.R || x.out2.R || x.ff2.R || x.proj_out.R;
    }

    // ---------------------------------------------------------------------------------------------
    // backward building blocks
    // ---------------------------------------------------------------------------------------------
    // ---------------------------------------------------------------------------------------------
    // full-parameter training (m.full_grad bound: ControlNet): weight / bias gradients of every conv and linear, accumulated into
    // the caller's flat f32 vector at the raw tensor's offset (PyTorch layouts).  One pixel-contraction GEMM per layer,
    //     dW[co][tap * cin + ci] = sum_m dY^T[co][m] . im2col^T[tap * cin + ci][m]
    // (a linear / 1 x 1 conv is the ks = 1 case), then one scatter-add; db by column sums.
    // ---------------------------------------------------------------------------------------------
    bool full() const { return m.full_grad != nullptr; }
    float* gptr(const std::string& key) const {
        auto it = m.full_off.find(key);
        return it == m.full_off.end() ? nullptr : m.full_grad + it->second;
    }
    // x: NHWC [xB][xH][xW][cin_src] (rows [M][cin_src] for a linear: xB = 1, xH = M, xW = 1); dY: rows [M][.] of pitch ldy, columns
    // col0 .. col0 + cout_src; the raw tensor is [cout][cin][ks][ks] with cout <= cout_src, cin <= cin_src (zero-padded layers).
    int wgrad(const void* x, int xB, int xH, int xW, int cin_src, const void* dY, int ldy, int col0, int Ho, int Wo, int cout_src, int ks,
              int stride, float* gW, float* gB, int cout, int cin, int geglu_half = 0) {
        const int M = xB * Ho * Wo, Mpad = (M + 63) / 64 * 64, taps = ks * ks;
        const int ncol = (taps * cin_src + 3) & ~3;  // the GEMM's N in multiples of 4 (3-channel images: 27 -> 28, one zero column)
        const size_t mk = m.arena.mark();
        T* dyT = static_cast<T*>(alloc((size_t)cout_src * Mpad * sizeof(T)));
        T* xT = static_cast<T*>(alloc((size_t)ncol * Mpad * sizeof(T)));
        float* tmp = static_cast<float*>(alloc((size_t)cout_src * ncol * sizeof(float)));
        if (!dyT || !xT || !tmp) return 7;
        if (!dry) {
            if (Mpad != M || ncol != taps * cin_src) {
                MRISR_CHECK_HIP(hipMemsetAsync(dyT, 0, (size_t)cout_src * Mpad * sizeof(T), st));
                MRISR_CHECK_HIP(hipMemsetAsync(xT, 0, (size_t)ncol * Mpad * sizeof(T), st));
            }
            TRY(launch_transpose<T>(static_cast<const T*>(dY) + col0, dyT, M, cout_src, ldy, Mpad, 0, 0, 1, M, st));
            if (gB) TRY(launch_colsum_gen<T>(dY, ldy, col0, gB, M, cout, geglu_half, st));
            TRY(launch_im2col_all_T<T>(x, xT, xB, xH, xW, cin_src, Ho, Wo, stride, ks / 2, ks, Mpad, st));
        }
        GemmArgs g;
        g.a0 = dyT; g.c0 = Mpad; g.lda0 = Mpad;
        g.w = xT; g.M = cout_src; g.N = ncol; g.K = Mpad;
        g.out_mode = OUT_F32; g.out = tmp; g.ldo = ncol;
        TRY(R::run_gemm(g));
        if (!dry && gW) TRY(launch_wgrad_accum_gen(tmp, ncol, cin_src, gW, cout, cin, taps, geglu_half, st));
        m.arena.release(mk);
        return 0;
    }
    int wgrad_conv(const Act& x, const Act& dy, const ConvW& cw, int stride) {
        if (!full()) return 0;
        const RawParam* w = m.find(cw.name + ".weight");
        MRISR_REQUIRE(w && dy.C == cw.cout && x.C == cw.cin, "conv weight gradient: operands");
        return wgrad(x.p, x.B, x.H, x.W, x.C, dy.p, dy.C, 0, dy.H, dy.W, dy.C, cw.ks, stride, gptr(cw.name + ".weight"), gptr(cw.name + ".bias"),
                     (int)w->shape[0], (int)w->shape[1]);
    }
    // a (possibly fused) linear: one gradient GEMM per fused module on its column section of dY
    int wgrad_linear(const LinW& lw, const void* x, int ldx, const void* dY, int ldy, int M) {
        if (!full()) return 0;
        MRISR_REQUIRE(ldx == lw.k && !lw.mod_names.empty(), "linear weight gradient: contiguous input rows of a named module");
        const int nmod = (int)lw.mod_names.size(), secN = lw.n / nmod;
        for (int j = 0; j < nmod; ++j) {
            const std::string& name = lw.mod_names[j];
            const bool geglu = name.size() > 13 && name.compare(name.size() - 13, 13, "ff.net.0.proj") == 0;
            TRY(wgrad(x, 1, M, 1, lw.k, dY, ldy, j * secN, M, 1, secN, 1, 1, gptr(name + ".weight"), gptr(name + ".bias"), secN, lw.k,
                      geglu ? secN / 2 : 0));
        }
        return 0;
    }

    // affine parameters of a GroupNorm(+SiLU) / LayerNorm (dy: gradient with respect to the norm's OUTPUT)
    int gn_affine(const Act& x, const void* dy, const NormW& nw, bool silu, float eps, const float* fwd_partial, int nsplit) {
        if (!full() || dry) return 0;
        float* gg = gptr(nw.name + ".weight");
        float* gb = gptr(nw.name + ".bias");
        MRISR_REQUIRE(gg && gb && nw.c == x.C, "GroupNorm affine gradient: parameters");
        return launch_gn_affine_grad<T>(x.p, dy, nw.g, nw.b, fwd_partial, nsplit, m.cfg.norm_num_groups, x.B, x.H * x.W, x.C, eps, silu ? 1 : 0, gg, gb, st);
    }
    int ln_affine(const void* x, const void* dy, const NormW& nw, int M, int C) {
        if (!full() || dry) return 0;
        float* gg = gptr(nw.name + ".weight");
        float* gb = gptr(nw.name + ".bias");
        MRISR_REQUIRE(gg && gb, "LayerNorm affine gradient: parameters");
        return launch_ln_affine_grad<T>(x, dy, M, C, 1e-5f, gg, gb, st);
    }
    // the time-embedding path, after every ResnetBlock has added its share to m.d_tproj:  tproj_out = Wp silu(emb) + bp,
    // emb = W2 silu(y1) + b2,  y1 = W1 s + b1  (Runner::time_embed); rows = 1 (scalar timestep) or B
    int time_embed_bwd(int B) {
        if (!full()) return 0;
        const int rows = m.t_scalar ? 1 : B, c0 = m.cfg.block_out_channels[0], temb = 4 * c0;
        const size_t mk = m.arena.mark();
        float* d_emb = static_cast<float*>(alloc((size_t)rows * temb * sizeof(float)));
        float* d_y1 = static_cast<float*>(alloc((size_t)rows * temb * sizeof(float)));
        if (!d_emb || !d_y1) return 7;
        if (!dry) {
            int off = 0;
            for (auto& mod : m.temb_mods) {
                const RawParam* w = m.find(mod + ".weight");
                MRISR_REQUIRE(w, "time_emb_proj weight");
                const int n = (int)w->shape[0];
                TRY(launch_small_wgrad(m.d_tproj + off, m.tproj_total, m.te_emb, temb, rows, n, temb, 1, gptr(mod + ".weight"), gptr(mod + ".bias"), st));
                off += n;
            }
            TRY(launch_small_dgrad<T>(m.d_tproj, m.tproj_total, m.tproj.w, rows, m.tproj_total, temb, m.te_emb, temb, d_emb, temb, st));
            TRY(launch_small_wgrad(d_emb, temb, m.te_y1, temb, rows, temb, temb, 1, gptr("time_embedding.linear_2.weight"), gptr("time_embedding.linear_2.bias"), st));
            TRY(launch_small_dgrad<T>(d_emb, temb, m.te2.w, rows, temb, temb, m.te_y1, temb, d_y1, temb, st));
            TRY(launch_small_wgrad(d_y1, temb, m.te_s, c0, rows, temb, c0, 0, gptr("time_embedding.linear_1.weight"), gptr("time_embedding.linear_1.bias"), st));
        }
        m.arena.release(mk);
        return 0;
    }

    // dX (+)= conv3x3(dY, wd).  mode 1: the forward conv had stride 2 -> dY is zero-stuffed to twice its size
    int conv_dgrad(const Act& dy, const ConvW& cw, int mode, void* out, bool acc) {
        MRISR_REQUIRE(cw.wd && dy.C == cw.cout, "conv dgrad weights");
        const int Ho = dy.H << mode, Wo = dy.W << mode;
        if (cw.cout % BK != 0 || cw.cin % 4 != 0) {  // conv_out (4 channels): far below one K tile
            MRISR_REQUIRE(mode == 0, "strided dgrad of a tiny conv");
            DirectConvArgs a;
            a.x = dy.p; a.w = cw.wd; a.y = out; a.B = dy.B; a.Hin = dy.H; a.Win = dy.W; a.Cin = cw.cout;
            a.Hout = Ho; a.Wout = Wo; a.Cout = cw.cin; a.ks = 3; a.stride = 1; a.pad = 1; a.act = ACT_NONE;
            a.add = acc ? out : nullptr;
            if (dry) return 0;
            return launch_direct_conv<T>(a, st);
        }
        GemmArgs g;
        g.a0 = dy.p; g.c0 = cw.cout; g.lda0 = cw.cout;
        g.conv = 1; g.B = dy.B; g.Hin = dy.H; g.Win = dy.W; g.Hout = Ho; g.Wout = Wo; g.stride = 1; g.ups = mode; g.zstuff = mode;
        g.w = cw.wd; g.M = dy.B * Ho * Wo; g.N = cw.cin; g.K = 9 * cw.cout;
        if (acc) { g.resid = out; g.ldr = cw.cin; }
        g.out = out; g.ldo = cw.cin;
        return R::run_gemm(g);
    }

    int gn_bwd(const Act& x0, const Act* x1, const NormW& nw, bool silu, float eps, const float* fwd_partial, int nsplit,
               const void* dy, void* dx0, bool acc0, void* dx1, bool acc1) {
        GroupNormBwdArgs a;
        a.x0 = x0.p; a.c0 = x0.C;
        a.x1 = x1 ? x1->p : nullptr; a.c1 = x1 ? x1->C : 0;
        a.B = x0.B; a.HW = x0.H * x0.W; a.groups = m.cfg.norm_num_groups; a.eps = eps;
        a.gamma = nw.g; a.beta = nw.b; a.silu = silu ? 1 : 0;
        a.dy = dy; a.dx0 = dx0; a.dx1 = dx1; a.acc0 = acc0 ? 1 : 0; a.acc1 = acc1 ? 1 : 0;
        a.fwd_partial = fwd_partial; a.nsplit = nsplit;
        a.bwd_partial = static_cast<float*>(alloc((size_t)a.B * nsplit * a.groups * 2 * sizeof(float)));
        if (!a.bwd_partial) return 7;
        if (dry) return 0;
        return launch_groupnorm_bwd<T>(a, st);
    }

    // backward of y = x W^T + s B (A x):  dX (+)= dY W + (dY s B) A;  dA, dB accumulated into m.grad
    int linear_bwd(const LinW& lw, const void* x, int ldx, const float* z, const void* dY, int ldy, int M, void* dX, bool acc,
                   bool need_dx) {
        float* dz = nullptr;
        GemmArgs g;
        bool in_kernel = false, dx_done = false;
        TRY(wgrad_linear(lw, x, ldx, dY, ldy, M));  // (full-parameter training only)
        if (need_dx) {
            MRISR_REQUIRE(lw.wT, "linear dgrad weights");
            g.a0 = dY; g.c0 = lw.n; g.lda0 = ldy;
            g.w = lw.wT; g.M = M; g.N = lw.k; g.K = lw.n;
            if (acc) { g.resid = dX; g.ldr = lw.k; }
            g.out = dX; g.ldo = lw.k;
            TRY(gemm_choose(g, sizeof(T) == 2));
            // as in the forward: dz = dY (s B) can ride along the dgrad GEMM (the <= 16 rows of s B^T as extra weight rows)
            in_kernel = lw.R && sizeof(T) == 2 && lw.R <= 16 && g.splitk == 1 && g.tile >= 14 && lora_in_kernel();
        }
        if (lw.R) {
            MRISR_REQUIRE(z && lw.loraBT && lw.loraAT && m.grad, "adapter backward state");
            dz = static_cast<float*>(alloc((size_t)M * lw.R * sizeof(float)));
            const size_t sb = std::max(lora_wgrad_scratch_bytes(M, lw.n, lw.R, sizeof(T)), lora_wgrad_scratch_bytes(M, lw.k, lw.R, sizeof(T)));
            float* scratch = static_cast<float*>(alloc(sb));
            if (!dz || !scratch) return 7;
            if (in_kernel) {  // the dgrad first: it produces dz for the A gradient
                g.lora_a = lw.loraBT; g.lora_R = lw.R; g.lora_zout = dz;
                g.lora_zld = lw.R; g.lora_b = lw.loraAT; g.lora_r = lw.R; g.lora_secN = INT_MAX;
                if (!dry) TRY(launch_gemm<T>(g, st));
                dx_done = true;
            }
            if (!dry) {
                if (!in_kernel) TRY(launch_lora_down<T>(dY, ldy, lw.loraBT, dz, M, lw.n, lw.R, st));
                const int nmod = (int)lw.mod_names.size();
                float *oA[3] = {nullptr, nullptr, nullptr}, *oB[3] = {nullptr, nullptr, nullptr};
                for (int j = 0; j < nmod; ++j) {
                    if (!lw.mod_lora[j]) continue;
                    oA[j] = m.grad + lw.offA[j];
                    oB[j] = m.grad + lw.offB[j];
                }
                TRY(launch_lora_wgrad<T>(dY, ldy, z, lw.R, M, lw.n, 0, lw.r, nmod, lw.secN, oB, m.lora_scale, scratch, st));
                TRY(launch_lora_wgrad<T>(x, ldx, dz, lw.R, M, lw.k, 1, lw.r, nmod, lw.secN, oA, 1.0f, scratch, st));
            }
        }
        if (!need_dx || dx_done) return 0;
        if (dz) { g.lora_z = dz; g.lora_zld = lw.R; g.lora_b = lw.loraAT; g.lora_r = lw.R; g.lora_secN = INT_MAX; }
        if (g.splitk > 1) {
            g.partial = static_cast<float*>(alloc((size_t)g.splitk * g.batch * g.M * g.N * sizeof(float)));
            if (!g.partial) return 7;
        }
        if (dry) return 0;
        return launch_gemm<T>(g, st);
    }

    struct AttnRec {
        const void* q = nullptr;   // [B*H][npad][dpad]
        const void* k = nullptr;   // [B*H][nkpad][dpad]
        const void* vt = nullptr;  // [B*H][dpad][nkpad]
        const void* P = nullptr;   // [B*H][N][nkpad] softmax probabilities (T): materialised path only
        const float* lse = nullptr;  // [B*H][npad] log-sum-exp: flash path only
        const void* out = nullptr; // attention output rows [M][C] (flash backward: D = rowsum(dO o O))
        int nk = 0, nkpad = 0;
    };
    bool flash() const { return sizeof(T) == 2 && m.cfg.flash_attention; }

    // materialised attention that keeps P for the backward
    int attention_t(const HeadBuf& hb, AttnRec* a, void* out_rows) {
        const int BH = hb.B * hb.H;
        const float scale = 1.0f / sqrtf((float)hb.hd);
        a->out = out_rows;
        if (flash()) {
            float* lse = static_cast<float*>(alloc((size_t)BH * hb.npad * sizeof(float)));
            if (!lse) return 7;
            a->lse = lse;
            AttnArgs fa;
            fa.q = a->q; fa.k = a->k; fa.vt = a->vt; fa.out = out_rows;
            fa.B = hb.B; fa.H = hb.H; fa.nq = hb.N; fa.nk = a->nk; fa.nkpad = a->nkpad; fa.hd = hb.hd; fa.dpad = hb.dpad;
            fa.scale = scale; fa.lse = lse;
            if (m.cfg.fp8_attention && m.cfg.fp8_train) {  // fp8 forward, bf16 backward from the same log-sum-exp (straight through)
                fa.k8 = alloc((size_t)BH * a->nkpad * hb.dpad);
                fa.vt8 = alloc((size_t)BH * a->nkpad * hb.dpad);
                fa.f8_scales = static_cast<float*>(alloc((size_t)BH * 4 * sizeof(float)));
                if (!fa.k8 || !fa.vt8 || !fa.f8_scales) return 7;
                if (dry) return 0;
                return launch_attention_fp8(fa, st);
            }
            if (dry) return 0;
            return launch_attention_bf16(fa, st);
        }
        const size_t cnt = (size_t)BH * hb.N * a->nkpad;
        float* S;
        void* P;
        size_t mk = 0;
        if (sizeof(T) == 4) {
            S = static_cast<float*>(alloc(cnt * sizeof(float)));
            P = S;
        } else {
            P = alloc(cnt * sizeof(T));
            mk = m.arena.mark();
            S = static_cast<float*>(alloc(cnt * sizeof(float)));
        }
        if (!S || !P) return 7;
        GemmArgs g;
        g.a0 = a->q; g.c0 = hb.dpad; g.lda0 = hb.dpad; g.a_bs = (long long)hb.npad * hb.dpad;
        g.w = a->k; g.w_bs = (long long)a->nkpad * hb.dpad;
        g.M = hb.N; g.N = a->nkpad; g.K = hb.dpad; g.batch = BH; g.alpha = scale;
        g.out_mode = OUT_F32; g.out = S; g.ldo = a->nkpad; g.o_bs = (long long)hb.N * a->nkpad;
        TRY(R::run_gemm(g));
        if (!dry) TRY(launch_softmax_rows<T>(S, a->nkpad, P, a->nkpad, (long long)BH * hb.N, a->nk, st));
        GemmArgs o;
        o.a0 = P; o.c0 = a->nkpad; o.lda0 = a->nkpad; o.a_bs = (long long)hb.N * a->nkpad;
        o.w = a->vt; o.w_bs = (long long)hb.dpad * a->nkpad;
        o.M = hb.N; o.N = hb.hd; o.K = a->nkpad; o.batch = BH;
        o.heads = hb.H; o.o_bs = (long long)hb.N * hb.H * hb.hd; o.o_hs = hb.hd;
        o.out = out_rows; o.ldo = hb.H * hb.hd;
        TRY(R::run_gemm(o));
        if (sizeof(T) == 2) m.arena.release(mk);
        a->P = P;
        return 0;
    }

    // dQ / dK / dV of one attention, written as token rows (head h -> columns h*hd..) at the given pointers / pitches.
    //   dV = P^T dO,  dP = dO V^T,  dS = scale * P o (dP - rowsum(dP o P)),  dQ = dS K,  dK = dS^T Q
    int attention_bwd(const HeadBuf& hb, const AttnRec& a, const void* dO_rows, T* dq, int ldq, T* dk, T* dv, int ldkv) {
        const int BH = hb.B * hb.H, N = hb.N, npad = hb.npad, dpad = hb.dpad, nk = a.nk, nkpad = a.nkpad;
        const float scale = 1.0f / sqrtf((float)hb.hd);
        const size_t qsz = (size_t)BH * npad * dpad, ksz = (size_t)BH * nkpad * dpad, psz = (size_t)BH * N * nkpad;
        if (flash()) {
            // P recomputed from the log-sum-exp inside the kernels; only [tokens x head dim] operands are staged
            void* doh = alloc(qsz * sizeof(T));
            float* dsum = static_cast<float*>(alloc((size_t)BH * npad * sizeof(float)));
            void* V = alloc(ksz * sizeof(T));
            void* Kt = alloc(ksz * sizeof(T));
            void* Qt = dk ? alloc(qsz * sizeof(T)) : nullptr;
            void* doht = dk ? alloc(qsz * sizeof(T)) : nullptr;
            if (!doh || !dsum || !V || !Kt || (dk && (!Qt || !doht))) return 7;
            TRY(zero(doh, qsz * sizeof(T)));
            TRY(zero(dsum, (size_t)BH * npad * sizeof(float)));
            if (dry) return 0;
            TRY(launch_attention_bwd_prep(dO_rows, a.out, doh, dsum, hb.B, N, hb.H, hb.hd, npad, dpad, st));
            TRY(launch_transpose<T>(a.vt, V, dpad, nkpad, nkpad, dpad, (long long)dpad * nkpad, (long long)nkpad * dpad, BH, dpad, st));
            TRY(launch_transpose<T>(a.k, Kt, nkpad, dpad, dpad, nkpad, (long long)nkpad * dpad, (long long)dpad * nkpad, BH, nkpad, st));
            if (dk) {
                TRY(launch_transpose<T>(a.q, Qt, npad, dpad, dpad, npad, (long long)npad * dpad, (long long)dpad * npad, BH, npad, st));
                TRY(launch_transpose<T>(doh, doht, npad, dpad, dpad, npad, (long long)npad * dpad, (long long)dpad * npad, BH, npad, st));
            }
            AttnBwdArgs g;
            g.q = a.q; g.k = a.k; g.v = V; g.doh = doh; g.qt = Qt; g.kt = Kt; g.doht = doht; g.lse = a.lse; g.dsum = dsum;
            g.dq = dq; g.ldq = ldq; g.dk = dk; g.dv = dv; g.ldkv = ldkv;
            g.B = hb.B; g.H = hb.H; g.nq = N; g.nk = nk; g.npad = npad; g.nkpad = nkpad; g.hd = hb.hd; g.dpad = dpad; g.scale = scale;
            return launch_attention_bwd_bf16(g, st);
        }
        T* dOh = static_cast<T*>(alloc(qsz * sizeof(T)));
        T* V = static_cast<T*>(alloc(ksz * sizeof(T)));
        float* dP = static_cast<float*>(alloc(psz * sizeof(float)));
        T* dS = static_cast<T*>(alloc(psz * sizeof(T)));
        T* Kt = static_cast<T*>(alloc(ksz * sizeof(T)));
        if (!dOh || !V || !dP || !dS || !Kt) return 7;
        TRY(zero(dOh, qsz * sizeof(T)));
        if (!dry) {
            TRY(launch_rows_to_heads<T>(dO_rows, hb.H * hb.hd, 0, dOh, hb.B, N, hb.H, hb.hd, npad, dpad, st));
            TRY(launch_transpose<T>(a.vt, V, dpad, nkpad, nkpad, dpad, (long long)dpad * nkpad, (long long)nkpad * dpad, BH, dpad, st));
            TRY(launch_transpose<T>(a.k, Kt, nkpad, dpad, dpad, nkpad, (long long)nkpad * dpad, (long long)dpad * nkpad, BH, nkpad, st));
        }
        {   // dP = dOh V^T
            GemmArgs g;
            g.a0 = dOh; g.c0 = dpad; g.lda0 = dpad; g.a_bs = (long long)npad * dpad;
            g.w = V; g.w_bs = (long long)nkpad * dpad;
            g.M = N; g.N = nkpad; g.K = dpad; g.batch = BH;
            g.out_mode = OUT_F32; g.out = dP; g.ldo = nkpad; g.o_bs = (long long)N * nkpad;
            TRY(R::run_gemm(g));
        }
        if (!dry) TRY(launch_softmax_bwd<T>(a.P, dP, dS, nkpad, (long long)BH * N, nk, scale, st));
        {   // dQ = dS K  (W operand = K^T)
            GemmArgs g;
            g.a0 = dS; g.c0 = nkpad; g.lda0 = nkpad; g.a_bs = (long long)N * nkpad;
            g.w = Kt; g.w_bs = (long long)dpad * nkpad;
            g.M = N; g.N = hb.hd; g.K = nkpad; g.batch = BH;
            g.heads = hb.H; g.o_bs = (long long)N * ldq; g.o_hs = hb.hd; g.out = dq; g.ldo = ldq;
            TRY(R::run_gemm(g));
        }
        if (!dk) return 0;
        T* At = static_cast<T*>(alloc((size_t)BH * nkpad * npad * sizeof(T)));  // dS^T, then P^T
        T* Wt = static_cast<T*>(alloc(qsz * sizeof(T)));                        // Q^T, then dOh^T
        if (!At || !Wt) return 7;
        TRY(zero(At, (size_t)BH * nkpad * npad * sizeof(T)));  // token pads N..npad stay zero through both uses
        auto kv_grad = [&](const void* a_src, const void* w_src, T* out) -> int {
            if (!dry) {
                TRY(launch_transpose<T>(a_src, At, N, nkpad, nkpad, npad, (long long)N * nkpad, (long long)nkpad * npad, BH, N, st));
                TRY(launch_transpose<T>(w_src, Wt, npad, dpad, dpad, npad, (long long)npad * dpad, (long long)dpad * npad, BH, npad, st));
            }
            GemmArgs g;
            g.a0 = At; g.c0 = npad; g.lda0 = npad; g.a_bs = (long long)nkpad * npad;
            g.w = Wt; g.w_bs = (long long)dpad * npad;
            g.M = nk; g.N = hb.hd; g.K = npad; g.batch = BH;
            g.heads = hb.H; g.o_bs = (long long)nk * ldkv; g.o_hs = hb.hd; g.out = out; g.ldo = ldkv;
            return R::run_gemm(g);
        };
        TRY(kv_grad(dS, a.q, dk));
        TRY(kv_grad(a.P, dOh, dv));
        return 0;
    }

    // ---------------------------------------------------------------------------------------------
    // ResnetBlock2D
    // ---------------------------------------------------------------------------------------------
    int resnet_t(const ResW& r, const Act& x, const Act* x1, Act* out) {
        Act o = new_act(x.B, x.H, x.W, r.cout);
        if (!o.p) return 7;
        Act xn, h, hn;
        float *p1 = nullptr, *p2 = nullptr;
        int ns1 = 1, ns2 = 1;
        TRY(R::gn(x, x1, r.n1, true, m.cfg.norm_eps, &xn, &p1, &ns1));
        const int div = m.t_scalar ? INT_MAX : x.H * x.W;
        TRY(R::conv3(xn, nullptr, r.c1, 1, 0, m.tproj_out + r.temb_off, m.tproj_total, div, nullptr, ACT_NONE, &h));
        TRY(R::gn(h, nullptr, r.n2, true, m.cfg.norm_eps, &hn, &p2, &ns2));
        Act res = x;
        if (r.has_sc) {
            GemmArgs g;
            if (x1) { g.a1 = x1->p; g.c1 = x1->C; g.lda1 = x1->C; }
            g.a0 = x.p; g.c0 = x.C; g.lda0 = x.C;
            g.w = r.sc.w; g.M = (int)x.rows(); g.N = r.cout; g.K = r.cin; g.bias = r.sc.b; g.out = o.p; g.ldo = r.cout;
            TRY(R::run_gemm(g));
            res = o;
        } else {
            MRISR_REQUIRE(!x1, "concat input requires a shortcut conv");
        }
        {
            GemmArgs g;
            g.a0 = hn.p; g.c0 = hn.C; g.lda0 = hn.C;
            g.conv = 1; g.B = x.B; g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W;
            g.w = r.c2.w; g.M = (int)x.rows(); g.N = r.cout; g.K = 9 * r.cout; g.bias = r.c2.b;
            g.resid = res.p; g.ldr = r.cout; g.out = o.p; g.ldo = r.cout;
            TRY(R::run_gemm(g));
        }
        *out = o;
        const bool lv = is_live(x) || (x1 && is_live(*x1));
        if (!lv) return 0;  // nothing trainable upstream: the block has no backward
        live.insert(o.p);
        const bool has_x1 = x1 != nullptr;
        const Act x1v = x1 ? *x1 : Act();
        const ResW* rp = &r;
        tape.push_back([=]() -> int {
            const ResW& rr = *rp;
            auto it = slots.find(o.p);
            MRISR_REQUIRE(it != slots.end() && it->second.written, "resnet output has no gradient");
            void* dO = it->second.g;
            Slot& s0 = slot(x);
            Slot* s1 = has_x1 ? &slot(x1v) : nullptr;
            const size_t mk = m.arena.mark();
            Act dOa = o; dOa.p = dO;
            Act dhn = new_act(o.B, o.H, o.W, rr.cout), dh = new_act(o.B, o.H, o.W, rr.cout), dxn = new_act(o.B, o.H, o.W, rr.cin);
            if (!dhn.p || !dh.p || !dxn.p) return 7;
            TRY(wgrad_conv(hn, dOa, rr.c2, 1));
            TRY(conv_dgrad(dOa, rr.c2, 0, dhn.p, false));
            TRY(gn_affine(h, dhn.p, rr.n2, true, m.cfg.norm_eps, p2, ns2));
            TRY(gn_bwd(h, nullptr, rr.n2, true, m.cfg.norm_eps, p2, ns2, dhn.p, dh.p, false, nullptr, false));
            if (full() && m.d_tproj && !dry)  // h = conv1(.) + time_emb_proj(silu(emb))[:, :, None, None]: its gradient is the per-image sum of dh
                TRY(launch_rowvec_grad<T>(dh.p, m.d_tproj, m.tproj_total, rr.temb_off, o.B, o.H * o.W, rr.cout, m.t_scalar, st));
            TRY(wgrad_conv(xn, dh, rr.c1, 1));
            if (full() && rr.has_sc) {
                MRISR_REQUIRE(!has_x1, "full-parameter training: single-source shortcut");
                TRY(wgrad_linear(rr.sc, x.p, x.C, dO, rr.cout, (int)x.rows()));
            }
            TRY(conv_dgrad(dh, rr.c1, 0, dxn.p, false));
            if (!has_x1) TRY(gn_affine(x, dxn.p, rr.n1, true, m.cfg.norm_eps, p1, ns1));
            TRY(gn_bwd(x, has_x1 ? &x1v : nullptr, rr.n1, true, m.cfg.norm_eps, p1, ns1, dxn.p, s0.g, s0.written,
                       s1 ? s1->g : nullptr, s1 ? s1->written : false));
            s0.written = true;
            if (s1) s1->written = true;
            if (rr.has_sc) {
                // 1x1 shortcut over the concatenated input: one dgrad GEMM per source, accumulated in place
                const T* wT = static_cast<const T*>(rr.sc.wT);
                MRISR_REQUIRE(wT, "shortcut dgrad weights");
                GemmArgs g;
                g.a0 = dO; g.c0 = rr.cout; g.lda0 = rr.cout; g.w = wT; g.M = (int)x.rows(); g.N = x.C; g.K = rr.cout;
                g.resid = s0.g; g.ldr = x.C; g.out = s0.g; g.ldo = x.C;
                TRY(R::run_gemm(g));
                if (s1) {
                    GemmArgs g1;
                    g1.a0 = dO; g1.c0 = rr.cout; g1.lda0 = rr.cout; g1.w = wT + (size_t)x.C * rr.cout; g1.M = (int)x.rows();
                    g1.N = x1v.C; g1.K = rr.cout; g1.resid = s1->g; g1.ldr = x1v.C; g1.out = s1->g; g1.ldo = x1v.C;
                    TRY(R::run_gemm(g1));
                }
            } else if (!dry) {
                TRY(launch_add_inplace<T>(s0.g, dO, (long long)x.numel(), st));
            }
            m.arena.release(mk);
            return 0;
        });
        return 0;
    }

    // ---------------------------------------------------------------------------------------------
    // Transformer2DModel / BasicTransformerBlock
    // ---------------------------------------------------------------------------------------------
    int transformer_t(XfW& xw, const Act& x, Act* out) {
        const int C = xw.C, M = (int)x.rows();
        const HeadBuf& hb = m.head_buf(x.H * x.W, C);
        const int Bc = m.ws_B * m.ctx_len;
        Act o = new_act(x.B, x.H, x.W, C);
        if (!o.p) return 7;
        Act xn;
        float* gp = nullptr;
        int gns = 1;
        TRY(R::gn(x, nullptr, xw.norm, false, 1e-6f, &xn, &gp, &gns));
        auto rows = [&](int c) { return static_cast<T*>(alloc((size_t)M * c * sizeof(T))); };
        T *t0 = rows(C), *n1 = rows(C), *ao1 = rows(C), *t1 = rows(C), *n2 = rows(C), *ao2 = rows(C), *t2 = rows(C), *n3 = rows(C);
        T *ffpre = rows(8 * C), *ff = rows(4 * C), *t3 = rows(C);
        const size_t hsz = (size_t)hb.B * hb.H * hb.npad * hb.dpad * sizeof(T);
        void *q1 = alloc(hsz), *k1 = alloc(hsz), *vt1 = alloc(hsz), *q2 = alloc(hsz);
        if (!t3 || !q2) return 7;
        TRY(zero(q1, hsz)); TRY(zero(k1, hsz)); TRY(zero(vt1, hsz)); TRY(zero(q2, hsz));
        float *z_pi = nullptr, *z_qkv = nullptr, *z_o1 = nullptr, *z_q2 = nullptr, *z_o2 = nullptr, *z_f2 = nullptr, *z_po = nullptr;
        TRY(R::linear(xn.p, M, C, xw.proj_in, ACT_NONE, nullptr, 0, nullptr, t0, C, &z_pi));
        TRY(R::layernorm(t0, xw.ln1, M, C, n1));
        {
            GemmArgs g;
            R::heads_args(&g, hb, q1, 0, k1, 0, vt1, 1, hb.N, hb.npad);
            TRY(R::linear(n1, M, C, xw.qkv, ACT_NONE, nullptr, 0, &g, nullptr, 0, &z_qkv));
        }
        AttnRec a1;
        a1.q = q1; a1.k = k1; a1.vt = vt1; a1.nk = hb.N; a1.nkpad = hb.npad;
        TRY(attention_t(hb, &a1, ao1));
        TRY(R::linear(ao1, M, C, xw.out1, ACT_NONE, t0, C, nullptr, t1, C, &z_o1));
        TRY(R::layernorm(t1, xw.ln2, M, C, n2));
        {
            GemmArgs g;
            R::heads_args(&g, hb, q2, 0, nullptr, 0, nullptr, 0, hb.N, hb.npad);
            TRY(R::linear(n2, M, C, xw.q2, ACT_NONE, nullptr, 0, &g, nullptr, 0, &z_q2));
        }
        AttnRec a2;
        a2.q = q2; a2.k = xw.kc; a2.vt = xw.vtc; a2.nk = m.ctx_len; a2.nkpad = m.ctx_pad;
        TRY(attention_t(hb, &a2, ao2));
        TRY(R::linear(ao2, M, C, xw.out2, ACT_NONE, t1, C, nullptr, t2, C, &z_o2));
        TRY(R::layernorm(t2, xw.ln3, M, C, n3));
        TRY(R::linear(n3, M, C, xw.ff1, ACT_NONE, nullptr, 0, nullptr, ffpre, 8 * C));
        if (!dry) TRY(launch_geglu_fwd<T>(ffpre, ff, M, 4 * C, st));
        TRY(R::linear(ff, M, 4 * C, xw.ff2, ACT_NONE, t2, C, nullptr, t3, C, &z_f2));
        TRY(R::linear(t3, M, C, xw.proj_out, ACT_NONE, x.p, C, nullptr, o.p, C, &z_po));
        *out = o;
        const bool x_live = is_live(x);
        if (!x_live && !has_lora(xw) && !full()) return 0;
        live.insert(o.p);
        const XfW* xp = &xw;
        const HeadBuf hbv = hb;
        float* zkv = z_kv.count(&xw) ? z_kv[&xw] : nullptr;
        tape.push_back([=]() -> int {
            const XfW& w = *xp;
            auto it = slots.find(o.p);
            MRISR_REQUIRE(it != slots.end() && it->second.written, "transformer output has no gradient");
            void* dO = it->second.g;
            Slot* sx = x_live ? &slot(x) : nullptr;
            const size_t mk = m.arena.mark();
            auto brows = [&](size_t r, int c) { return static_cast<T*>(alloc(r * c * sizeof(T))); };
            T *dt = brows(M, C), *dn = brows(M, C), *dao = brows(M, C), *dq = brows(M, C);
            T *dff = brows(M, 4 * C), *dpre = brows(M, 8 * C), *dqkv = brows(M, 3 * C);
            const bool want_kv = w.kv2.R || full();
            T* dkv = want_kv ? brows(Bc, 2 * C) : nullptr;
            if (!dt || !dn || !dao || !dq || !dff || !dpre || !dqkv || (want_kv && !dkv)) return 7;
            // o = proj_out(t3) + x
            TRY(linear_bwd(w.proj_out, t3, C, z_po, dO, C, M, dt, false, true));
            // t3 = t2 + ff2(geglu(ff1(LN3(t2))))
            TRY(linear_bwd(w.ff2, ff, 4 * C, z_f2, dt, C, M, dff, false, true));
            if (!dry) TRY(launch_geglu_bwd<T>(ffpre, dff, dpre, M, 4 * C, st));
            TRY(linear_bwd(w.ff1, n3, C, nullptr, dpre, 8 * C, M, dn, false, true));
            TRY(ln_affine(t2, dn, w.ln3, M, C));
            if (!dry) TRY(launch_layernorm_bwd<T>(t2, dn, dt, w.ln3.g, M, C, 1e-5f, 1, st));
            // t2 = t1 + out2(attn(q2(LN2(t1)), K_ctx, V_ctx))
            TRY(linear_bwd(w.out2, ao2, C, z_o2, dt, C, M, dao, false, true));
            TRY(attention_bwd(hbv, a2, dao, dq, C, dkv, dkv ? dkv + C : nullptr, 2 * C));
            if (dkv) TRY(linear_bwd(w.kv2, m.ctx_rows, m.cfg.cross_attention_dim, zkv, dkv, 2 * C, Bc, nullptr, false, false));
            TRY(linear_bwd(w.q2, n2, C, z_q2, dq, C, M, dn, false, true));
            TRY(ln_affine(t1, dn, w.ln2, M, C));
            if (!dry) TRY(launch_layernorm_bwd<T>(t1, dn, dt, w.ln2.g, M, C, 1e-5f, 1, st));
            // t1 = t0 + out1(attn(qkv(LN1(t0))))
            TRY(linear_bwd(w.out1, ao1, C, z_o1, dt, C, M, dao, false, true));
            TRY(attention_bwd(hbv, a1, dao, dqkv, 3 * C, dqkv + C, dqkv + 2 * C, 3 * C));
            TRY(linear_bwd(w.qkv, n1, C, z_qkv, dqkv, 3 * C, M, dn, false, true));
            TRY(ln_affine(t0, dn, w.ln1, M, C));
            if (!dry) TRY(launch_layernorm_bwd<T>(t0, dn, dt, w.ln1.g, M, C, 1e-5f, 1, st));
            // t0 = proj_in(GN(x))
            if (sx) {
                TRY(linear_bwd(w.proj_in, xn.p, C, z_pi, dt, C, M, dn, false, true));
                TRY(gn_affine(x, dn, w.norm, false, 1e-6f, gp, gns));
                TRY(gn_bwd(x, nullptr, w.norm, false, 1e-6f, gp, gns, dn, sx->g, sx->written, nullptr, false));
                sx->written = true;
                if (!dry) TRY(launch_add_inplace<T>(sx->g, dO, (long long)x.numel(), st));
            } else if (w.proj_in.R) {
                TRY(linear_bwd(w.proj_in, xn.p, C, z_pi, dt, C, M, nullptr, false, false));
            }
            m.arena.release(mk);
            return 0;
        });
        return 0;
    }

    // ---------------------------------------------------------------------------------------------
    // whole network
    // ---------------------------------------------------------------------------------------------
    int set_context_t(const mrisr_tensor& ehs) {
        MRISR_REQUIRE(ehs.ndim == 3 && ehs.shape[2] == m.cfg.cross_attention_dim, "encoder_hidden_states shape");
        MRISR_REQUIRE((int)ehs.shape[0] == m.ws_B && (int)ehs.shape[1] == m.ctx_len, "context shape vs planned workspace");
        if (!dry) TRY(launch_nchw_to_nhwc<T>(ehs.data, ehs.dtype, m.ctx_rows, m.ws_B * m.ctx_len, m.cfg.cross_attention_dim, 1, 1, st));
        int rc = 0;
        for_each_xf(m, [&](XfW& x) {
            if (rc) return;
            const HeadBuf& hb = m.head_buf_for_C(x.C);
            GemmArgs g;
            R::heads_args(&g, hb, x.kc, 0, x.vtc, 1, nullptr, 0, m.ctx_len, m.ctx_pad);
            float* z = nullptr;
            rc = R::linear(m.ctx_rows, m.ws_B * m.ctx_len, m.cfg.cross_attention_dim, x.kv2, ACT_NONE, nullptr, 0, &g, nullptr, 0, &z);
            z_kv[&x] = z;
        });
        m.ctx_valid = false;  // the cache belongs to this step's adapters only
        return rc;
    }

    // x = x + feature (T2I-Adapter): when the backward reaches this point every consumer of x has contributed, so d(x) is
    // also d(feature); hand it to the caller's tensor (mrisr_train_set_intrablock_grads) for the adapter's own backward
    int export_feature_grad(const Act& x, int ib) {
        if (ib >= (int)m.d_intra.size() || !m.d_intra[ib].data) return 0;
        MRISR_REQUIRE(is_live(x), "adapter feature enters before any trainable block");
        const mrisr_tensor out = m.d_intra[ib];
        tape.push_back([=]() -> int {
            auto it = slots.find(x.p);
            MRISR_REQUIRE(it != slots.end() && it->second.written, "feature position has no gradient");
            Act gx = x;
            gx.p = it->second.g;
            return R::export_act(gx, out, 1.0f);
        });
        return 0;
    }

    // y = copy(x) + r (a ControlNet residual, diffusers: `down_block_res_sample + down_block_additional_residual`, out of place - the
    // encoder's own flow and its saved activations keep the un-added tensor): d(r) = d(y) -> the caller's tensor; d(x) += d(y)
    int add_residual_t(const Act& x, const mrisr_tensor& r, const mrisr_tensor* d_out, Act* y) {
        *y = new_act(x.B, x.H, x.W, x.C);
        if (!y->p) return 7;
        if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(y->p, x.p, x.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
        TRY(R::add_external(*y, r));
        live.insert(y->p);  // trainable upstream: the ControlNet
        const Act yy = *y;
        const bool x_live = is_live(x);
        const bool want = d_out && d_out->data;
        const mrisr_tensor out = want ? *d_out : mrisr_tensor{};
        tape.push_back([=]() -> int {
            auto it = slots.find(yy.p);
            MRISR_REQUIRE(it != slots.end() && it->second.written, "residual position has no gradient");
            Act gy = yy;
            gy.p = it->second.g;
            if (want) TRY(R::export_act(gy, out, 1.0f));
            if (x_live) {
                Slot& sx = slot(x);
                if (!dry) {
                    if (sx.written) TRY(launch_add_inplace<T>(sx.g, it->second.g, (long long)x.numel(), st));
                    else MRISR_CHECK_HIP(hipMemcpyAsync(sx.g, it->second.g, x.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
                }
                sx.written = true;
            }
            return 0;
        });
        return 0;
    }

    // dst.grad (+)= src.grad   (y = copy(x) [+ constant])
    int pass_grad(const Act& y, const Act& x) {
        if (!is_live(x)) return 0;
        live.insert(y.p);
        tape.push_back([=]() -> int {
            auto it = slots.find(y.p);
            MRISR_REQUIRE(it != slots.end() && it->second.written, "copy has no gradient");
            Slot& sx = slot(x);
            if (!dry) {
                if (sx.written) TRY(launch_add_inplace<T>(sx.g, it->second.g, (long long)x.numel(), st));
                else MRISR_CHECK_HIP(hipMemcpyAsync(sx.g, it->second.g, x.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
            }
            sx.written = true;
            return 0;
        });
        return 0;
    }

    // ---------------------------------------------------------------------------------------------
    // ControlNet with every parameter trainable (full-parameter mode).  Forward = Runner::controlnet_forward's sequence, recorded;
    // the backward is seeded with d(loss)/d(residual k) - what the UNet's step exported (mrisr_train_set_controlnet_residuals).
    // ---------------------------------------------------------------------------------------------
    // ControlNet condition embedding (App. A.6: conv_in, six blocks, conv_out; SiLU after all but the last) with the pre-activations kept
    // - the inference path fuses the SiLU into the conv epilogue and caches only the result.  Layers 1.. run zero-padded to 64-channel
    // multiples (model.hip conv_padded): gradients flow through the padded banks, the raw tensors receive their sub-blocks.
    std::vector<Act> ce_in, ce_pre;
    int set_cond_t(const mrisr_tensor& cond) {
        Act e, y;
        TRY(R::import_act(cond, &e, false));
        ce_in.clear();
        ce_pre.clear();
        for (size_t k = 0; k < m.ce.size(); ++k) {
            const bool last = k + 1 == m.ce.size();
            constexpr int KQ = 128 / (int)sizeof(T);
            ce_in.push_back(e);
            if (e.C % KQ == 0 && m.ce[k].cout % 4 == 0) TRY(R::conv3(e, nullptr, m.ce[k], m.ce_stride[k], 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
            else TRY(R::direct(e, m.ce[k], m.ce_stride[k], ACT_NONE, nullptr, &y));
            ce_pre.push_back(y);
            if (!last) {
                Act a = new_act(y.B, y.H, y.W, y.C);
                if (!a.p) return 7;
                if (!dry) TRY(launch_silu_fwd<T>(y.p, a.p, (long long)y.numel(), st));
                e = a;
            } else {
                e = y;
            }
        }
        MRISR_REQUIRE(e.numel() * sizeof(T) <= m.cond_emb_bytes, "condition embedding larger than planned");
        if (!dry) MRISR_CHECK_HIP(hipMemcpyAsync(m.cond_emb, e.p, e.numel() * sizeof(T), hipMemcpyDeviceToDevice, st));
        m.cond_valid = false;  // (the cache holds THIS step's weights only)
        return 0;
    }
    int set_cond_bwd(const Act& d_ce) {  // d_ce: gradient with respect to the embedding = the gradient of conv_in's output
        const size_t mk = m.arena.mark();
        Act dcur = d_ce;
        for (size_t kk = m.ce.size(); kk-- > 0;) {
            const ConvW& cw = m.ce[kk];
            const bool last = kk + 1 == m.ce.size();
            const Act& pre = ce_pre[kk];
            const Act& xin = ce_in[kk];
            Act dpre = dcur;
            if (!last) {
                dpre = new_act(pre.B, pre.H, pre.W, pre.C);
                if (!dpre.p) return 7;
                if (!dry) TRY(launch_silu_bwd<T>(dcur.p, pre.p, dpre.p, (long long)pre.numel(), st));
            }
            const RawParam* w = m.find(cw.name + ".weight");
            MRISR_REQUIRE(w && dpre.C == cw.cout && xin.C == cw.cin, "condition-embedding gradient: operands");
            TRY(wgrad(xin.p, xin.B, xin.H, xin.W, xin.C, dpre.p, dpre.C, 0, dpre.H, dpre.W, dpre.C, cw.ks, m.ce_stride[kk], gptr(cw.name + ".weight"),
                      gptr(cw.name + ".bias"), (int)w->shape[0], (int)w->shape[1]));
            if (kk > 0) {
                Act dx = new_act(xin.B, xin.H, xin.W, xin.C);
                if (!dx.p) return 7;
                TRY(conv_dgrad(dpre, cw, m.ce_stride[kk] == 2 ? 1 : 0, dx.p, false));
                dcur = dx;
            }
        }
        m.arena.release(mk);
        return 0;
    }

    std::vector<Act> cn_skips;
    Act cn_x;  // mid-block output
    int cn_B = 0;
    int cn_forward(const mrisr_tensor& sample, const long long* t_dev, int t_scalar, const mrisr_tensor& ehs, const mrisr_tensor& cond, float scale,
                   mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out) {
        m.arena.reset();
        const int B = (int)sample.shape[0];
        TRY(set_context_t(ehs));
        TRY(set_cond_t(cond));
        TRY(R::time_embed(t_dev, t_scalar, B));
        {
            const size_t nb = (size_t)(t_scalar ? 1 : B) * m.tproj_total * sizeof(float);
            m.d_tproj = static_cast<float*>(alloc(nb));
            TRY(zero(m.d_tproj, nb));
        }
        cn_B = B;
        Act s, x;
        TRY(R::import_act(sample, &s, false));
        Act ce;
        ce.p = m.cond_emb; ce.B = B; ce.H = s.H; ce.W = s.W; ce.C = m.cfg.block_out_channels[0];
        TRY(R::direct(s, m.conv_in, 1, ACT_NONE, &ce, &x));
        live.insert(x.p);
        {   // conv_in's own weights (4 input channels: the generic pixel-contraction path)
            const Act sv = s, xv = x;
            tape.push_back([=]() -> int {
                auto it = slots.find(xv.p);
                MRISR_REQUIRE(it != slots.end() && it->second.written, "conv_in output has no gradient");
                Act dy = xv; dy.p = it->second.g;
                TRY(wgrad_conv(sv, dy, m.conv_in, 1));
                return set_cond_bwd(dy);  // x = conv_in(sample) + embedding: the embedding's gradient is the same tensor
            });
        }
        cn_skips.clear();
        cn_skips.push_back(x);
        for (int i = 0; i < m.cfg.num_levels; ++i) {
            Level& lv = m.down[i];
            const bool has_attn = !lv.xf.empty();
            for (size_t j = 0; j < lv.res.size(); ++j) {
                Act y;
                TRY(resnet_t(lv.res[j], x, nullptr, &y));
                x = y;
                if (has_attn) {
                    TRY(transformer_t(lv.xf[j], x, &y));
                    x = y;
                }
                cn_skips.push_back(x);
            }
            if (lv.has_down) {
                Act y;
                TRY(R::conv3(x, nullptr, lv.down, 2, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
                live.insert(y.p);
                const ConvW* cw = &lv.down;
                const Act xin = x;
                tape.push_back([=]() -> int {
                    auto it = slots.find(y.p);
                    MRISR_REQUIRE(it != slots.end() && it->second.written, "downsample output has no gradient");
                    Slot& sx = slot(xin);
                    Act dy = y; dy.p = it->second.g;
                    TRY(wgrad_conv(xin, dy, *cw, 2));
                    TRY(conv_dgrad(dy, *cw, 1, sx.g, sx.written));
                    sx.written = true;
                    return 0;
                });
                x = y;
                cn_skips.push_back(x);
            }
        }
        {
            Act y;
            TRY(resnet_t(m.mid_r0, x, nullptr, &y));
            x = y;
            TRY(transformer_t(m.mid_xf, x, &y));
            x = y;
            TRY(resnet_t(m.mid_r1, x, nullptr, &y));
            x = y;
        }
        cn_x = x;
        MRISR_REQUIRE(n_down == (int)cn_skips.size(), "ControlNet output count");
        for (int k = 0; k <= n_down; ++k) {  // the zero convs (k == n_down: the mid block's)
            const Act& a = k < n_down ? cn_skips[k] : cn_x;
            const LinW& zw = k < n_down ? m.cn_down[k] : m.cn_mid;
            Act o = new_act(a.B, a.H, a.W, a.C);
            if (!o.p) return 7;
            TRY(R::linear(a.p, (int)a.rows(), a.C, zw, ACT_NONE, nullptr, 0, nullptr, o.p, o.C));
            mrisr_tensor* dst = k < n_down ? &down_out[k] : mid_out;
            if (dst && dst->data) TRY(R::export_act(o, *dst, scale));
        }
        return 0;
    }
    int cn_backward(const mrisr_tensor* d_down, int n_down, const mrisr_tensor* d_mid, float scale) {
        MRISR_REQUIRE(n_down == (int)cn_skips.size() && d_mid, "one gradient per ControlNet residual");
        for (int k = n_down; k >= 0; --k) {
            const Act& a = k < n_down ? cn_skips[k] : cn_x;
            const LinW& zw = k < n_down ? m.cn_down[k] : m.cn_mid;
            const mrisr_tensor& src = k < n_down ? d_down[k] : *d_mid;
            MRISR_REQUIRE(src.ndim == 4 && src.shape[0] == a.B && src.shape[1] == a.C && src.shape[2] == a.H && src.shape[3] == a.W, "residual gradient shape");
            Slot& sa = slot(a);  // (allocates the gradient buffer on first use: BEFORE the mark, it must outlive this scope)
            const size_t mk = m.arena.mark();
            Act dy;
            TRY(R::import_act(src, &dy, true));
            if (scale != 1.0f && !dry) TRY(launch_scale_inplace<T>(dy.p, scale, (long long)dy.numel(), st));
            TRY(linear_bwd(zw, a.p, a.C, nullptr, dy.p, a.C, (int)a.rows(), sa.g, sa.written, true));
            sa.written = true;
            m.arena.release(mk);
        }
        for (size_t i = tape.size(); i-- > 0;) TRY(tape[i]());
        return time_embed_bwd(cn_B);
    }

    int step(const mrisr_tensor& sample, const long long* t_dev, int t_scalar, const mrisr_tensor& ehs, const mrisr_tensor* intrablock,
             int n_intra, const mrisr_tensor& target, float* loss_dev, const mrisr_tensor* pred_out) {
        m.arena.reset();
        const int B = (int)sample.shape[0];
        TRY(set_context_t(ehs));
        TRY(R::time_embed(t_dev, t_scalar, B));
        Act s, x;
        TRY(R::import_act(sample, &s, false));
        TRY(R::direct(s, m.conv_in, 1, ACT_NONE, nullptr, &x));
        // ---- encoder ----
        std::vector<Act> skips;
        skips.push_back(x);
        int ib = 0;
        for (int i = 0; i < m.cfg.num_levels; ++i) {
            Level& lv = m.down[i];
            const bool has_attn = !lv.xf.empty();
            for (size_t j = 0; j < lv.res.size(); ++j) {
                Act y;
                TRY(resnet_t(lv.res[j], x, nullptr, &y));
                x = y;
                if (has_attn) {
                    TRY(transformer_t(lv.xf[j], x, &y));
                    x = y;
                    if (j + 1 == lv.res.size() && ib < n_intra) {
                        TRY(R::add_external(x, intrablock[ib]));  // x += feature: d(feature) = d(x)
                        TRY(export_feature_grad(x, ib++));
                    }
                }
                skips.push_back(x);
            }
            if (lv.has_down) {
                Act y;
                TRY(R::conv3(x, nullptr, lv.down, 2, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
                if (is_live(x)) {
                    live.insert(y.p);
                    const ConvW* cw = &lv.down;
                    const Act xin = x;
                    tape.push_back([=]() -> int {
                        auto it = slots.find(y.p);
                        MRISR_REQUIRE(it != slots.end() && it->second.written, "downsample output has no gradient");
                        Slot& sx = slot(xin);
                        Act dy = y; dy.p = it->second.g;
                        TRY(conv_dgrad(dy, *cw, 1, sx.g, sx.written));
                        sx.written = true;
                        return 0;
                    });
                }
                x = y;
                skips.push_back(x);
            }
            if (!has_attn && ib < n_intra) {
                // in place on the last skip (see Runner::encoder): d(feature) = d(x) with x's slot collecting the mid
                // block's AND the decoder's (skip) contributions
                TRY(R::add_external(x, intrablock[ib]));
                TRY(export_feature_grad(x, ib++));
            }
        }
        // ---- ControlNet residuals on the skips (after the whole down path, before the mid block: res_srdiff.py:73-78 /
        // diffusers `down_block_res_samples = [s + r for ...]`) ----
        if (!m.tr_down.empty()) {
            MRISR_REQUIRE(m.tr_down.size() == skips.size(), "one ControlNet residual per skip connection");
            for (size_t k = 0; k < skips.size(); ++k) {
                Act y;
                TRY(add_residual_t(skips[k], m.tr_down[k], k < m.d_tr_down.size() ? &m.d_tr_down[k] : nullptr, &y));
                skips[k] = y;
            }
        }
        // ---- mid ----
        {
            Act y;
            TRY(resnet_t(m.mid_r0, x, nullptr, &y));
            x = y;
            TRY(transformer_t(m.mid_xf, x, &y));
            x = y;
            TRY(resnet_t(m.mid_r1, x, nullptr, &y));
            x = y;
            if (m.has_tr_mid) {
                TRY(add_residual_t(x, m.tr_mid, m.d_tr_mid.data ? &m.d_tr_mid : nullptr, &y));
                x = y;
            }
        }
        // ---- decoder ----
        for (int i = 0; i < m.cfg.num_levels; ++i) {
            Level& lv = m.up[i];
            for (size_t j = 0; j < lv.res.size(); ++j) {
                Act sk = skips.back();
                skips.pop_back();
                Act y;
                TRY(resnet_t(lv.res[j], x, &sk, &y));
                x = y;
                if (!lv.xf.empty()) {
                    TRY(transformer_t(lv.xf[j], x, &y));
                    x = y;
                }
            }
            if (lv.has_up) {
                Act y;
                TRY(R::conv3(x, nullptr, lv.up, 1, 1, nullptr, 0, 1, nullptr, ACT_NONE, &y));
                if (is_live(x)) {
                    live.insert(y.p);
                    const ConvW* cw = &lv.up;
                    const Act xin = x;
                    tape.push_back([=]() -> int {
                        // y = conv(nearest_x2(x)):  dU = dgrad(dy) at the x2 size, dx = 2x2 sum pool of dU
                        auto it = slots.find(y.p);
                        MRISR_REQUIRE(it != slots.end() && it->second.written, "upsample output has no gradient");
                        Slot& sx = slot(xin);
                        const size_t mk = m.arena.mark();
                        Act dU = new_act(y.B, y.H, y.W, cw->cin);
                        if (!dU.p) return 7;
                        Act dy = y; dy.p = it->second.g;
                        TRY(conv_dgrad(dy, *cw, 0, dU.p, false));
                        if (!dry) TRY(launch_sumpool2<T>(dU.p, sx.g, xin.B, xin.H, xin.W, xin.C, sx.written ? 1 : 0, st));
                        sx.written = true;
                        m.arena.release(mk);
                        return 0;
                    });
                }
                x = y;
            }
        }
        // ---- head ----
        Act xn, y;
        float* gp = nullptr;
        int gns = 1;
        TRY(R::gn(x, nullptr, m.norm_out, true, m.cfg.norm_eps, &xn, &gp, &gns));
        if (m.conv_out.cout % 4 == 0) TRY(R::conv3(xn, nullptr, m.conv_out, 1, 0, nullptr, 0, 1, nullptr, ACT_NONE, &y));
        else TRY(R::direct(xn, m.conv_out, 1, ACT_NONE, nullptr, &y));
        if (pred_out) TRY(R::export_act(y, *pred_out, 1.0f));
        // ---- loss = mean((eps_hat - eps)^2), d(loss)/d(eps_hat) ----
        MRISR_REQUIRE(target.ndim == 4 && target.dtype == MRISR_F32 && target.layout == MRISR_NCHW && target.shape[0] == y.B &&
                          target.shape[1] == y.C && target.shape[2] == y.H && target.shape[3] == y.W,
                      "target: f32 NCHW of the prediction's shape");
        Act dy = new_act(y.B, y.H, y.W, y.C);
        if (!dy.p) return 7;
        if (!dry) {
            MRISR_CHECK_HIP(hipMemsetAsync(loss_dev, 0, sizeof(float), st));
            TRY(launch_mse_grad<T>(y.p, static_cast<const float*>(target.data), dy.p, loss_dev, y.B, y.C, y.H, y.W, st));
        }
        // ---- backward ----
        {
            MRISR_REQUIRE(is_live(x), "no trainable adapter reaches the output");
            Slot& sx = slot(x);
            const size_t mk = m.arena.mark();
            Act dxn = new_act(x.B, x.H, x.W, x.C);
            if (!dxn.p) return 7;
            TRY(conv_dgrad(dy, m.conv_out, 0, dxn.p, false));
            TRY(gn_bwd(x, nullptr, m.norm_out, true, m.cfg.norm_eps, gp, gns, dxn.p, sx.g, false, nullptr, false));
            sx.written = true;
            m.arena.release(mk);
        }
        for (size_t i = tape.size(); i-- > 0;) TRY(tape[i]());
        return 0;
    }
};

// ================================================================================================
// full-parameter training of a ControlNet handle
// ================================================================================================
template <typename T>
static int full_train_prepare_t(Model& m, hipStream_t st) {
    int err = 0;
    auto conv_dgrad = [&](ConvW& c) {
        if (!c.w || c.ks != 3 || c.wd) return;
        const RawParam* w = m.find(c.name + ".weight");
        if (!w) { err = 3; set_error("missing parameter: " + c.name + ".weight"); return; }
        c.wd = m.new_packed((size_t)w->numel() * sizeof(T), false);
        if (!c.wd) { err = 4; return; }
        long long blocks = (w->numel() + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(pack_conv_dgrad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float*>(w->data->p),
                           static_cast<T*>(c.wd), c.cout, c.cin);
    };
    auto lin_t = [&](LinW& l) {
        if (!l.w) return;
        if (!l.wT) l.wT = m.new_packed((size_t)l.n * l.k * sizeof(T), false);
        if (!l.wT) { err = 4; return; }
        if (launch_transpose<T>(l.w, l.wT, l.n, l.k, l.k, l.n, 0, 0, 1, l.n, st)) err = 5;
    };
    for (auto& lv : m.down) {
        for (auto& r : lv.res) { conv_dgrad(r.c1); conv_dgrad(r.c2); if (r.has_sc) lin_t(r.sc); }
        if (lv.has_down) conv_dgrad(lv.down);
        for (auto& x : lv.xf) for (LinW* l : {&x.proj_in, &x.qkv, &x.out1, &x.q2, &x.out2, &x.ff1, &x.ff2, &x.proj_out}) lin_t(*l);
    }
    for (ResW* r : {&m.mid_r0, &m.mid_r1}) { conv_dgrad(r->c1); conv_dgrad(r->c2); if (r->has_sc) lin_t(r->sc); }
    for (LinW* l : {&m.mid_xf.proj_in, &m.mid_xf.qkv, &m.mid_xf.out1, &m.mid_xf.q2, &m.mid_xf.out2, &m.mid_xf.ff1, &m.mid_xf.ff2, &m.mid_xf.proj_out}) lin_t(*l);
    for (auto& l : m.cn_down) lin_t(l);
    lin_t(m.cn_mid);
    for (size_t k = 1; k < m.ce.size(); ++k) {  // condition embedding: zero-padded dgrad banks (layer 0 needs no input gradient)
        ConvW& c = m.ce[k];
        const RawParam* w = m.find(c.name + ".weight");
        if (!w || c.ks != 3) { err = 3; set_error("condition embedding: missing / unexpected layer " + c.name); break; }
        if (!c.wd) c.wd = m.new_packed((size_t)c.cout * c.cin * 9 * sizeof(T), false);
        if (!c.wd) { err = 4; break; }
        if (launch_pack_conv_dgrad_padded<T>(static_cast<const float*>(w->data->p), c.wd, (int)w->shape[0], (int)w->shape[1], c.cout, c.cin, st)) err = 5;
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return err;
}

int Model::full_train_prepare(hipStream_t st) {
    MRISR_REQUIRE(finalized && is_controlnet, "full-parameter training is implemented for a finalized ControlNet handle");
    MRISR_REQUIRE(cfg.lora_rank == 0, "full-parameter training: no adapters on the ControlNet");
    full_off.clear();
    full_trainables.clear();
    full_unsupported.clear();
    long long off = 0;
    for (auto& kv : raw) {  // key order: the layout is a property of the parameter names alone
        const RawParam& r = kv.second;
        const long long rows = r.shape.empty() ? 1 : r.shape[0];
        full_off[kv.first] = off;
        full_trainables.push_back({kv.first, off, r.numel(), (int)rows, (int)(r.numel() / (rows ? rows : 1))});
        off += r.numel();
        // what this build differentiates: every conv / linear weight and bias of the encoder, mid block and zero convs, the affine
        // parameters of every norm, the time-embedding MLP and its per-block projections, the condition embedding - everything
        const std::string& k = kv.first;
        auto has = [&](const char* s) { return k.find(s) != std::string::npos; };
        (void)has;  // (every tensor is differentiated; full_unsupported stays empty)
    }
    n_full = off;
    return cfg.compute_dtype == MRISR_F32 ? full_train_prepare_t<float>(*this, st) : full_train_prepare_t<bf16>(*this, st);
}

int Model::full_train_bind(float* theta_dev, float* grad_dev, int init_from_model, hipStream_t st) {
    if (full_off.empty()) TRY(full_train_prepare(st));
    MRISR_REQUIRE(theta_dev && grad_dev, "theta / grad device buffers");
    full_theta = theta_dev;
    full_grad = grad_dev;
    if (init_from_model)
        for (auto& t : full_trainables)
            MRISR_CHECK_HIP(hipMemcpyAsync(theta_dev + t.offset, raw.at(t.key).data->p, (size_t)t.numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// after an optimiser step: theta -> the f32 masters, every packed weight (forward and dgrad copies) rebuilt in place
int Model::full_train_refresh(hipStream_t st) {
    MRISR_REQUIRE(full_theta, "bind the trainable vector first");
    for (auto& t : full_trainables)
        MRISR_CHECK_HIP(hipMemcpyAsync(raw.at(t.key).data->p, full_theta + t.offset, (size_t)t.numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    // re-pack INTO the existing buffers: the forward copies (finalize) and, still in the same buffer sequence, the dgrad copies
    repacking = true;
    repack_cursor = 0;
    int rc = finalize(st);
    if (!rc) rc = cfg.compute_dtype == MRISR_F32 ? full_train_prepare_t<float>(*this, st) : full_train_prepare_t<bf16>(*this, st);
    repacking = false;
    // finalize rebuilt the module structs: the per-block workspace pointers planned into them (cached context K / V^T, ...) are gone ->
    // plan again at the next forward (same sizes: no allocation, the dry passes only)
    ws_key.clear();
    train_ws_key.clear();
    cond_valid = false;
    ctx_valid = false;
    return rc;
}

template <typename T>
static int cn_forward_t(Model& m, const mrisr_tensor& sample, const long long* t, int t_scalar, const mrisr_tensor& ehs, const mrisr_tensor& cond,
                        float scale, mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out, hipStream_t st) {
    m.full_tape.reset();
    if (m.train_ws_key != m.ws_key) {  // dry pass of forward + backward sizes the arena exactly
        ++m.ws_gen;
        m.arena.dry = true;
        m.arena.reset();
        m.arena.peak = 0;
        int rc;
        {
            Trainer<T> tr(m, st, true);
            rc = tr.cn_forward(sample, t, t_scalar, ehs, cond, scale, down_out, n_down, mid_out);
            if (!rc) {
                std::vector<mrisr_tensor> dd(down_out, down_out + n_down);
                rc = tr.cn_backward(dd.data(), n_down, mid_out, scale);
            }
        }
        m.arena.dry = false;
        if (rc) return rc;
        MRISR_CHECK_HIP(hipStreamSynchronize(st));
        TRY(m.arena.buf.reserve(m.arena.peak + 4096, false));
        m.arena.reset();
        m.train_ws_key = m.ws_key;
    }
    auto tr = std::make_shared<Trainer<T>>(m, st, false);
    TRY(tr->cn_forward(sample, t, t_scalar, ehs, cond, scale, down_out, n_down, mid_out));
    m.full_tape = tr;
    return 0;
}

int Model::controlnet_train_forward(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs, const mrisr_tensor* cond,
                                    float scale, mrisr_tensor* down_out, int n_down, mrisr_tensor* mid_out, hipStream_t st) {
    MRISR_REQUIRE(is_controlnet && full_grad, "bind the ControlNet's trainable vector first (mrisr_controlnet_train_bind)");
    MRISR_REQUIRE(sample && sample->ndim == 4 && timestep && timestep->dtype == MRISR_I64 && ehs && ehs->ndim == 3 && cond && down_out && mid_out,
                  "ControlNet training forward: sample, timestep, encoder_hidden_states, controlnet_cond, outputs");
    const int B = (int)sample->shape[0], h = (int)sample->shape[2], w = (int)sample->shape[3];
    const int t_scalar = timestep->ndim == 0 || timestep->shape[0] == 1;
    MRISR_REQUIRE(t_scalar || timestep->shape[0] == B, "timestep length");
    keep = true;
    int rc = ensure_workspace(B, h, w, (int)ehs->shape[1], st);
    if (!rc) {
        const long long* t = static_cast<const long long*>(timestep->data);
        rc = cfg.compute_dtype == MRISR_F32 ? cn_forward_t<float>(*this, *sample, t, t_scalar, *ehs, *cond, scale, down_out, n_down, mid_out, st)
                                            : cn_forward_t<bf16>(*this, *sample, t, t_scalar, *ehs, *cond, scale, down_out, n_down, mid_out, st);
    }
    keep = false;
    return rc;
}

int Model::controlnet_train_backward(const mrisr_tensor* d_down, int n_down, const mrisr_tensor* d_mid, float scale, hipStream_t st) {
    MRISR_REQUIRE(full_tape, "run mrisr_controlnet_train_forward first (its recorded forward is consumed by ONE backward)");
    keep = true;
    int rc;
    if (cfg.compute_dtype == MRISR_F32) rc = static_cast<Trainer<float>*>(full_tape.get())->cn_backward(d_down, n_down, d_mid, scale);
    else rc = static_cast<Trainer<bf16>*>(full_tape.get())->cn_backward(d_down, n_down, d_mid, scale);
    keep = false;
    full_tape.reset();
    return rc;
}

template <typename T>
static int train_step_t(Model& m, const mrisr_tensor& sample, const long long* t, int t_scalar, const mrisr_tensor& ehs,
                        const mrisr_tensor* intrablock, int n_intra, const mrisr_tensor& target, float* loss_dev,
                        const mrisr_tensor* pred_out, hipStream_t st) {
    if (m.train_ws_key != m.ws_key) {
        // dry pass of forward + backward sizes the arena exactly
        ++m.ws_gen;  // the arena may be reallocated below
        m.arena.dry = true;
        m.arena.reset();
        m.arena.peak = 0;
        int rc;
        {
            Trainer<T> tr(m, st, true);
            rc = tr.step(sample, t, t_scalar, ehs, intrablock, n_intra, target, loss_dev, pred_out);
        }
        m.arena.dry = false;
        if (rc) return rc;
        MRISR_CHECK_HIP(hipStreamSynchronize(st));
        TRY(m.arena.buf.reserve(m.arena.peak + 4096, false));
        m.arena.reset();
        m.train_ws_key = m.ws_key;
    }
    Trainer<T> tr(m, st, false);
    return tr.step(sample, t, t_scalar, ehs, intrablock, n_intra, target, loss_dev, pred_out);
}

int Model::train_step(const mrisr_tensor* sample, const mrisr_tensor* timestep, const mrisr_tensor* ehs, const mrisr_tensor* intrablock,
                      int n_intra, const mrisr_tensor* target, float* loss_dev, mrisr_tensor* pred_out, hipStream_t st) {
    MRISR_REQUIRE(train_ready && (n_trainable == 0 || (theta && grad)), "bind the trainable vector first (mrisr_train_bind)");
    MRISR_REQUIRE(sample && sample->ndim == 4 && sample->shape[1] == cfg.in_channels, "sample must be [B, in_channels, h, w]");
    MRISR_REQUIRE(timestep && timestep->dtype == MRISR_I64 && timestep->ndim <= 1, "timestep: device int64, 0-dim or [B]");
    MRISR_REQUIRE(ehs && ehs->ndim == 3, "encoder_hidden_states must be [B, L, D]");
    MRISR_REQUIRE(target && loss_dev, "target / loss");
    const int B = (int)sample->shape[0], h = (int)sample->shape[2], w = (int)sample->shape[3];
    const int t_scalar = timestep->ndim == 0 || timestep->shape[0] == 1;
    MRISR_REQUIRE(t_scalar || timestep->shape[0] == B, "timestep length");
    keep = true;
    int rc = ensure_workspace(B, h, w, (int)ehs->shape[1], st);
    if (!rc) {
        const long long* t = static_cast<const long long*>(timestep->data);
        rc = cfg.compute_dtype == MRISR_F32
                 ? train_step_t<float>(*this, *sample, t, t_scalar, *ehs, intrablock, n_intra, *target, loss_dev, pred_out, st)
                 : train_step_t<bf16>(*this, *sample, t, t_scalar, *ehs, intrablock, n_intra, *target, loss_dev, pred_out, st);
    }
    keep = false;
    return rc;
}

}  // namespace mrisr
