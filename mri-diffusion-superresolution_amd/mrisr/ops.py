"""Single-op entry points of the C ABI (``mrisr_op_*``): the same kernels the models launch, exposed so the parity
tests can check each one against a plain PyTorch reference of the op."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _nhwc(x: torch.Tensor):
    """torch NCHW tensor -> (contiguous NHWC buffer, descriptor with logical NCHW shape)."""
    buf = x.permute(0, 2, 3, 1).contiguous()
    return buf, L.as_tensor(buf, L.MRISR_NHWC, shape=x.shape)


def _f32(x: Optional[torch.Tensor]):
    return None if x is None else C.c_void_p(x.detach().to(torch.float32).contiguous().data_ptr())


def conv3x3(x, weight, bias=None, x2=None, stride=1, upsample=False, act=L.ACT_NONE, splitk=0, tile=0, subpix=False):
    """x [B,C,H,W] (bf16/f32, cuda), weight [Cout,Cin,3,3] f32.  Returns NCHW tensor of x.dtype.  ``upsample``: nearest x2 first
    (diffusers Upsample2D); with ``subpix`` in its sub-pixel form (four 2 x 2 parity convs on the low-resolution input, bf16)."""
    xb, tx = _nhwc(x)
    t2 = None
    if x2 is not None:
        x2b, t2 = _nhwc(x2)
    w = weight.detach().to(torch.float32).contiguous()
    b = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    B, _, H, W = x.shape
    Hc, Wc = (H * 2, W * 2) if upsample else (H, W)
    Ho, Wo = (Hc - 1) // stride + 1, (Wc - 1) // stride + 1
    cout = w.shape[0]
    yb = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device)
    ty = L.as_tensor(yb, L.MRISR_NHWC, shape=(B, cout, Ho, Wo))
    L.check(L.lib().mrisr_op_conv3x3(C.byref(tx), C.byref(t2) if t2 else None, C.c_void_p(w.data_ptr()),
                                     C.c_void_p(b.data_ptr()) if b is not None else None, cout, stride,
                                     (2 if subpix else 1) if upsample else 0, act, splitk, tile, C.byref(ty), L.stream_ptr()))
    return yb.permute(0, 3, 1, 2)


def linear(x, weight, bias=None, act=L.ACT_NONE, splitk=0, tile=0):
    """x [M,K], weight [N,K] f32 -> [M,N] (GEGLU: [M,N/2])."""
    x = x.contiguous()
    w = weight.detach().to(torch.float32).contiguous()
    b = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    n = w.shape[0]
    y = torch.empty((x.shape[0], n // 2 if act == L.ACT_GEGLU else n), dtype=x.dtype, device=x.device)
    tx, ty = L.as_tensor(x), L.as_tensor(y)
    L.check(L.lib().mrisr_op_linear(C.byref(tx), C.c_void_p(w.data_ptr()),
                                    C.c_void_p(b.data_ptr()) if b is not None else None, n, act, splitk, tile,
                                    C.byref(ty), L.stream_ptr()))
    return y


def ln_linear(x, gamma, beta, weight, bias=None, act=L.ACT_NONE):
    """LayerNorm(x) W^T + bias, the normalisation fused into the row-panel GEMM (bf16, K = 320 / 640)."""
    x = x.contiguous()
    w = weight.detach().to(torch.float32).contiguous()
    b = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    ga, be = (t.detach().to(torch.float32).contiguous() for t in (gamma, beta))
    n = w.shape[0]
    y = torch.empty((x.shape[0], n // 2 if act == L.ACT_GEGLU else n), dtype=x.dtype, device=x.device)
    tx, ty = L.as_tensor(x), L.as_tensor(y)
    L.check(L.lib().mrisr_op_ln_linear(C.byref(tx), C.c_void_p(ga.data_ptr()), C.c_void_p(be.data_ptr()), C.c_void_p(w.data_ptr()),
                                       C.c_void_p(b.data_ptr()) if b is not None else None, n, act, C.byref(ty), L.stream_ptr()))
    return y


def mlp(x, gamma, beta, w1, b1, w2, b2, residual=True):
    """x + FF2(GEGLU(FF1(LayerNorm(x)))) in one kernel (bf16 rows of width 320; the hidden activation stays on-chip):
    w1 [2H][320] = diffusers ff.net.0.proj.weight (value half then gate half), w2 [320][H] = ff.net.2.weight."""
    x = x.contiguous()
    f = lambda t: t.detach().to(torch.float32).contiguous() if t is not None else None
    ga, be, w1f, b1f, w2f, b2f = (f(t) for t in (gamma, beta, w1, b1, w2, b2))
    hidden = w2f.shape[1]
    y = torch.empty((x.shape[0], w2f.shape[0]), dtype=x.dtype, device=x.device)
    tx, ty = L.as_tensor(x), L.as_tensor(y)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    L.check(L.lib().mrisr_op_mlp(C.byref(tx), p(ga), p(be), p(w1f), p(b1f), p(w2f), p(b2f), hidden, 1 if residual else 0,
                                 C.byref(ty), L.stream_ptr()))
    return y


def linear_fp8(x, weight, bias=None, act=L.ACT_NONE, gamma=None, beta=None):
    """[LayerNorm(x)] W^T + bias with OCP e4m3 operands on the fp8 MFMA (row-panel kernel; bf16 in / out, K = 320 / 640)."""
    x = x.contiguous()
    w = weight.detach().to(torch.float32).contiguous()
    b = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    ga = gamma.detach().to(torch.float32).contiguous() if gamma is not None else None
    be = beta.detach().to(torch.float32).contiguous() if beta is not None else None
    n = w.shape[0]
    y = torch.empty((x.shape[0], n // 2 if act == L.ACT_GEGLU else n), dtype=x.dtype, device=x.device)
    tx, ty = L.as_tensor(x), L.as_tensor(y)
    L.check(L.lib().mrisr_op_linear_fp8(C.byref(tx), C.c_void_p(ga.data_ptr()) if ga is not None else None,
                                        C.c_void_p(be.data_ptr()) if be is not None else None, C.c_void_p(w.data_ptr()),
                                        C.c_void_p(b.data_ptr()) if b is not None else None, n, act, C.byref(ty), L.stream_ptr()))
    return y


def groupnorm(x, gamma, beta, groups=32, eps=1e-5, silu=False, x2=None):
    xb, tx = _nhwc(x)
    t2 = None
    ctot = x.shape[1]
    if x2 is not None:
        x2b, t2 = _nhwc(x2)
        ctot += x2.shape[1]
    g = gamma.detach().to(torch.float32).contiguous()
    b = beta.detach().to(torch.float32).contiguous()
    B, _, H, W = x.shape
    yb = torch.empty((B, H, W, ctot), dtype=x.dtype, device=x.device)
    ty = L.as_tensor(yb, L.MRISR_NHWC, shape=(B, ctot, H, W))
    L.check(L.lib().mrisr_op_groupnorm(C.byref(tx), C.byref(t2) if t2 else None, C.c_void_p(g.data_ptr()),
                                       C.c_void_p(b.data_ptr()), groups, C.c_float(eps), 1 if silu else 0,
                                       C.byref(ty), L.stream_ptr()))
    return yb.permute(0, 3, 1, 2)


def layernorm(x, gamma, beta, eps=1e-5):
    x = x.contiguous()
    g = gamma.detach().to(torch.float32).contiguous()
    b = beta.detach().to(torch.float32).contiguous()
    y = torch.empty_like(x)
    tx, ty = L.as_tensor(x), L.as_tensor(y)
    L.check(L.lib().mrisr_op_layernorm(C.byref(tx), C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()), C.c_float(eps),
                                       C.byref(ty), L.stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return y


def attention(q, k, v, heads, flash=True, fp8=False):
    """q [B,N,C], k/v [B,Nk,C] -> [B,N,C]  (softmax(q k^T / sqrt(d)) v per head).  ``fp8``: Q K^T and P V with OCP e4m3 operands
    (per-head scales, f32 softmax) - BASELINE configs[4]."""
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty_like(q)
    tq, tk, tv, to = (L.as_tensor(t) for t in (q, k, v, out))
    L.check(L.lib().mrisr_op_attention(C.byref(tq), C.byref(tk), C.byref(tv), heads, 2 if fp8 else (1 if flash else 0), C.byref(to),
                                       L.stream_ptr()))
    return out


def attention_backward(q, k, v, dout, heads):
    """Gradients (dq, dk, dv) of ``attention`` w.r.t. its bf16 inputs for an upstream ``dout`` [B,N,C]: the flash
    forward (keeping the log-sum-exp) followed by the two backward kernels the fine-tuning step uses."""
    q, k, v, dout = (t.contiguous() for t in (q, k, v, dout))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ts = [L.as_tensor(t) for t in (q, k, v, dout, dq, dk, dv)]
    L.check(L.lib().mrisr_op_attention_bwd(C.byref(ts[0]), C.byref(ts[1]), C.byref(ts[2]), C.byref(ts[3]), heads,
                                           C.byref(ts[4]), C.byref(ts[5]), C.byref(ts[6]), L.stream_ptr()))
    return dq, dk, dv
