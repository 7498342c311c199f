"""GPU: each HIP kernel class against a plain PyTorch fp32 reference of the same op, through the C ABI
(mrisr_op_*).  Tolerances: f32 path 1e-3 relative (north_star); bf16 path compared against the fp32 reference of the
bf16-rounded inputs with a relative-L2 bound (bf16 has 8 mantissa bits)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f32": 1e-3, "bf16": 1.2e-2}


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DT[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,tile,splitk", [
    (256, 128, 128, 1, 1), (300, 320, 320, 0, 1), (1000, 64, 192, 2, 1), (77, 640, 768, 3, 1),
    (512, 1280, 2560, 4, 1), (512, 256, 4096, 1, 4), (130, 68, 256, 4, 3), (2048, 320, 320, 2, 1),
])
def test_linear(dt, M, N, K, tile, splitk):
    from mrisr import ops
    x, w, b = _rnd((M, K), dt, 1), _rnd((N, K), "f32", 2, K ** -0.5), _rnd((N,), "f32", 3)
    wq = w.to(DT[dt]).float()
    ref = F.linear(x.float(), wq, b)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile, splitk=splitk)
    assert rel(y, ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_linear_geglu_and_activations(dt):
    from mrisr import _lib as L
    from mrisr import ops
    M, C = 384, 128
    x, w, b = _rnd((M, C), dt, 4), _rnd((8 * C, C), "f32", 5, C ** -0.5), _rnd((8 * C,), "f32", 6)
    h = F.linear(x.float(), w.to(DT[dt]).float(), b)
    u, g = h.chunk(2, dim=-1)
    ref = u * F.gelu(g)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU)
    assert y.shape == (M, 4 * C) and rel(y, ref) < TOL[dt]
    for act, fn in ((L.ACT_RELU, F.relu), (L.ACT_SILU, F.silu)):
        y = ops.linear(x.cuda(), w[:256].cuda(), b[:256].cuda(), act=act)
        assert rel(y, fn(h[:, :256])) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H,stride,ups,tile,splitk", [
    (2, 64, 64, 16, 1, False, 0, 1), (1, 128, 320, 12, 1, False, 1, 1), (2, 64, 128, 16, 2, False, 3, 1),
    (2, 128, 64, 8, 1, True, 2, 1), (3, 256, 256, 4, 1, False, 1, 5), (1, 64, 64, 5, 1, False, 4, 2),
    (1, 320, 320, 32, 1, False, 2, 1),
])
def test_conv3x3(dt, B, Cin, Cout, H, stride, ups, tile, splitk):
    from mrisr import ops
    x = _rnd((B, Cin, H, H), dt, 7)
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 8, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 9)
    xi = x.float()
    if ups:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xi, w.to(DT[dt]).float(), b, stride=stride, padding=1)
    y = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=ups, tile=tile, splitk=splitk)
    assert y.shape == ref.shape and rel(y, ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv3x3_skip_concat(dt):
    from mrisr import ops
    x1, x2 = _rnd((2, 128, 8, 8), dt, 10), _rnd((2, 64, 8, 8), dt, 11)
    w, b = _rnd((128, 192, 3, 3), "f32", 12, (9 * 192) ** -0.5), _rnd((128,), "f32", 13)
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), w.to(DT[dt]).float(), b, padding=1)
    y = ops.conv3x3(x1.cuda(), w.cuda(), b.cuda(), x2=x2.cuda())
    assert rel(y, ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,C,C2,H,silu", [(2, 320, 0, 16, True), (3, 64, 0, 8, False), (2, 1280, 640, 8, True),
                                           (1, 2560, 0, 4, True), (2, 640, 320, 32, True), (32, 64, 64, 4, False)])
def test_groupnorm(dt, B, C, C2, H, silu):
    from mrisr import ops
    x = _rnd((B, C, H, H), dt, 14) * 1.7 + 0.3
    x2 = _rnd((B, C2, H, H), dt, 15) if C2 else None
    g, b = 1 + 0.2 * _rnd((C + C2,), "f32", 16), 0.2 * _rnd((C + C2,), "f32", 17)
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    ref = F.group_norm(xin, 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    y = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 32, 1e-5, silu, x2=x2.cuda() if x2 is not None else None)
    assert rel(y, ref) < TOL[dt]


@pytest.mark.parametrize("B,C,C2,H,W,silu", [
    (8, 320, 0, 32, 32, True), (8, 320, 320, 32, 32, True), (3, 640, 0, 16, 16, True), (8, 1280, 640, 16, 16, False),
    (8, 640, 320, 16, 16, True), (8, 1280, 0, 8, 8, True), (16, 1280, 1280, 8, 8, True), (8, 1280, 640, 8, 8, True),
    (5, 1280, 0, 4, 4, True), (8, 1280, 1280, 4, 4, False), (8, 640, 320, 32, 32, True), (8, 64, 0, 5, 7, True),
    (2, 128, 128, 64, 64, True), (8, 512, 0, 24, 40, False)])
def test_groupnorm_one_pass_kernel(B, C, C2, H, W, silu):
    """The one-pass bf16 GroupNorm (image slab in registers) on every UNet / VAE geometry: group widths 2...80 (vectors
    straddling groups at 10 / 20 / 60 channels per group), skip-concat inside a slab, non-square and odd images, batches that
    are / are not a multiple of the XCD count, and the (960 channels at 32x32) shape that falls back to the two-kernel path.
    Checked against torch (bf16 tolerance) and against the two-kernel path (same f32 statistics -> at most 1 bf16 ulp apart)."""
    import ctypes as C_
    from mrisr import _lib as L
    from mrisr import ops
    x = _rnd((B, C, H, W), "bf16", 114) * 1.7 + 0.3
    x2 = _rnd((B, C2, H, W), "bf16", 115) * 0.6 - 1.0 if C2 else None
    g, b = 1 + 0.2 * _rnd((C + C2,), "f32", 116), 0.2 * _rnd((C + C2,), "f32", 117)
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    ref = F.group_norm(xin, 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    lib = L.lib()
    outs = []
    try:
        for on in (1, 0):
            lib.mrisr_debug_gn_fused(C_.c_int(on))
            outs.append(ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 32, 1e-5, silu, x2=x2.cuda() if x2 is not None else None).float().cpu())
    finally:
        lib.mrisr_debug_gn_fused(C_.c_int(1))
    assert rel(outs[0], ref) < TOL["bf16"] and rel(outs[1], ref) < TOL["bf16"]
    d = (outs[0] - outs[1]).abs()
    assert float((d / outs[1].abs().clamp_min(0.05)).max()) < 2 ** -6, float((d / outs[1].abs().clamp_min(0.05)).max())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,C", [(100, 320), (64, 640), (33, 1280), (7, 64)])
def test_layernorm(dt, M, C):
    from mrisr import ops
    x = _rnd((M, C), dt, 18) * 2 + 0.5
    g, b = 1 + 0.2 * _rnd((C,), "f32", 19), 0.2 * _rnd((C,), "f32", 20)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5)
    assert rel(ops.layernorm(x.cuda(), g.cuda(), b.cuda()), ref) < TOL[dt]


def _sdpa(q, k, v, H):
    B, N, C = q.shape
    d = C // H
    qh, kh, vh = (t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("dt,flash", [("f32", False), ("bf16", False), ("bf16", True)])
@pytest.mark.parametrize("B,N,Nk,C,H", [(2, 256, 256, 320, 8), (1, 1024, 1024, 320, 8), (2, 64, 64, 1280, 8),
                                        (2, 16, 16, 1280, 8), (2, 256, 77, 640, 8), (1, 200, 77, 64, 8),
                                        (1, 64, 64, 256, 8)])
def test_attention(dt, flash, B, N, Nk, C, H):
    from mrisr import ops
    q, k, v = _rnd((B, N, C), dt, 21), _rnd((B, Nk, C), dt, 22), _rnd((B, Nk, C), dt, 23)
    ref = _sdpa(q, k, v, H)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, flash=flash)
    assert rel(y, ref) < (2e-2 if dt == "bf16" else 1e-3)


def test_attention_flash_online_softmax_rescale_branch():
    """Force the running max to jump late in the key sequence (guide rule 26): one key aligned with every query
    sits in the LAST tile, so all earlier tiles' accumulators must be rescaled by exp(m_old - m_new)."""
    from mrisr import ops
    B, N, C, H = 1, 256, 320, 8
    q, k, v = _rnd((B, N, C), "bf16", 24), _rnd((B, N, C), "bf16", 25), _rnd((B, N, C), "bf16", 26)
    k[:, 250] = q.float().mean(1) .to(torch.bfloat16) * 6.0
    ref = _sdpa(q, k, v, H)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, flash=True)
    assert rel(y, ref) < 2e-2


@pytest.mark.parametrize("B,N,Nk,C,H", [(2, 256, 256, 320, 8), (1, 1024, 1024, 320, 8), (2, 64, 64, 1280, 8),
                                        (2, 16, 16, 1280, 8), (2, 256, 77, 640, 8), (1, 200, 77, 64, 8),
                                        (1, 64, 64, 256, 8), (1, 100, 130, 1024, 8)])
def test_attention_backward_matches_autograd(B, N, Nk, C, H):
    """Flash backward (P recomputed from the log-sum-exp, dQ pass + dK/dV pass) against torch autograd of the f32 SDPA on
    the same bf16-rounded inputs; ragged query / key counts, every padded head dim the SD-1.5 levels use."""
    from mrisr import ops
    q, k, v = _rnd((B, N, C), "bf16", 41), _rnd((B, Nk, C), "bf16", 42), _rnd((B, Nk, C), "bf16", 43)
    do = _rnd((B, N, C), "bf16", 44)
    with torch.enable_grad():
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
        _sdpa(qf, kf, vf, H).backward(do.float())
    dq, dk, dv = ops.attention_backward(q.cuda(), k.cuda(), v.cuda(), do.cuda(), H)
    assert rel(dq, qf.grad) < 2e-2, ("dq", rel(dq, qf.grad))
    assert rel(dk, kf.grad) < 2e-2, ("dk", rel(dk, kf.grad))
    assert rel(dv, vf.grad) < 2e-2, ("dv", rel(dv, vf.grad))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,stride", [(2, 4, 320, 32, 32, 1), (3, 4, 64, 7, 5, 1), (1, 4, 320, 64, 64, 1), (5, 4, 32, 3, 3, 1),
                                                   (2, 3, 128, 16, 16, 1), (2, 4, 320, 16, 16, 2), (1, 8, 24, 9, 9, 1)])
def test_conv3x3_small_fan_in(dt, B, Cin, Cout, H, W, stride):
    """Fan-in below one K tile: conv_in (4 channels; bf16 runs the matrix-core form - image borders, a pixel count that is not a
    multiple of 16, one wave tile spanning several image rows) and the generic direct kernels (3 / 8 channels, stride 2, f32)."""
    from mrisr import ops
    x = _rnd((B, Cin, H, W), dt, 61)
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 62, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 63)
    wq = w.to(torch.bfloat16).float() if dt == "bf16" else w
    ref = F.conv2d(x.float(), wq, b, stride=stride, padding=1)
    y = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), stride=stride)
    assert y.shape == ref.shape
    assert rel(y, ref) < (4e-3 if dt == "bf16" else 1e-5), rel(y, ref)


@pytest.mark.parametrize("tile", [14, 15, 16, 17, 18, 25, 26, 27, 28, 29, 30, 31])
@pytest.mark.parametrize("M,N,K,splitk", [(512, 320, 256, 1), (1000, 640, 384, 1), (256, 1280, 2048, 4), (300, 68, 192, 1),
                                          (4096, 960, 384, 1), (130, 320, 64, 1)])
def test_linear_buffer_addressed_kernel(tile, M, N, K, splitk):
    """The buffer-addressed bf16 kernel (descriptor + per-lane offset + scalar K walk): every tile shape, ragged M/N, split-K, and
    K as short as one K tile."""
    from mrisr import ops
    x, w, b = _rnd((M, K), "bf16", 31), _rnd((N, K), "f32", 32, K ** -0.5), _rnd((N,), "f32", 33)
    ref = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile, splitk=splitk)
    assert rel(y, ref) < TOL["bf16"]


@pytest.mark.parametrize("tile", [14, 15, 16, 17, 18, 25, 26, 27, 28, 29, 30, 31])
@pytest.mark.parametrize("B,Cin,Cout,H,stride,ups,splitk", [(2, 64, 320, 16, 1, False, 1), (1, 320, 320, 32, 1, False, 1),
                                                            (2, 128, 64, 8, 1, True, 1), (2, 64, 128, 16, 2, False, 1),
                                                            (3, 256, 256, 4, 1, False, 4), (1, 64, 64, 5, 1, False, 1)])
def test_conv3x3_buffer_addressed_kernel(tile, B, Cin, Cout, H, stride, ups, splitk):
    from mrisr import ops
    x = _rnd((B, Cin, H, H), "bf16", 34)
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 35, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 36)
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xi, w.to(torch.bfloat16).float(), b, stride=stride, padding=1)
    y = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=ups, tile=tile, splitk=splitk)
    assert rel(y, ref) < TOL["bf16"]


@pytest.mark.parametrize("tile,B,C1,C2,Cout,H,splitk", [
    (41, 2, 64, 0, 320, 32, 1), (41, 1, 320, 0, 320, 16, 1), (41, 2, 128, 64, 160, 16, 1), (41, 1, 256, 0, 64, 32, 2),
    (42, 2, 64, 0, 128, 32, 1), (42, 1, 128, 128, 320, 16, 2), (43, 3, 64, 0, 320, 16, 1), (43, 1, 128, 0, 100, 32, 1),
    (44, 1, 64, 64, 384, 32, 1), (44, 2, 192, 0, 640, 16, 1), (45, 1, 64, 0, 64, 64, 1), (45, 2, 128, 0, 320, 16, 1),
    (41, 1, 64, 0, 160, 64, 1), (43, 4, 128, 0, 320, 8, 1), (43, 2, 64, 64, 96, 8, 2),
])
def test_conv3x3_halo_kernel(tile, B, C1, C2, Cout, H, splitk):
    """The LDS-halo conv kernel (the input neighbourhood staged once per 64-channel chunk, nine taps read shifted
    fragments from it): every tile shape, image widths 16/32/64, skip-concat source, split over channel chunks,
    ragged Cout, image borders (zero padding through out-of-range DMA) and tiles that start mid-image."""
    from mrisr import ops
    x1 = _rnd((B, C1, H, H), "bf16", 51)
    x2 = _rnd((B, C2, H, H), "bf16", 52) if C2 else None
    Cin = C1 + C2
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 53, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 54)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    ref = F.conv2d(xin, w.to(torch.bfloat16).float(), b, padding=1)
    y = ops.conv3x3(x1.cuda(), w.cuda(), b.cuda(), x2=x2.cuda() if x2 is not None else None, tile=tile, splitk=splitk)
    assert rel(y, ref) < TOL["bf16"]


@pytest.mark.parametrize("tile,B,C1,C2,Cout,H,splitk", [
    (46, 2, 64, 0, 320, 32, 1), (47, 1, 320, 0, 320, 16, 1), (48, 2, 128, 64, 160, 16, 1), (48, 1, 256, 0, 64, 32, 2),
    (47, 1, 64, 64, 384, 32, 1), (48, 1, 64, 0, 160, 64, 1), (47, 1, 128, 0, 100, 32, 1), (46, 8, 320, 0, 320, 16, 2),
    (48, 8, 640, 320, 320, 32, 1), (47, 8, 320, 320, 320, 32, 1), (48, 16, 64, 0, 320, 32, 1),
])
def test_conv3x3_halo_kernel_256_rows(tile, B, C1, C2, Cout, H, splitk):
    """The eight-wave halo kernel (256-row tiles, one weight tile per phase for all of them, a ring of 2 / 3 / 4 weight stages
    filled ahead with counted waits): widths 16 / 32 / 64, skip-concat source, split over channel chunks, ragged Cout (clamped
    weight rows), one phase sequence longer than the ring and shorter tails, many more tiles than CUs; the same bits twice."""
    from mrisr import ops
    x1 = _rnd((B, C1, H, H), "bf16", 55)
    x2 = _rnd((B, C2, H, H), "bf16", 56) if C2 else None
    Cin = C1 + C2
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 57, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 58)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    ref = F.conv2d(xin, w.to(torch.bfloat16).float(), b, padding=1)
    x1c, x2c, wc, bc = x1.cuda(), (x2.cuda() if x2 is not None else None), w.cuda(), b.cuda()
    y = ops.conv3x3(x1c, wc, bc, x2=x2c, tile=tile, splitk=splitk)
    assert rel(y, ref) < TOL["bf16"], rel(y, ref)
    assert float((y.float().cpu() - ref).abs().max()) < 0.05 * float(ref.abs().max())
    for _ in range(3):
        assert torch.equal(y, ops.conv3x3(x1c, wc, bc, x2=x2c, tile=tile, splitk=splitk))


@pytest.mark.parametrize("tile,M,N,K", [(50, 256, 320, 320), (50, 4096, 960, 320), (50, 64, 160, 320), (51, 320, 640, 320), (51, 1024, 128, 320),
                                        (52, 192, 640, 640), (52, 2048, 64, 640)])
def test_linear_weight_stationary_kernel(tile, M, N, K):
    """The weight-stationary short-K kernel (slab of the weight resident in LDS, activation fragments straight from global
    memory one row block ahead): bias, residual in place, GEGLU, several row blocks per workgroup."""
    from mrisr import _lib as L
    from mrisr import ops
    x, w, b = _rnd((M, K), "bf16", 61), _rnd((N, K), "f32", 62, K ** -0.5), _rnd((N,), "f32", 63)
    ref = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
    assert rel(ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile), ref) < TOL["bf16"]
    if N % 32 == 0:
        u, g = ref.chunk(2, dim=-1)
        assert rel(ops.linear(x.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU, tile=tile), u * F.gelu(g)) < TOL["bf16"]


@pytest.mark.parametrize("tile,M,N,K", [(60, 128, 320, 320), (60, 4096, 960, 320), (60, 200, 336, 320), (60, 32768 + 40, 64, 320), (60, 1000, 2560, 320),
                                        (61, 256, 640, 640), (61, 8192 + 16, 1920, 640), (61, 130, 48, 640),
                                        (64, 256, 640, 640), (64, 8192 + 16, 1920, 640), (64, 100, 48, 640), (64, 4096, 5120, 640),
                                        (65, 64, 320, 320), (65, 8192 + 40, 960, 320), (65, 200, 336, 320), (65, 1000, 2560, 320)])
def test_linear_row_panel_kernel(tile, M, N, K):
    """The row-panel kernel (a workgroup's 128 rows resident in registers for the whole K, weights streamed through LDS in
    40-KB chunks of whole rows): ragged M (rows past M read zeros and are not stored), a partial last chunk (N % chunk != 0),
    several chunks per workgroup and several workgroups per panel, bias, GEGLU; and the LayerNorm prologue on the resident rows
    against F.layer_norm + F.linear."""
    from mrisr import _lib as L
    from mrisr import ops
    x, w, b = _rnd((M, K), "bf16", 64), _rnd((N, K), "f32", 65, K ** -0.5), _rnd((N,), "f32", 66)
    wq = w.to(torch.bfloat16).float()
    ref = F.linear(x.float(), wq, b)
    assert rel(ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile), ref) < TOL["bf16"]
    assert rel(ops.linear(x.cuda(), w.cuda(), None, tile=tile), F.linear(x.float(), wq)) < TOL["bf16"]
    if N % 32 == 0 and tile in (60, 65):  # GEGLU pairs (u, gate) fragments: 4 fragments per chunk only
        u, g = ref.chunk(2, dim=-1)
        assert rel(ops.linear(x.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU, tile=tile), u * F.gelu(g)) < TOL["bf16"]
    # LayerNorm prologue: rows with a large common offset (the statistics are two-pass, exact in f32) and a non-trivial affine
    xo = (x.float() * 0.5 + 3.0).to(torch.bfloat16)
    ga, be = 1 + 0.1 * _rnd((K,), "f32", 67), 0.1 * _rnd((K,), "f32", 68)
    xn = F.layer_norm(xo.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()   # the un-fused path rounds LN(x) to bf16 too
    got = ops.ln_linear(xo.cuda(), ga.cuda(), be.cuda(), w.cuda(), b.cuda())
    assert rel(got, F.linear(xn, wq, b)) < TOL["bf16"]
    if N % 32 == 0 and K == 320:
        u, g = F.linear(xn, wq, b).chunk(2, dim=-1)
        assert rel(ops.ln_linear(xo.cuda(), ga.cuda(), be.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU), u * F.gelu(g)) < TOL["bf16"]


@pytest.mark.parametrize("K,tile", [(320, 60), (320, 65), (640, 64), (640, 61)])
@pytest.mark.parametrize("M,N", [(4096, 960), (8192, 320), (16384, 320), (32768, 320), (4096, 2560), (2048 + 64, 1280)])
def test_row_panel_kernel_many_workgroups_repeatable(M, N, K, tile):
    """More workgroups than CUs (two co-resident per CU), one chunk or several per workgroup, with and without the LayerNorm
    prologue: every launch gives the SAME bits and they are right.  (An early version of the kernel passed single runs and
    failed here: fragment 0 of the second workgroup on a CU came out wrong with the prologue on.)"""
    from mrisr import _lib as L
    from mrisr import ops
    x = (_rnd((M, K), "f32", 71) * 0.5 + 3.0).to(torch.bfloat16)
    w, b = _rnd((N, K), "f32", 72, K ** -0.5), _rnd((N,), "f32", 73)
    ga, be = 1 + 0.1 * _rnd((K,), "f32", 74), 0.1 * _rnd((K,), "f32", 75)
    wq = w.to(torch.bfloat16).float()
    xn = F.layer_norm(x.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()
    ref_ln, ref = F.linear(xn, wq, b), F.linear(x.float(), wq, b)
    xc, wc, bc, gc, bec = x.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda()
    first_ln = first = None
    for rep in range(4):
        got_ln = ops.ln_linear(xc, gc, bec, wc, bc)
        got = ops.linear(xc, wc, bc, tile=tile)
        if rep == 0:
            first_ln, first = got_ln.clone(), got.clone()
            assert rel(got_ln, ref_ln) < TOL["bf16"] and rel(got, ref) < TOL["bf16"], (rel(got_ln, ref_ln), rel(got, ref))
            # element-wise too: a few wrong rows would hide in the norm
            assert float((got_ln.float().cpu() - ref_ln).abs().max()) < 0.05 * float(ref_ln.abs().max())
        else:
            assert torch.equal(got_ln, first_ln) and torch.equal(got, first), rep
    if N % 32 == 0 and K == 320:
        u, gg = ref_ln.chunk(2, dim=-1)
        a = ops.ln_linear(xc, gc, bec, wc, bc, act=L.ACT_GEGLU)
        assert rel(a, u * F.gelu(gg)) < TOL["bf16"] and torch.equal(a, ops.ln_linear(xc, gc, bec, wc, bc, act=L.ACT_GEGLU))


@pytest.mark.parametrize("kind,shape,tile,splitk", [
    ("lin", (2048, 1280, 5120), 26, 2), ("lin", (512, 1280, 2560), 17, 4), ("lin", (2048 - 40, 1280 - 32, 5120), 25, 4),
    ("conv", (32, 1280, 1280, 4), 26, 8), ("conv", (8, 640, 640, 16), 43, 2), ("conv", (2, 1280, 640, 32), 41, 4)])
def test_splitk_reduced_inside_the_gemm_kernel(kind, shape, tile, splitk):
    """Split-K without the reduce launch: the last split of a tile to arrive sums the f32 slabs in split order and runs the kernel's
    own epilogue.  Same bits as the separate reduce kernel (same order of the same f32 additions), the same bits on every repeat
    whichever split arrives last (30 launches, many more tiles x splits than CUs), the counters back at zero (the second and later
    launches would otherwise never reduce), ragged edges, and right against the f32 reference."""
    from mrisr import _lib as L
    from mrisr import ops
    lib = L.lib()
    if kind == "lin":
        M, N, K = shape
        x, w, b = _rnd((M, K), "bf16", 91), _rnd((N, K), "f32", 92, K ** -0.5), _rnd((N,), "f32", 93)
        ref = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
        xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
        run = lambda: ops.linear(xc, wc, bc, tile=tile, splitk=splitk)
    else:
        B, Cin, Cout, H = shape
        x, w, b = _rnd((B, Cin, H, H), "bf16", 94), _rnd((Cout, Cin, 3, 3), "f32", 95, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 96)
        ref = F.conv2d(x.float(), w.to(torch.bfloat16).float(), b, padding=1)
        xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
        run = lambda: ops.conv3x3(xc, wc, bc, tile=tile, splitk=splitk)
    try:
        lib.mrisr_debug_sk_inkernel(0)
        two_pass = run()
        lib.mrisr_debug_sk_inkernel(1)
        first = run()
        assert rel(first, ref) < TOL["bf16"], rel(first, ref)
        assert torch.equal(first, two_pass), float((first.float() - two_pass.float()).abs().max())
        for rep in range(30):
            assert torch.equal(run(), first), rep
    finally:
        lib.mrisr_debug_sk_inkernel(-1)


@pytest.mark.parametrize("M,H,bias,residual", [(128, 1280, True, True), (33000, 1280, True, True), (1000, 64, False, False),
                                               (4096 + 77, 320, True, False)])
def test_fused_feed_forward_kernel(M, H, bias, residual):
    """norm3 -> ff.net.0.proj (GEGLU) -> ff.net.2 -> + hidden in one kernel (C = 320; BasicTransformerBlock's feed-forward,
    diffusers attention.py): against the f32 chain with the kernel's own rounding points (bf16 LayerNorm output, bf16 hidden
    activation), ragged last panel, more panels than CUs, repeatable bits; and against the two-kernel path it replaces."""
    from mrisr import _lib as L
    from mrisr import ops
    C = 320
    x = (_rnd((M, C), "f32", 81) * 0.7 + 0.5).to(torch.bfloat16)
    w1, b1 = _rnd((2 * H, C), "f32", 82, C ** -0.5), (_rnd((2 * H,), "f32", 83) if bias else None)
    w2, b2 = _rnd((C, H), "f32", 84, H ** -0.5), (_rnd((C,), "f32", 85) if bias else None)
    ga, be = 1 + 0.1 * _rnd((C,), "f32", 86), 0.1 * _rnd((C,), "f32", 87)
    q = lambda t: t.to(torch.bfloat16).float()
    xn = q(F.layer_norm(x.float(), (C,), ga, be, 1e-5))
    u, g = F.linear(xn, q(w1), b1).chunk(2, dim=-1)
    ref = F.linear(q(u * F.gelu(g)), q(w2), b2) + (x.float() if residual else 0)
    c = lambda t: t.cuda() if t is not None else None
    xc = x.cuda()
    got = ops.mlp(xc, c(ga), c(be), c(w1), c(b1), c(w2), c(b2), residual=residual)
    assert rel(got, ref) < TOL["bf16"], rel(got, ref)
    assert float((got.float().cpu() - ref).abs().max()) < 0.05 * float(ref.abs().max())
    for _ in range(3):
        assert torch.equal(got, ops.mlp(xc, c(ga), c(be), c(w1), c(b1), c(w2), c(b2), residual=residual))
    # the path it replaces: LayerNorm-prologue GEGLU projection, then the second projection (+ residual in f32 here)
    hid = ops.ln_linear(xc, c(ga), c(be), c(w1), c(b1), act=L.ACT_GEGLU)
    two = ops.linear(hid, c(w2), c(b2)).float() + (xc.float() if residual else 0)
    assert rel(got, two) < TOL["bf16"], rel(got, two)


@pytest.mark.parametrize("tile", [32, 33, 34, 35, 36, 37, 38, 39])
@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (512, 1280, 2560), (256, 384, 192), (128, 128, 320)])
def test_linear_counted_ring_kernel(tile, M, N, K):
    """The 3- / 4-stage variants of the buffer-addressed kernel (counted vmcnt ring, NSTAGE - 1 K tiles in flight): K shorter
    than, equal to and much longer than the ring, bias, GEGLU, repeatable bits."""
    from mrisr import _lib as L
    from mrisr import ops
    bm, bn = {32: (64, 64), 33: (64, 64), 34: (64, 128), 35: (128, 64), 36: (128, 128), 37: (128, 160), 38: (128, 128), 39: (64, 160)}[tile]
    if bn == 160:  # the 160-wide tiles (5 fragments per wave along N) tile N = 1280 / 640 / 320, not the 128-multiples of the other cases
        N = {1280: 1280, 384: 320, 128: 160}[N]
    x, w, b = _rnd((M, K), "bf16", 91), _rnd((N, K), "f32", 92, K ** -0.5), _rnd((N,), "f32", 93)
    ref = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile)
    assert rel(y, ref) < TOL["bf16"]
    for _ in range(3):  # (raw-barrier ring: a race would show as differing bits)
        assert torch.equal(y, ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile))
    if bn != 160:  # (no (u, gate) pairing in a 5-fragment wave tile)
        u, g = ref.chunk(2, dim=-1)
        assert rel(ops.linear(x.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU, tile=tile), u * F.gelu(g)) < TOL["bf16"]


def _fq(t):
    """e4m3 fake-quant with one scale per row (the scheme of the fp8 projections; oracle.unet.fp8_fake_quant_rows)."""
    sc = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-20) / 448.0
    return (t / sc).to(torch.float8_e4m3fn).float() * sc


@pytest.mark.parametrize("M,N,K", [(128, 320, 320), (4096, 960, 320), (200, 336, 320), (8192 + 40, 320, 320), (1000, 2560, 320),
                                   (256, 640, 640), (4096 + 16, 1920, 640), (130, 48, 640)])
def test_linear_fp8_row_panel_kernel(M, N, K):
    """BASELINE configs[4]: the fp8 (OCP e4m3, v_mfma_f32_16x16x32_fp8_fp8) form of the row-panel kernel.  Against the SAME
    quantisation done in torch (per-row activation scales, per-channel weight scales, exact accumulation) the kernel must be as
    close as the bf16 kernel is to its reference - that pins the operand layout, the scales and the epilogue; against the
    unquantised product the error is the fp8 quantisation error itself (3 mantissa bits: a few percent)."""
    from mrisr import _lib as L
    from mrisr import ops
    x, w, b = _rnd((M, K), "bf16", 81), _rnd((N, K), "f32", 82, K ** -0.5), _rnd((N,), "f32", 83)
    wq = w.to(torch.bfloat16).float()
    exact = F.linear(x.float(), wq, b)
    ref8 = F.linear(_fq(x.float()), _fq(wq), b)
    got = ops.linear_fp8(x.cuda(), w.cuda(), b.cuda())
    assert rel(got, ref8) < TOL["bf16"], rel(got, ref8)
    assert 5e-3 < rel(got, exact) < 8e-2, rel(got, exact)          # fp8 really ran, and its error is the expected size
    assert rel(ops.linear_fp8(x.cuda(), w.cuda(), None), F.linear(_fq(x.float()), _fq(wq))) < TOL["bf16"]
    if N % 32 == 0:
        u, g = ref8.chunk(2, dim=-1)
        assert rel(ops.linear_fp8(x.cuda(), w.cuda(), b.cuda(), act=L.ACT_GEGLU), u * F.gelu(g)) < TOL["bf16"]
    # LayerNorm prologue in front of the quantisation
    xo = (x.float() * 0.5 + 3.0).to(torch.bfloat16)
    ga, be = 1 + 0.1 * _rnd((K,), "f32", 84), 0.1 * _rnd((K,), "f32", 85)
    xn = F.layer_norm(xo.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()
    got = ops.linear_fp8(xo.cuda(), w.cuda(), b.cuda(), gamma=ga.cuda(), beta=be.cuda())
    assert rel(got, F.linear(_fq(xn), _fq(wq), b)) < TOL["bf16"]
    assert torch.equal(got, ops.linear_fp8(xo.cuda(), w.cuda(), b.cuda(), gamma=ga.cuda(), beta=be.cuda()))


def test_unet_fp8_projections_match_fake_quant_oracle():
    """SD-1.5 channel widths (320 / 640, the widths the fp8 projections serve) + rank-4 LoRA: the engine with fp8_linears against
    the oracle with the same linears fake-quantised (oracle.unet.FP8_LINEARS), within the bf16 engine's own bound; and the fp8
    kernels really ran (profiler classes)."""
    import json

    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = ou.UNetConfig(block_out_channels=(320, 640), attn_levels=(True, True), cross_attention_dim=64)
    up = ou.init_unet_params(cfg, seed=171, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=172)
    p = {**up, **lora}
    g = torch.Generator().manual_seed(173)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, 64), generator=g)
    t = torch.tensor([40, 700])
    ref = ou.unet_forward(p, cfg, x, t, ctx)
    served = ("proj_in", "proj_out", "to_q", "to_k", "to_v", "to_out.0", "ff.net.0.proj")
    try:
        # the engine's fp8 set: K in {320, 640} linears of the transformer blocks (attn2.to_k / to_v have K = the context width);
        # the K = 320 feed-forward runs in the fused bf16 kernel (LayerNorm -> FF1 -> GEGLU -> FF2 in one launch), not in fp8
        ou.FP8_LINEARS = lambda name, K: K in (320, 640) and name.endswith(served) and not (K == 320 and name.endswith("ff.net.0.proj"))
        ref8 = ou.unet_forward(p, cfg, x, t, ctx)
    finally:
        ou.FP8_LINEARS = None
    lib = L.lib()
    net8 = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, fp8="all")
    net8.load_state_dict(p)
    net8(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda())
    lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
    out8 = net8(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
    buf = C.create_string_buffer(1 << 20)
    n = lib.mrisr_prof_report(buf, len(buf))
    classes = json.loads(buf.value[:n].decode())
    lib.mrisr_prof_reset()
    assert any(k.startswith("gemm_fp8_rp320") for k in classes) and any(k.startswith("gemm_fp8_rp640") for k in classes), sorted(classes)
    net16 = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
    net16.load_state_dict(p)
    out16 = net16(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    e8, e16, e8x = rel(out8, ref8), rel(out16, ref), rel(out8, ref)
    print(f"fp8 engine vs fake-quant oracle {e8:.4e}; bf16 engine vs oracle {e16:.4e}; fp8 engine vs unquantised oracle {e8x:.4e}")
    # e4m3 carries 3 mantissa bits: ~5 % through the whole network; the two fp8 computations round different bf16 / f32 inputs, so
    # their quantisation noise is only partly shared - the bound is the fp8 noise level, the layout / scale check is the op test above
    assert e16 < 5e-2 and e8 < 8e-2 and e8x < 1.0e-1
    with pytest.raises(ValueError):
        mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", fp8=True)


def test_unet_with_specialised_kernels_preferred():
    """SD-1.5 channel widths at a tiny spatial size, bf16 + explicit LoRA: the same forward with the autotuner's choice and
    with the weight-stationary / halo kernels preferred wherever they are eligible (incl. the in-kernel LoRA
    down-projection inside the weight-stationary kernel) - all within the bf16 bound of the CPU oracle."""
    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = ou.UNetConfig(block_out_channels=(320, 640), attn_levels=(True, True), cross_attention_dim=64)
    up = ou.init_unet_params(cfg, seed=71, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=72)
    g = torch.Generator().manual_seed(73)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, 64), generator=g)
    t = torch.tensor([40, 700])
    ref = ou.unet_forward({**up, **lora}, cfg, x, t, ctx)
    lib = L.lib()
    try:
        for pref in (0, 50, 52, 41, 43, 60, 61, 64, 65):  # (ring tiles 32-36: test_unet_counted_ring_tiles_forced below)
            lib.mrisr_debug_prefer_tile(C.c_int(pref))
            net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
            net.load_state_dict({**up, **lora})
            out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
            assert rel(out, ref) < 5e-2, (pref, rel(out, ref))
    finally:
        lib.mrisr_debug_prefer_tile(C.c_int(0))


@pytest.mark.parametrize("tile", [14, 15, 16, 17, 18, 28, 29, 30])
def test_buffer_addressed_kernel_concat_and_geglu(tile):
    from mrisr import _lib as L
    from mrisr import ops
    x1, x2 = _rnd((2, 128, 8, 8), "bf16", 37), _rnd((2, 64, 8, 8), "bf16", 38)
    w, b = _rnd((128, 192, 3, 3), "f32", 39, (9 * 192) ** -0.5), _rnd((128,), "f32", 40)
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), w.to(torch.bfloat16).float(), b, padding=1)
    assert rel(ops.conv3x3(x1.cuda(), w.cuda(), b.cuda(), x2=x2.cuda(), tile=tile), ref) < TOL["bf16"]
    M, Cc = 384, 128
    x, wg, bg = _rnd((M, Cc), "bf16", 41), _rnd((8 * Cc, Cc), "f32", 42, Cc ** -0.5), _rnd((8 * Cc,), "f32", 43)
    h = F.linear(x.float(), wg.to(torch.bfloat16).float(), bg)
    u, g = h.chunk(2, dim=-1)
    assert rel(ops.linear(x.cuda(), wg.cuda(), bg.cuda(), act=L.ACT_GEGLU, tile=tile), u * F.gelu(g)) < TOL["bf16"]


def test_gemm_kernels_with_poisoned_lds_give_the_same_bits():
    """Ordering check for every LDS-DMA-fed GEMM kernel (VERDICT r2 "what's weak" 4): with `mrisr_debug_gemm_flags(2048)` a
    workgroup fills its whole LDS allocation with 0xFF bytes (NaN in bf16 / f32 / e4m3) before its first DMA.  Without that a
    freshly scheduled workgroup inherits the previous workgroup's LDS image - the same weights at the same offsets more often
    than not - so a `ds_read` that is not ordered behind its DMA (issuing wave's covering vmcnt + a barrier the reader passed)
    reads plausible stale bytes and passes reference checks AND repeatability screens.  Poisoned, it reads NaN.  Every variant
    below must produce exactly the bits of the un-poisoned run: row-panel (K = 320 / 640, 4- and 2-wave, LayerNorm prologue,
    GEGLU, several chunks per workgroup, more workgroups than CUs, ragged last panel), fused feed-forward, tiled GEMM incl. a
    ragged tile and split-K, halo conv."""
    import ctypes as C
    from mrisr import _lib as L
    from mrisr import ops
    lib = L.lib()
    cases = []
    for (M, N, K, tile) in [(32768, 320, 320, 60), (4096, 2560, 320, 60), (2048 + 64, 1280, 320, 65), (8192, 640, 640, 64),
                            (4096 + 32, 1920, 640, 61)]:
        x = (_rnd((M, K), "f32", 171) * 0.5 + 3.0).to(torch.bfloat16).cuda()
        w, b = _rnd((N, K), "f32", 172, K ** -0.5).cuda(), _rnd((N,), "f32", 173).cuda()
        ga, be = (1 + 0.1 * _rnd((K,), "f32", 174)).cuda(), (0.1 * _rnd((K,), "f32", 175)).cuda()
        cases.append((f"rp {M}x{N}x{K} t{tile}", lambda x=x, w=w, b=b, tile=tile: ops.linear(x, w, b, tile=tile)))
        cases.append((f"rp+ln {M}x{N}x{K}", lambda x=x, w=w, b=b, ga=ga, be=be: ops.ln_linear(x, ga, be, w, b)))
        if K == 320 and N % 32 == 0:
            cases.append((f"rp+ln+geglu {M}x{N}", lambda x=x, w=w, b=b, ga=ga, be=be: ops.ln_linear(x, ga, be, w, b, act=L.ACT_GEGLU)))
    xm = (_rnd((4096 + 96, 320), "f32", 181) * 0.7 + 0.5).to(torch.bfloat16).cuda()
    w1, b1 = _rnd((2560, 320), "f32", 182, 320 ** -0.5).cuda(), _rnd((2560,), "f32", 183).cuda()
    w2, b2 = _rnd((320, 1280), "f32", 184, 1280 ** -0.5).cuda(), _rnd((320,), "f32", 185).cuda()
    gm, bm = (1 + 0.1 * _rnd((320,), "f32", 186)).cuda(), (0.1 * _rnd((320,), "f32", 187)).cuda()
    cases.append(("mlp fused", lambda: ops.mlp(xm, gm, bm, w1, b1, w2, b2, residual=True)))
    xl = _rnd((2048 - 40, 1280), "bf16", 191).cuda()
    wl, bl_ = _rnd((1280 - 32, 1280), "f32", 192, 1280 ** -0.5).cuda(), _rnd((1280 - 32,), "f32", 193).cuda()
    for tile, sk in [(25, 1), (26, 2), (17, 1), (14, 1)]:
        cases.append((f"tiled t{tile} sk{sk}", lambda tile=tile, sk=sk: ops.linear(xl, wl, bl_, tile=tile, splitk=sk)))
    xc = _rnd((8, 640, 16, 16), "bf16", 194).cuda()
    wc, bc = _rnd((640, 640, 3, 3), "f32", 195, (9 * 640) ** -0.5).cuda(), _rnd((640,), "f32", 196).cuda()
    for tile in (41, 43, 26):
        cases.append((f"conv t{tile}", lambda tile=tile: ops.conv3x3(xc, wc, bc, tile=tile)))
    try:
        for name, run in cases:
            lib.mrisr_debug_gemm_flags(C.c_int(0))
            clean = run().clone()
            assert bool(torch.isfinite(clean.float()).all()), name
            lib.mrisr_debug_gemm_flags(C.c_int(2048))
            for rep in range(3):
                got = run()
                assert bool(torch.isfinite(got.float()).all()), (name, rep, "NaN: a ds_read ran ahead of its LDS-DMA")
                assert torch.equal(got, clean), (name, rep)
    finally:
        lib.mrisr_debug_gemm_flags(C.c_int(0))


@pytest.mark.parametrize("B,N,Nk,C,H", [(2, 256, 256, 320, 8), (1, 1024, 1024, 320, 8), (2, 64, 64, 1280, 8), (2, 16, 16, 1280, 8),
                                        (2, 256, 77, 640, 8), (1, 200, 77, 64, 8), (1, 64, 64, 256, 8), (1, 100, 130, 1024, 8)])
def test_attention_fp8(B, N, Nk, C, H):
    """BASELINE configs[4] "fp8 attention": Q K^T and P V on v_mfma_f32_16x16x32_fp8_fp8 (OCP e4m3), f32 softmax, per-head scales
    chosen so that the matrix core's output IS the log2-domain score (csrc/attn.hip).  Against the same quantisation rule done in
    torch (oracle.unet.sdpa_fp8_fake_quant) the kernel is as close as the bf16 kernel is to exact SDPA - that pins operand layouts,
    scales, the row-sum row and the normalisation; against exact SDPA the distance is the e4m3 noise (3 mantissa bits).  Every
    padded head dim of the SD-1.5 levels, ragged query / key counts, the 77-key cross-attention shape."""
    from mrisr import ops
    from oracle import unet as ou
    q, k, v = _rnd((B, N, C), "bf16", 121), _rnd((B, Nk, C), "bf16", 122), _rnd((B, Nk, C), "bf16", 123)
    hm = lambda t: t.float().view(t.shape[0], t.shape[1], H, C // H).transpose(1, 2)
    ref8 = ou.sdpa_fp8_fake_quant(hm(q), hm(k), hm(v)).transpose(1, 2).reshape(B, N, C)
    exact = _sdpa(q, k, v, H)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, fp8=True)
    assert bool(torch.isfinite(y.float()).all())
    e8, ex = rel(y, ref8), rel(y, exact)
    # the kernel rounds p = 2^(s - m) against its lazily moved reference m, the torch rule against the true row maximum: the two p
    # differ by a factor that is not a power of two, so their e4m3 roundings are independent - the distance between the two fp8
    # results is the P quantisation noise (~3 %), not a layout check; the layout / scale checks proper are the structured cases below
    assert e8 < 5e-2, (e8, ex)
    assert 5e-3 < ex < 1.2e-1, ex   # fp8 really ran, and its error is the expected size
    assert torch.equal(y, ops.attention(q.cuda(), k.cuda(), v.cuda(), H, fp8=True))
    # (1) uniform attention (q = 0: every p is exactly 1): the output is the mean of the QUANTISED V - pins the V^T fragment layout,
    #     sV, the row-sum row and the normalisation to bf16 rounding
    z = torch.zeros_like(q)
    y0 = ops.attention(z.cuda(), k.cuda(), v.cuda(), H, fp8=True)
    r0 = ou.sdpa_fp8_fake_quant(hm(z), hm(k), hm(v)).transpose(1, 2).reshape(B, N, C)
    assert rel(y0, r0) < 1e-2, rel(y0, r0)
    # (2) one-hot attention (query i strongly aligned with key perm[i]: p is 1 for that key and underflows for the rest): the output
    #     is row perm[i] of the quantised V - pins the Q / K fragment layouts, the score scale and the key order
    perm = torch.randperm(Nk, generator=torch.Generator().manual_seed(5))[torch.arange(N) % Nk]
    kk = torch.sign(_rnd((B, Nk, C), "f32", 127)).to(torch.bfloat16)  # +-1 keys: distinct directions, |k|^2 = d
    qq = (kk[:, perm].float() * 6.0).to(torch.bfloat16)               # score of the aligned key: 6 sqrt(d) >> the others
    y1 = ops.attention(qq.cuda(), kk.cuda(), v.cuda(), H, fp8=True)
    r1 = ou.sdpa_fp8_fake_quant(hm(qq), hm(kk), hm(v)).transpose(1, 2).reshape(B, N, C)
    assert rel(y1, r1) < 1e-2, rel(y1, r1)


def test_attention_fp8_online_softmax_rescale_branch():
    """The lazily moved reference of the fp8 kernel (guide rule 26): one key aligned with every query sits in the LAST tile."""
    from mrisr import ops
    from oracle import unet as ou
    B, N, C, H = 1, 256, 320, 8
    q, k, v = _rnd((B, N, C), "bf16", 124), _rnd((B, N, C), "bf16", 125), _rnd((B, N, C), "bf16", 126)
    k[:, 250] = q.float().mean(1).to(torch.bfloat16) * 6.0
    hm = lambda t: t.float().view(B, N, H, C // H).transpose(1, 2)
    ref8 = ou.sdpa_fp8_fake_quant(hm(q), hm(k), hm(v)).transpose(1, 2).reshape(B, N, C)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, fp8=True)
    assert rel(y, ref8) < 5e-2, rel(y, ref8)
    assert rel(y, _sdpa(q, k, v, H)) < 1.2e-1
    # zero operands (amax = 0 in the scale rule) stay finite: uniform attention over V
    z = torch.zeros((1, 64, 320), dtype=torch.bfloat16)
    y0 = ops.attention(z.cuda(), z.cuda(), v[:, :64].cuda(), H, fp8=True)
    assert rel(y0, v[:, :64].float().mean(1, keepdim=True).expand(-1, 64, -1)) < 5e-2


def test_unet_fp8_attention_and_fp8_training_forward():
    """configs[4] composed: fp8 projections + fp8 attention in the model - inference against the fake-quant oracle; and the
    mixed-precision TRAINING step (`fp8_train`): forward through the fp8 kernels, backward in bf16 straight through the quantisers.
    Checked: the fp8 kernels really ran (profiler classes) in both; the fp8 loss is the fake-quant oracle's loss; the LoRA gradient
    bucket stays within the fp8 noise of exact autograd; with fp8_train off the training forward is the bf16 one bit for bit."""
    import ctypes as C
    import json

    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = ou.UNetConfig(block_out_channels=(320, 640), attn_levels=(True, True), cross_attention_dim=64)
    up = ou.init_unet_params(cfg, seed=271, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=272)
    p = {**up, **lora}
    g = torch.Generator().manual_seed(273)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, 64), generator=g)
    tgt = torch.randn((2, 4, 16, 16), generator=g)
    t = torch.tensor([40, 700])
    served = ("proj_in", "proj_out", "to_q", "to_k", "to_v", "to_out.0", "ff.net.0.proj")
    lp = {k_: v_.clone().requires_grad_(True) for k_, v_ in lora.items()}
    with torch.enable_grad():
        pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx)
        loss_exact = torch.nn.functional.mse_loss(pred, tgt)
        loss_exact.backward()
    try:
        ou.FP8_ATTENTION = True
        ref8a = ou.unet_forward(p, cfg, x, t, ctx)          # fp8 attention only
        # the training forward keeps LayerNorm / GEGLU as their own kernels, so ALL served K = 320 / 640 linears run in fp8 there
        ou.FP8_LINEARS = lambda name, K: K in (320, 640) and name.endswith(served)
        ref8_train = ou.unet_forward(p, cfg, x, t, ctx)
    finally:
        ou.FP8_LINEARS = None
        ou.FP8_ATTENTION = False
    lib = L.lib()

    def classes_of(fn):
        lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
        out = fn()
        torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
        buf = C.create_string_buffer(1 << 20)
        n = lib.mrisr_prof_report(buf, len(buf))
        lib.mrisr_prof_reset()
        return out, json.loads(buf.value[:n].decode())

    xa, ta, ca, ga = x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda()
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, fp8_attention=True)
    net.load_state_dict(p)
    net(xa, ta, encoder_hidden_states=ca)
    out, cls = classes_of(lambda: net(xa, ta, encoder_hidden_states=ca).sample)
    assert "flash_attention_fp8" in cls and "attention_quant_fp8" in cls and "flash_attention" not in cls, sorted(cls)
    e = rel(out, ref8a)
    print(f"fp8-attention engine vs fake-quant oracle {e:.4e}; vs exact oracle {rel(out, pred.detach()):.4e}")
    assert e < 5e-2 and rel(out, pred.detach()) < 8e-2
    # ---- mixed-precision training ----
    net8 = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, lora_fused=True, fp8="all", fp8_attention=True,
                                      fp8_train=True)
    net8.load_state_dict(p)
    tr8 = mrisr.LoRATrainer(net8)
    tr8.forward_backward(xa, ta, ca, ga)  # plan + tune
    tr8.zero_grad()
    (loss8, pred8), cls = classes_of(lambda: tr8.forward_backward(xa, ta, ca, ga, return_pred=True))
    assert "flash_attention_fp8" in cls and any(k_.startswith("gemm_fp8_rp") for k_ in cls), sorted(cls)
    assert any(k_.startswith("flash_attention_bwd") for k_ in cls), sorted(cls)   # bf16 flash backward from the fp8 forward's log-sum-exp
    loss8_ref = float(torch.nn.functional.mse_loss(ref8_train, tgt))
    flat_ref = torch.cat([lp[k_].grad.reshape(-1) for k_, _, _ in tr8.layout])
    e_pred, e_grad = rel(pred8, ref8_train), rel(tr8.grad, flat_ref)
    print(f"fp8 training forward vs fake-quant oracle {e_pred:.4e}; loss {float(loss8):.5f} vs {loss8_ref:.5f} (exact {float(loss_exact):.5f}); "
          f"LoRA gradient bucket vs exact autograd {e_grad:.4e}")
    assert e_pred < 8e-2 and abs(float(loss8) - loss8_ref) / loss8_ref < 3e-2
    assert e_grad < 2.5e-1, e_grad
    # fp8_train off: the same model flags train through the bf16 forward, bit for bit the plain bf16 model's step
    netb = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, lora_fused=True, fp8="all", fp8_attention=True)
    netb.load_state_dict(p)
    net16 = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, lora_fused=True)
    net16.load_state_dict(p)
    trb, tr16 = mrisr.LoRATrainer(netb), mrisr.LoRATrainer(net16)
    lb, pb = trb.forward_backward(xa, ta, ca, ga, return_pred=True)
    l16, p16 = tr16.forward_backward(xa, ta, ca, ga, return_pred=True)
    print(f"fp8_train off vs plain bf16: pred equal {torch.equal(pb, p16)}, grad rel {rel(trb.grad, tr16.grad):.3e}")
    assert torch.equal(pb, p16) and rel(trb.grad, tr16.grad) < 1e-6
    assert rel(tr16.grad, flat_ref) < 6e-2
    with pytest.raises(ValueError):
        mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", fp8_train=True)
    with pytest.raises(ValueError):
        mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", fp8_attention=True)


@pytest.mark.parametrize("B,Cin,Cout,H", [(32, 640, 640, 16), (8, 1280, 1280, 8), (2, 64, 128, 8), (1, 320, 64, 12), (3, 128, 64, 5)])
def test_upsample_conv_subpixel_form(B, Cin, Cout, H):
    """`nearest x2 -> conv3x3` (diffusers Upsample2D in every decoder level but the last) as four 2 x 2 convs on the LOW-resolution
    input, one per output parity, with pre-summed taps - 4/9 of the MACs, exact in exact arithmetic (borders included: the
    up-sampled image's zero padding is the low-resolution image's).  Against F.interpolate + conv2d on the same bf16-rounded
    operands, and against the literal up-sampled implicit GEMM it replaces; odd sizes, one image, several tile shapes."""
    from mrisr import ops
    x = _rnd((B, Cin, H, H), "bf16", 201)
    w, b = _rnd((Cout, Cin, 3, 3), "f32", 202, (9 * Cin) ** -0.5), _rnd((Cout,), "f32", 203)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.to(torch.bfloat16).float(), b, padding=1)
    lit = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), upsample=True)
    sub = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), upsample=True, subpix=True)
    assert sub.shape == ref.shape
    # (the pre-summed taps are rounded to bf16 once instead of three / nine times: the two forms differ by weight rounding only)
    assert rel(lit, ref) < 4e-3 and rel(sub, ref) < 6e-3, (rel(lit, ref), rel(sub, ref))
    d = (sub.float().cpu() - ref).abs()
    # borders and every parity: no pixel class is off (a wrong window origin shows as O(1) on a quarter of the pixels or on an edge)
    for py in (0, 1):
        for px in (0, 1):
            assert float(d[:, :, py::2, px::2].max()) < 0.05 * float(ref.abs().max()), (py, px)
    for sl in (d[:, :, 0], d[:, :, -1], d[:, :, :, 0], d[:, :, :, -1]):
        assert float(sl.max()) < 0.05 * float(ref.abs().max())
    assert torch.equal(sub, ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), upsample=True, subpix=True))


def test_unet_decoder_uses_the_subpixel_upsampler_and_matches_the_literal_form():
    """Model level: with the sub-pixel form forced on at any size (mrisr_debug_subpix(1)) the bf16 UNet agrees with the literal
    up-sampled convs (mrisr_debug_subpix(0)) to weight-rounding level and with the oracle within the bf16 bound; the f32 engine and
    the training graph never use it."""
    import ctypes as C
    import json
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = ou.UNetConfig(block_out_channels=(64, 128, 256), attn_levels=(True, True, False), cross_attention_dim=64)
    up = ou.init_unet_params(cfg, seed=301, perturb_norm=True)
    g = torch.Generator().manual_seed(302)
    x, ctx = torch.randn((2, 4, 16, 16), generator=g), torch.randn((2, 77, 64), generator=g)
    t = torch.tensor([10, 900])
    with torch.no_grad():
        ref = ou.unet_forward(up, cfg, x, t, ctx)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    net.load_state_dict(up)
    lib = L.lib()
    try:
        lib.mrisr_debug_subpix(C.c_int(0))
        lit = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample.clone()
        lib.mrisr_debug_subpix(C.c_int(1))
        net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda())
        lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
        sub = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample.clone()
        torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
        buf = C.create_string_buffer(1 << 20)
        n = lib.mrisr_prof_report(buf, len(buf))
        cls = json.loads(buf.value[:n].decode())
        lib.mrisr_prof_reset()
    finally:
        lib.mrisr_debug_subpix(C.c_int(-1))
    assert cls.get("subpix_shuffle", {}).get("launches") == 2, sorted(cls)   # two upsamplers in a three-level decoder
    assert rel(sub, ref) < 5e-2 and rel(lit, ref) < 5e-2 and rel(sub, lit) < 2e-2, (rel(sub, ref), rel(lit, ref), rel(sub, lit))
