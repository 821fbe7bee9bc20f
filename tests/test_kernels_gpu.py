"""GPU parity of every HIP kernel behind the C ABI against a plain fp32 PyTorch statement of the same op
on the same bf16-rounded inputs (floating-point kernels: tolerance written per test; the bf16 output
rounding alone is 2^-9 relative)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
F16 = torch.float16


def _ops():
    from stablediffusion_amd import ops
    return ops


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def q(x):
    """Round to bf16 and back: what the kernel actually sees."""
    return x.to(BF16).float()


def rel_err(a, b):
    """max|a-b| / max|b| - the per-tensor relative error of SURVEY.md §8(d)."""
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12)).item()


# ----------------------------------------------------------------------------- MFMA layout probes


def test_gemm_identity_layout(dev):
    """A = I against an ASYMMETRIC W exposes any row/col swap in the fragment or C/D maps (exact in bf16)."""
    ops = _ops()
    n, k = 128, 128
    w = (torch.arange(n, device=dev)[:, None] * 3 + torch.arange(k, device=dev)[None, :] * 5 % 7).float() % 64
    x = torch.eye(k, device=dev)
    pw = ops.pack_linear(w)
    out = ops.linear(x.to(BF16), pw, out_dtype=torch.float32)
    assert torch.equal(out, w.t().contiguous())


@pytest.mark.parametrize("m,n,k", [(1, 320, 320), (77, 1280, 1024), (128, 128, 64), (1000, 320, 1280),
                                   (4096, 640, 640), (130, 4, 320), (300, 200, 192), (50, 64, 96), (33, 24, 8)])
def test_linear_bias(dev, m, n, k):
    ops = _ops()
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    pw = ops.pack_linear(w, b)
    out = ops.linear(x.to(BF16), pw)
    ref = F.linear(q(x), q(w), b)
    assert out.shape == (m, n)
    assert rel_err(out, ref) < 6e-3
    out32 = ops.linear(x.to(BF16), pw, out_dtype=torch.float32)
    assert rel_err(out32, ref) < 2e-5 * math.sqrt(k) + 1e-5


def test_linear_epilogues(dev):
    ops = _ops()
    m, n, k = 512, 640, 320
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    res = rnd((m, n), dev, 4)
    emb = rnd((4, 2 * n), dev, 5)[:, n:]   # strided rows, 4 "batches" of 128 rows
    pw = ops.pack_linear(w, b)
    out = ops.linear(x.to(BF16), pw, res=res.to(BF16), emb=emb, rows_per_batch=128, out_dtype=torch.float32)
    ref = F.linear(q(x), q(w), b) + emb.repeat_interleave(128, dim=0) + q(res)
    assert rel_err(out, ref) < 1e-4
    out = ops.linear(x.to(BF16), pw, act=1, out_dtype=torch.float32)
    assert rel_err(out, F.silu(F.linear(q(x), q(w), b))) < 1e-4
    out = ops.linear(x.to(BF16), pw, alpha=0.25, bias=False, out_dtype=torch.float32)
    assert rel_err(out, 0.25 * F.linear(q(x), q(w))) < 1e-4


def test_linear_concat(dev):
    """Two-pointer A == torch.cat on the channel dim (openai_model.py:1010 feeding a 1x1 skip conv)."""
    ops = _ops()
    m, k1, k2, n = 256, 640, 320, 320
    x1, x2 = rnd((m, k1), dev, 1), rnd((m, k2), dev, 2)
    w, b = rnd((n, k1 + k2), dev, 3, 0.03), rnd((n,), dev, 4)
    out = ops.linear(x1.to(BF16), ops.pack_linear(w, b), x2=x2.to(BF16), out_dtype=torch.float32)
    ref = F.linear(torch.cat([q(x1), q(x2)], -1), q(w), b)
    assert rel_err(out, ref) < 1e-4


def test_geglu(dev):
    ops = _ops()
    m, c = 300, 320
    x = rnd((m, c), dev, 1)
    w, b = rnd((8 * c, c), dev, 2, c ** -0.5), rnd((8 * c,), dev, 3)
    out = ops.linear(x.to(BF16), ops.pack_geglu(w, b), out_dtype=torch.float32)
    a, gate = F.linear(q(x), q(w), b).chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    assert out.shape == (m, 4 * c)
    assert rel_err(out, ref) < 1e-4
    # 16-bit output: the paired 16-byte store form of the epilogue with the packed polynomial erf (|erf error| <= 1e-6
    # instead of erff): a rounding boundary may fall differently - by one ulp, in well under 1 % of the elements (the
    # negative gates, where 1 + erf is small and the absolute error weighs more) - nothing more
    out16 = ops.linear(x.to(BF16), ops.pack_geglu(w, b))
    assert out16.dtype == BF16 and rel_err(out16.float(), out) < 5e-3
    differ = out16 != out.to(BF16)
    assert float(differ.float().mean()) < 1e-2
    ulp = out.abs().clamp_min(1e-30).log2().floor().exp2() * 2.0 ** -7          # bf16: 8 significant bits
    # (one ulp, plus the polynomial's ABSOLUTE GELU error of 2.4e-6 times the value branch where the gated product is tiny)
    assert bool(((out16.float() - out).abs() <= 1.01 * ulp + 1e-5 * a.abs())[differ].all())
    # ... and with an output width that is not a multiple of 32 (falls back to the quad stores)
    w3, b3 = rnd((2 * 80, c), dev, 4, c ** -0.5), rnd((2 * 80,), dev, 5)
    a3, g3 = F.linear(q(x), q(w3), b3).chunk(2, dim=-1)
    o3 = ops.linear(x.to(BF16), ops.pack_geglu(w3, b3))
    assert o3.shape == (m, 80) and rel_err(o3.float(), a3 * F.gelu(g3)) < 1e-2


def test_split_k(dev):
    ops = _ops()
    m, n, k = 128, 1280, 11520
    assert ops._auto_split_k(m, n, k) > 1
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    res = rnd((m, n), dev, 4)
    out = ops.linear(x.to(BF16), ops.pack_linear(w, b), res=res.to(BF16), out_dtype=torch.float32)
    ref = F.linear(q(x), q(w), b) + q(res)
    assert rel_err(out, ref) < 1e-4
    # GEGLU through split-K
    w2, b2 = rnd((2560, 2560), dev, 5, 0.02), rnd((2560,), dev, 6)
    x2 = rnd((64, 2560), dev, 7)
    assert ops._auto_split_k(64, 2560, 2560) > 1
    out = ops.linear(x2.to(BF16), ops.pack_geglu(w2, b2), out_dtype=torch.float32)
    a, gate = F.linear(q(x2), q(w2), b2).chunk(2, dim=-1)
    assert rel_err(out, a * F.gelu(gate)) < 1e-4


@pytest.mark.parametrize("n", [160, 320, 200, 960])
def test_column_tile_variants_agree(dev, n, monkeypatch):
    """The block tilings (128 | 160 columns x 128 | 256 rows, 4- and 3-stage rings) accumulate every output in
    the same k order: bit-identical results for linear and conv (split-K pinned to 1: the heuristic may split the
    tilings differently)."""
    ops = _ops()
    monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: 1)
    m, k = 300, 1280
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    xi = rnd((2, 24, 8, 64), dev, 4).to(BF16)
    wt, bias = rnd((n, 64, 3, 3), dev, 5, 1 / 24), rnd((n,), dev, 6)
    lin, conv = [], []
    for tile_m in (128, 129, 256):
        monkeypatch.setattr(ops, "FORCE_TILE_M", tile_m)
        for t in (128, 160):
            lin.append(ops.linear(x.to(BF16), ops.pack_linear(w, b, tile_n=t), out_dtype=torch.float32))
            conv.append(ops.conv3x3(xi, ops.pack_conv3x3(wt, bias, tile_n=t)))
    for o in lin[1:]:
        assert torch.equal(lin[0], o)
    for o in conv[1:]:
        assert torch.equal(conv[0], o)
    assert rel_err(lin[0], F.linear(q(x), q(w), b)) < 1e-4
    ref = F.conv2d(xi.float().permute(0, 3, 1, 2), q(wt), bias, padding=1)
    assert rel_err(conv[0].permute(0, 3, 1, 2), ref) < 6e-3
    monkeypatch.setattr(ops, "FORCE_TILE_M", 0)
    with pytest.raises(Exception):
        pw = ops.pack_geglu(rnd((640, 64), dev, 7), rnd((640,), dev, 8))
        pw.tile_n = 160
        ops.linear(rnd((8, 64), dev, 9).to(BF16), pw)


def test_tall_tile_full_path(dev, monkeypatch):
    """256-row tile on everything the epilogue does: GEGLU, split-K, concat A, emb + residual, stride-2 and
    upsampling conv, ragged M."""
    ops = _ops()
    monkeypatch.setattr(ops, "FORCE_TILE_M", 256)
    m, c = 700, 320
    x = rnd((m, c), dev, 1)
    w, b = rnd((8 * c, c), dev, 2, c ** -0.5), rnd((8 * c,), dev, 3)
    out = ops.linear(x.to(BF16), ops.pack_geglu(w, b), out_dtype=torch.float32)
    a, gate = F.linear(q(x), q(w), b).chunk(2, dim=-1)
    assert rel_err(out, a * F.gelu(gate)) < 1e-4
    x1, x2 = rnd((m, 640), dev, 4), rnd((m, 320), dev, 5)
    w2, b2 = rnd((320, 960), dev, 6, 0.03), rnd((320,), dev, 7)
    res, emb = rnd((m, 320), dev, 8), rnd((7, 320), dev, 9)
    out = ops.linear(x1.to(BF16), ops.pack_linear(w2, b2), x2=x2.to(BF16), res=res.to(BF16), emb=emb,
                     rows_per_batch=100, out_dtype=torch.float32)
    ref = F.linear(torch.cat([q(x1), q(x2)], -1), q(w2), b2) + q(res) + emb.repeat_interleave(100, dim=0)
    assert rel_err(out, ref) < 1e-4
    xs, ws, bs = rnd((130, 11520), dev, 10), rnd((1280, 11520), dev, 11, 0.01), rnd((1280,), dev, 12)
    assert ops._auto_split_k(130, 1280, 11520, 160) > 1
    out = ops.linear(xs.to(BF16), ops.pack_linear(ws, bs), out_dtype=torch.float32)
    assert rel_err(out, F.linear(q(xs), q(ws), bs)) < 1e-4
    for stride, ups in ((2, False), (1, True), (1, False)):
        xi = rnd((3, 64, 10, 14), dev, 13)
        wt, bias = rnd((96, 64, 3, 3), dev, 14, 1 / 24), rnd((96,), dev, 15)
        out = ops.conv3x3(q(xi).permute(0, 2, 3, 1).contiguous().to(BF16), ops.pack_conv3x3(wt, bias), stride=stride, upsample=ups)
        xin = F.interpolate(q(xi), scale_factor=2, mode="nearest") if ups else q(xi)
        assert rel_err(out.permute(0, 3, 1, 2), F.conv2d(xin, q(wt), bias, stride=stride, padding=1)) < 6e-3


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("m,n,k,tile_m", [(512, 1280, 1280, 64), (2048, 640, 640, 128), (130, 320, 1280, 64),
                                          (1000, 1288, 640, 128), (64, 64, 64, 64), (1, 200, 320, 128)])
def test_small_tile_64_columns(dev, dt, m, n, k, tile_m, monkeypatch):
    """sdmi_gemm_args.tile_n = 64 (round 3): 64 x 64 and 128 x 64 blocks for the launches of a few hundred rows (the UNet at
    batch 1 - 2).  Same k order inside the fp32 accumulation as the larger tiles: unsplit results are bit-identical to the
    128 x 160 tile's on every epilogue (bias, activation, residual, concat A, embedding rows, K padding, fp32 and 16-bit
    outputs); split-K and the fp32 reference within rounding; GEGLU, conv and the two-stage ring are refused."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    one = lambda m_, n_, k_, t_=128: 1
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    res = rnd((m, n), dev, 4)
    pw = ops.pack_linear(w, b, dtype=dt)

    def both(fn):
        """fn() on the forced small tile and on the forced 128-row / 160- or 128-column tile, both unsplit"""
        monkeypatch.setattr(ops, "_auto_split_k", one)
        monkeypatch.setattr(ops, "FORCE_TILE_N", 64)
        monkeypatch.setattr(ops, "FORCE_TILE_M", tile_m)
        small = fn()
        monkeypatch.setattr(ops, "FORCE_TILE_N", 0)
        monkeypatch.setattr(ops, "FORCE_TILE_M", 128)
        large = fn()
        monkeypatch.setattr(ops, "FORCE_TILE_M", 0)
        monkeypatch.setattr(ops, "_auto_split_k", ops._AUTO_SPLIT_K)
        return small, large

    for kw in (dict(res=res.to(dt)), dict(act=1), dict(res=res.to(dt), act=2), dict(out_dtype=torch.float32)):
        s_, l_ = both(lambda: ops.linear(x.to(dt), pw, **kw))
        assert torch.equal(s_, l_), kw
    ref = F.linear(qd(x), qd(w), b)
    s32, _ = both(lambda: ops.linear(x.to(dt), pw, out_dtype=torch.float32))
    assert rel_err(s32, ref) < 2e-5 * math.sqrt(k) + 1e-5
    # split-K on the small tile
    if k >= 256:
        monkeypatch.setattr(ops, "FORCE_TILE_N", 64)
        monkeypatch.setattr(ops, "FORCE_TILE_M", tile_m)
        monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: 2)
        sp = ops.linear(x.to(dt), pw, res=res.to(dt), out_dtype=None)
        monkeypatch.setattr(ops, "_auto_split_k", ops._AUTO_SPLIT_K)
        monkeypatch.setattr(ops, "FORCE_TILE_N", 0)
        monkeypatch.setattr(ops, "FORCE_TILE_M", 0)
        assert rel_err(sp, ref + qd(res)) < (6e-3 if dt == BF16 else 1e-3)
    # concat A + embedding rows + residual; K padding (k_valid)
    if k >= 128 and m >= 8:
        k1 = (k // 128) * 64
        x1, x2 = x[:, :k1].contiguous().to(dt), x[:, k1:].contiguous().to(dt)
        rpb = max(m // 4, 1)
        emb = rnd(((m + rpb - 1) // rpb, n), dev, 5)
        s_, l_ = both(lambda: ops.linear(x1, pw, x2=x2, emb=emb, rows_per_batch=rpb, res=res.to(dt), out_dtype=torch.float32))
        assert torch.equal(s_, l_)
        assert rel_err(s_, ref + qd(res) + emb.repeat_interleave(rpb, dim=0)[:m]) < 2e-5 * math.sqrt(k) + 1e-5
    kp = k - 24
    if kp > 0:
        wp, xp = rnd((n, kp), dev, 6, kp ** -0.5), rnd((m, kp), dev, 7)
        pwp = ops.pack_linear(wp, b, dtype=dt)
        s_, l_ = both(lambda: ops.linear(xp.to(dt), pwp, out_dtype=torch.float32))
        assert torch.equal(s_, l_)
        assert rel_err(s_, F.linear(qd(xp), qd(wp), b)) < 2e-5 * math.sqrt(k) + 1e-5
    # refusals through the C ABI: GEGLU, conv, the two-stage ring, a 256-row tile
    o = torch.empty(m, n, dtype=dt, device=dev)
    xd = x.to(dt)

    def args():
        a = ops.hip.GemmArgs()
        a.mode, a.A, a.lda, a.M, a.W, a.N, a.K = 0, xd.data_ptr(), k, m, pw.w.data_ptr(), n, pw.k
        a.out, a.ld_out, a.dtype, a.tile_n, a.alpha, a.batch, a.split_k = o.data_ptr(), n, 1 if dt == F16 else 0, 64, 1.0, 1, 1
        a.k_valid = k if k != pw.k else 0
        a.bias = pw.bias.data_ptr()
        return a
    if n % 8 == 0:
        a = args()
        assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == 0
        torch.cuda.synchronize()
        assert rel_err(o, ref) < (6e-3 if dt == BF16 else 1e-3)
    for field, val in (("geglu", 1), ("stages", 2), ("tile_m", 256), ("mode", 1), ("algo", 2)):
        a = args()
        if field == "geglu":
            a.N = (n // 32) * 16 or 16
        setattr(a, field, val)
        assert ops.hip.lib().sdmi_gemm(C.byref(a), None) != 0, field


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("b,h,w,cin,cout,stride,ups", [(2, 8, 8, 1280, 1280, 1, False), (1, 20, 12, 64, 320, 1, False),
                                                        (2, 16, 16, 128, 192, 2, False), (1, 8, 8, 128, 64, 1, True),
                                                        (1, 16, 16, 8, 320, 1, False)])
def test_small_tile_conv3x3(dev, dt, b, h, w, cin, cout, stride, ups, monkeypatch):
    """conv3x3 on the 128 x 64 block (tile_n = 64, mode 1): bit-identical to the 128-row / 160- or 128-column tile unsplit
    (bias + embedding rows + residual epilogue, stride 2, fused nearest x 2 upsampling, the general loader of conv_in),
    split-K and torch's conv2d within rounding; what the plan takes for the one-row-tile convs of the UNet at batch 2."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x = rnd((b, cin, h, w), dev, 31)
    wt, bias = rnd((cout, cin, 3, 3), dev, 32, (9 * cin) ** -0.5), rnd((cout,), dev, 33)
    xi = x.permute(0, 2, 3, 1).contiguous().to(dt)
    pw = ops.pack_conv3x3(wt, bias, dtype=dt)
    ho, wo = (h * (2 if ups else 1) + 2 - 3) // stride + 1, (w * (2 if ups else 1) + 2 - 3) // stride + 1
    emb = rnd((b, cout), dev, 34) if (stride == 1 and not ups) else None
    res = rnd((b, ho, wo, cout), dev, 35).to(dt) if (stride == 1 and not ups) else None
    kw = dict(stride=stride, upsample=ups, emb=emb, res=res)
    monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: 1)
    monkeypatch.setattr(ops, "FORCE_TILE_M", 128)
    large = ops.conv3x3(xi, pw, **kw)
    monkeypatch.setattr(ops, "FORCE_TILE_N", 64)
    small = ops.conv3x3(xi, pw, **kw)
    assert torch.equal(small, large)
    xin = F.interpolate(qd(x), scale_factor=2, mode="nearest") if ups else qd(x)
    ref = F.conv2d(xin, qd(wt), bias, stride=stride, padding=1)
    if emb is not None:
        ref = ref + emb[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    assert rel_err(small.permute(0, 3, 1, 2), ref) < (6e-3 if dt == BF16 else 1.5e-3)
    if pw.k >= 256:
        monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: 3)
        sp = ops.conv3x3(xi, pw, **kw)
        assert rel_err(sp.permute(0, 3, 1, 2), ref) < (6e-3 if dt == BF16 else 1.5e-3)
    monkeypatch.setattr(ops, "FORCE_TILE_N", 0)
    monkeypatch.setattr(ops, "FORCE_TILE_M", 0)
    monkeypatch.setattr(ops, "_auto_split_k", ops._AUTO_SPLIT_K)
    assert ops._plan_gemm(128, 1280, 11520, 160, False, False, True, True)[3] == 64        # 8 x 8 x 1280 at UNet batch 2
    assert ops._plan_gemm(16384, 640, 5760, 160, False, False, True, True)[3] != 64


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("b,h,w,cin,cout,split", [(1, 16, 16, 128, 320, 1), (3, 16, 16, 192, 160, 1), (2, 32, 32, 64, 640, 1),
                                                 (1, 64, 64, 128, 320, 1), (2, 32, 16, 256, 480, 1), (1, 16, 16, 256, 320, 2),
                                                 (2, 32, 32, 320, 160, 3)])
def test_conv3x3_halo_tiles(dev, dt, b, h, w, cin, cout, split, monkeypatch):
    """The halo-tile conv3x3 kernel (sdmi_conv_halo.hip: whole row strips of a 16 | 32 | 64-pixel-wide image on the 256 x 160 tile, the
    activation tile staged once per 64-channel chunk) against the tiled kernel (FORCE_ALGO = 1) and torch's conv2d: bias +
    embedding rows + residual, several images per launch, strips in the middle of an image (64 x 64: 16 blocks), one chunk
    and several, split-K over chunks.  K runs chunk-major there, so the two kernels agree to fp32 rounding of the sums,
    not bit for bit; the second launch of each must reproduce the first."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x = rnd((b, cin, h, w), dev, 41)
    wt, bias = rnd((cout, cin, 3, 3), dev, 42, (9 * cin) ** -0.5), rnd((cout,), dev, 43)
    xi = x.permute(0, 2, 3, 1).contiguous().to(dt)
    pw = ops.pack_conv3x3(wt, bias, dtype=dt)
    emb = rnd((b, cout), dev, 44)
    res = rnd((b, h, w, cout), dev, 45).to(dt)
    monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: split)
    monkeypatch.setattr(ops, "FORCE_TILE_M", 256)
    halo = ops.conv3x3(xi, pw, emb=emb, res=res)
    halo2 = ops.conv3x3(xi, pw, emb=emb, res=res)
    assert torch.equal(halo, halo2)
    monkeypatch.setattr(ops, "FORCE_ALGO", 1)
    tiled = ops.conv3x3(xi, pw, emb=emb, res=res)
    monkeypatch.setattr(ops, "FORCE_ALGO", 0)
    ref = F.conv2d(qd(x), qd(wt), bias, padding=1) + emb[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    tol = 6e-3 if dt == BF16 else 1.5e-3
    assert rel_err(halo.permute(0, 3, 1, 2), ref) < tol
    assert rel_err(tiled.permute(0, 3, 1, 2), ref) < tol
    # the two kernels against each other: the same products summed in another order, then rounded to 16 bits
    assert rel_err(halo.float(), tiled.float()) < (4e-3 if dt == BF16 else 5e-4)
    assert not torch.equal(halo, tiled) or cin == 64      # (one chunk: the same order - else this test is not reaching the new kernel)
    # no residual / embedding, 16-bit output straight from the accumulators
    plain = ops.conv3x3(xi, pw)
    assert rel_err(plain.permute(0, 3, 1, 2), F.conv2d(qd(x), qd(wt), bias, padding=1)) < tol


def test_small_tile_in_the_token_stream(dev, monkeypatch):
    """The 64-column tile as a producer and as a consumer of row statistics: a 1280-wide stream written on it carries
    40 slots per row (one per 32 columns), which the LayerNorm-folded projection reads in three batches."""
    ops = _ops()
    m, k, n = 512, 1280, 1280
    x0 = rnd((m, k), dev, 71).to(F16)
    w0, b0 = rnd((n, k), dev, 72, 0.03), rnd((n,), dev, 73)
    r = (rnd((m, n), dev, 74) * 2.0 + 0.7).to(F16)
    prod = ops.pack_linear(w0, b0, dtype=F16)
    assert ops._plan_gemm(m, n, k, 160, False, False, True)[3] == 64       # what the UNet's level-2 launches take at batch 2
    t, (st, slots) = ops.linear(x0, prod, res=r, row_stats=True)
    assert slots == 40 and st.shape == (m, 40, 2)
    monkeypatch.setattr(ops, "SMALL_TILE", False)
    t_l, (st_l, slots_l) = ops.linear(x0, prod, res=r, row_stats=True)
    monkeypatch.setattr(ops, "SMALL_TILE", True)
    assert slots_l != 40 and torch.equal(t, t_l)
    tot, tot_l = st.double().sum(1), st_l.double().sum(1)
    assert (tot - tot_l).abs().max() < 1e-5 * tot_l.abs().max()
    gamma, beta = 1.0 + 0.2 * rnd((n,), dev, 75), 0.3 * rnd((n,), dev, 76)
    ln = (gamma, beta, 1e-5)
    w1 = rnd((n, n), dev, 77, 0.03)
    cons = ops.pack_linear(w1, dtype=F16, ln=ln)
    out = ops.linear(t, cons, ln=ln, stats=(st, slots))              # small tile, folded LayerNorm from 40 slots
    monkeypatch.setattr(ops, "SMALL_TILE", False)
    out_l = ops.linear(t, cons, ln=ln, stats=(st_l, slots_l))
    monkeypatch.setattr(ops, "SMALL_TILE", True)
    ref = F.linear(F.layer_norm(t.float(), (n,), gamma, beta, 1e-5), w1)
    assert rel_err(out, ref) < 1.5e-3 and rel_err(out_l, ref) < 1.5e-3
    assert rel_err(out, out_l.float()) < 1.2e-3
    # two of these workgroups per CU (a 128 x 64 launch of > 256 blocks), repeated: the folded epilogue once lost the mean
    # term in a few outputs per launch there (a packed-fma operand form, see lnf_apply) - every launch must agree with the
    # 128 x 160 tile to rounding
    for (m2, k2, n2) in ((2100, 640, 1920), (4000, 1280, 1280)):
        x2 = (rnd((m2, k2), dev, 81) * (0.5 + 2.0 * torch.rand(m2, 1, device=dev)) + 1.5 * rnd((m2, 1), dev, 82)).to(F16)
        g2, b2 = 1.0 + 0.2 * rnd((k2,), dev, 83), 0.3 * rnd((k2,), dev, 84)
        pw2 = ops.pack_linear(rnd((n2, k2), dev, 85, 0.06), rnd((n2,), dev, 86), dtype=F16, ln=(g2, b2, 1e-5))
        st2 = ops.row_stats_of(x2)
        monkeypatch.setattr(ops, "_auto_split_k", lambda m_, n_, k_, t_=128: 1)
        monkeypatch.setattr(ops, "FORCE_TILE_M", 128)
        want = ops.linear(x2, pw2, ln=(g2, b2, 1e-5), stats=st2).float()
        monkeypatch.setattr(ops, "FORCE_TILE_N", 64)
        for _ in range(20):
            got = ops.linear(x2, pw2, ln=(g2, b2, 1e-5), stats=st2).float()
            assert (got - want).abs().max() < 2e-2
        monkeypatch.setattr(ops, "FORCE_TILE_N", 0)
        monkeypatch.setattr(ops, "FORCE_TILE_M", 0)
        monkeypatch.setattr(ops, "_auto_split_k", ops._AUTO_SPLIT_K)
    # a stream too wide for 48 slots keeps the larger tiles
    wide_w = ops.pack_linear(rnd((1920, 320), dev, 78, 0.05), dtype=F16)
    _, (st_w, slots_w) = ops.linear(rnd((256, 320), dev, 79).to(F16), wide_w, row_stats=True)
    assert slots_w <= 48


@pytest.mark.parametrize("dt", [BF16, F16])
def test_wide_tile_256x256x32(dev, dt, monkeypatch):
    """The 256 x 256 x 32 tile (sdmi_gemm_wide.h, tile_n = 256) against the fp32 reference and against the tiled kernel
    on the same inputs: linear with bias + residual / SiLU, ragged M, concat A, split-K (slices of odd and even
    length), conv stride 1 / 2 / upsampled with the embedding add.  Tolerance (both comparisons) = the 16-bit output rounding: the two kernels
    accumulate in the same k order, but the plan may split K for one and not for the other."""
    ops = _ops()
    tol = 6e-3 if dt == BF16 else 1.5e-3
    qd = lambda t: t.to(dt).float()
    def both(f):
        monkeypatch.setattr(ops, "FORCE_WIDE", 0)
        monkeypatch.setattr(ops, "WIDE_TILE", False)
        tiled = f()
        monkeypatch.setattr(ops, "FORCE_WIDE", 1)
        monkeypatch.setattr(ops, "WIDE_TILE", True)
        wide = f()
        return tiled, wide
    for (m, n, k) in ((1000, 512, 320), (300, 256, 64), (520, 1280, 2560)):
        x, w, b, res = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3), rnd((m, n), dev, 4)
        pw = ops.pack_linear(w, b, dtype=dt)
        assert pw.w.shape[0] % 256 == 0
        tiled, wide = both(lambda: ops.linear(x.to(dt), pw, res=res.to(dt)))
        assert rel_err(tiled, wide) < tol
        assert rel_err(wide, F.linear(qd(x), qd(w), b) + qd(res)) < tol
        tiled, wide = both(lambda: ops.linear(x.to(dt), pw, act=1))
        assert rel_err(tiled, wide) < tol
        assert rel_err(wide, F.silu(F.linear(qd(x), qd(w), b))) < tol
        for sk in (2, 3):
            if k // 64 < sk:
                continue
            monkeypatch.setattr(ops, "_auto_split_k", lambda *a, s_=sk, **kw: s_)
            tiled, wide = both(lambda: ops.linear(x.to(dt), pw, res=res.to(dt)))
            monkeypatch.setattr(ops, "_auto_split_k", ops._AUTO_SPLIT_K)
            assert rel_err(wide, F.linear(qd(x), qd(w), b) + qd(res)) < tol      # (slice boundaries differ from the tiled kernel's)
            assert rel_err(wide, tiled) < tol
    x1, x2 = rnd((600, 640), dev, 5), rnd((600, 640), dev, 6)
    w, b = rnd((1280, 1280), dev, 7, 0.03), rnd((1280,), dev, 8)
    pw = ops.pack_linear(w, b, dtype=dt)
    tiled, wide = both(lambda: ops.linear(x1.to(dt), pw, x2=x2.to(dt)))
    assert rel_err(tiled, wide) < tol
    assert rel_err(wide, F.linear(torch.cat([qd(x1), qd(x2)], -1), qd(w), b)) < tol
    for (bb, h, ww, cin, cout, stride, ups) in ((2, 17, 13, 256, 256, 1, False), (2, 8, 8, 128, 512, 1, True), (2, 16, 16, 64, 256, 2, False)):
        xi = rnd((bb, cin, h, ww), dev, 9)
        wt, bias, emb = rnd((cout, cin, 3, 3), dev, 10, (9 * cin) ** -0.5), rnd((cout,), dev, 11), rnd((bb, cout), dev, 12)
        pc = ops.pack_conv3x3(wt, bias, dtype=dt)
        xin = qd(xi).permute(0, 2, 3, 1).contiguous().to(dt)
        tiled, wide = both(lambda: ops.conv3x3(xin, pc, stride=stride, upsample=ups, emb=emb))
        assert rel_err(tiled, wide) < tol
        xr = F.interpolate(qd(xi), scale_factor=2, mode="nearest") if ups else qd(xi)
        ref = F.conv2d(xr, qd(wt), bias, stride=stride, padding=1) + emb[:, :, None, None]
        assert rel_err(wide.permute(0, 3, 1, 2), ref) < tol
    # the C entry point refuses the tile where it does not apply (fp32 output without split-K), ops falls back silently
    x, w = rnd((300, 64), dev, 13), rnd((256, 64), dev, 14)
    pw = ops.pack_linear(w, None, dtype=dt)
    monkeypatch.setattr(ops, "FORCE_WIDE", 1)
    out = ops.linear(x.to(dt), pw, out_dtype=torch.float32)
    assert rel_err(out, F.linear(qd(x), qd(w))) < 1e-4
    a = ops.hip.GemmArgs()
    o16 = torch.empty(300, 256, dtype=dt, device=dev)
    a.mode, a.A, a.lda, a.M, a.N, a.K, a.W, a.out, a.ld_out = 0, x.to(dt).data_ptr(), 64, 300, 256, 64, pw.w.data_ptr(), o16.data_ptr(), 256
    a.dtype, a.alpha, a.tile_n, a.out_kind, a.batch, a.split_k = ops._dt(dt), 1.0, 256, 1, 1, 1
    assert ops.hip.lib().sdmi_gemm_wide_applies(C.byref(a)) == 0
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) != 0


@pytest.mark.parametrize("dt", [F16, BF16])
def test_wide_tile_geglu_epilogue(dev, dt, monkeypatch):
    """GEGLU on the 256 x 256 x 32 tile (value / gate column blocks paired inside a wave, two runs of 32 gated columns per
    wave) against the tiled kernel and the fp32 formula value * gelu_erf(gate) (attention.py:72-91): ragged M, the three
    UNet widths."""
    ops = _ops()
    tol = 8e-3 if dt == BF16 else 2e-3
    for m, inner, k in ((1000, 512, 320), (4096, 1280, 320), (777, 2560, 640), (520, 5120, 1280)):
        x = rnd((m, k), dev, 1).to(dt)
        w, b = rnd((2 * inner, k), dev, 2, k ** -0.5), rnd((2 * inner,), dev, 3)
        pw = ops.pack_geglu(w, b, dtype=dt)
        assert pw.w.shape[0] % 256 == 0
        monkeypatch.setattr(ops, "FORCE_WIDE", 0)
        monkeypatch.setattr(ops, "WIDE_TILE", False)
        tiled = ops.linear(x, pw)
        monkeypatch.setattr(ops, "FORCE_WIDE", 1)
        monkeypatch.setattr(ops, "WIDE_TILE", True)
        a = ops.hip.GemmArgs()
        wide = ops.linear(x, pw)
        monkeypatch.setattr(ops, "FORCE_WIDE", 0)
        y = F.linear(x.float(), w.to(dt).float(), b)
        ref = y[:, :inner] * F.gelu(y[:, inner:])
        assert wide.shape == (m, inner)
        assert rel_err(wide, tiled) < tol, (m, inner, k)
        assert rel_err(wide, ref) < tol, (m, inner, k)


def test_plan_takes_the_wide_tile_only_where_measured(dev):
    ops = _ops()
    assert ops._plan_gemm(4096, 3840, 1280, 160, True)[3] == 256          # level-2 qkv: 55 -> 43 us
    assert ops._plan_gemm(131072, 512, 4608, 128, True)[3] == 256         # decoder 512-channel conv
    assert ops._plan_gemm(4096, 1280, 11520, 160, True)[3] == 160         # 80 wide tiles: split-K on the small tile wins
    assert ops._plan_gemm(65536, 1280, 320, 160, True)[3] == 160          # short K: not measured
    assert ops._plan_gemm(4096, 3840, 1280, 160, False)[3] == 160
    assert ops._plan_gemm(65536, 2560, 320, 128, True, True)[3] == 256        # level-0 GEGLU: 193 -> 170 us
    assert ops._plan_gemm(16384, 5120, 640, 128, True, True)[3] == 256        # level-1 GEGLU: 142 -> 132 us
    assert ops._plan_gemm(4096, 10240, 1280, 128, True, True)[3] == 128       # level-2 GEGLU: the wide tile loses 3 %
    assert ops._plan_gemm(8192, 2560, 320, 128, True, True)[3] == 128         # batch 1: not measured


@pytest.mark.parametrize("b,h,w,cin,cout,stride,ups", [
    (2, 16, 16, 64, 128, 1, False), (1, 32, 32, 320, 320, 1, False), (2, 16, 16, 128, 128, 2, False),
    (2, 8, 8, 64, 64, 1, True), (2, 12, 20, 192, 96, 1, False), (1, 9, 7, 64, 32, 2, False),
    (2, 8, 8, 1280, 1280, 1, False),
])
def test_conv3x3(dev, b, h, w, cin, cout, stride, ups):
    ops = _ops()
    x = rnd((b, cin, h, w), dev, 1)
    wt, bias = rnd((cout, cin, 3, 3), dev, 2, (9 * cin) ** -0.5), rnd((cout,), dev, 3)
    x_nhwc = q(x).permute(0, 2, 3, 1).contiguous().to(BF16)
    out = ops.conv3x3(x_nhwc, ops.pack_conv3x3(wt, bias), stride=stride, upsample=ups)
    xi = F.interpolate(q(x), scale_factor=2, mode="nearest") if ups else q(x)
    ref = F.conv2d(xi, q(wt), bias, stride=stride, padding=1)
    assert out.shape == (b, ref.shape[2], ref.shape[3], cout)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 6e-3


@pytest.mark.parametrize("h,w", [(16, 16), (12, 20), (7, 9)])
def test_conv3x3_asymmetric_pad(dev, h, w):
    """VAE encoder DownSample: F.pad(x, (0, 1, 0, 1)) + conv3x3 stride 2 without padding (model.py:87,99-102)."""
    ops = _ops()
    x = rnd((2, 64, h, w), dev, 1)
    wt, bias = rnd((96, 64, 3, 3), dev, 2, 1 / 24), rnd((96,), dev, 3)
    out = ops.conv3x3(q(x).permute(0, 2, 3, 1).contiguous().to(BF16), ops.pack_conv3x3(wt, bias), stride=2, asym_pad=True)
    ref = F.conv2d(F.pad(q(x), (0, 1, 0, 1)), q(wt), bias, stride=2, padding=0)
    assert out.shape == (2, ref.shape[2], ref.shape[3], 96)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 6e-3


def test_conv3x3_edges(dev):
    """conv_in (4 -> C, channels padded to 8), conv_out (C -> 4, fp32 NCHW out) and emb + residual epilogue."""
    ops = _ops()
    x = rnd((2, 4, 16, 16), dev, 1)
    wt, bias = rnd((64, 4, 3, 3), dev, 2, 1 / 6), rnd((64,), dev, 3)
    x8 = ops.nchw_to_nhwc(x)
    assert x8.shape == (2, 16, 16, 8)
    assert torch.equal(x8[..., :4].float(), q(x).permute(0, 2, 3, 1))
    assert x8[..., 4:].abs().max().item() == 0
    out = ops.conv3x3(x8, ops.pack_conv3x3(wt, bias))
    assert rel_err(out.permute(0, 3, 1, 2), F.conv2d(q(x), q(wt), bias, padding=1)) < 6e-3

    h = rnd((2, 64, 16, 16), dev, 4)
    w2, b2 = rnd((4, 64, 3, 3), dev, 5, 1 / 24), rnd((4,), dev, 6)
    h_nhwc = q(h).permute(0, 2, 3, 1).contiguous().to(BF16)
    out = ops.conv3x3(h_nhwc, ops.pack_conv3x3(w2, b2), out_nchw_f32=True)
    assert out.shape == (2, 4, 16, 16) and out.dtype == torch.float32
    assert rel_err(out, F.conv2d(q(h), q(w2), b2, padding=1)) < 1e-4

    w3, b3 = rnd((64, 64, 3, 3), dev, 7, 1 / 24), rnd((64,), dev, 8)
    emb, res = rnd((2, 64), dev, 9), rnd((2, 16, 16, 64), dev, 10)
    out = ops.conv3x3(h_nhwc, ops.pack_conv3x3(w3, b3), emb=emb, res=res.to(BF16))
    ref = F.conv2d(q(h), q(w3), b3, padding=1) + emb[:, :, None, None] + q(res).permute(0, 3, 1, 2)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 6e-3


@pytest.mark.parametrize("dt", [BF16, torch.float16])
@pytest.mark.parametrize("b,h,w,cin,cout,kind", [(2, 20, 24, 128, 3, "nchw"), (1, 33, 17, 320, 4, "nchw"),
                                                 (3, 9, 11, 64, 8, "nhwc"), (2, 16, 16, 96, 16, "nhwc"), (1, 5, 3, 32, 1, "nchw")])
def test_conv3x3_narrow(dev, dt, b, h, w, cin, cout, kind):
    """conv3x3 with <= 16 output channels (UNet `out` 320 -> 4, decoder conv_out 128 -> 3) runs the byte-bound narrow
    kernel (sdmi_conv_narrow.hip): against F.conv2d on the rounded operands and against the tiled kernel (algo = 1) -
    templated (Cin 128 / 320) and run-time channel loops, odd Cin / 32, row counts that are not multiples of 16, image
    borders, fp32 NCHW and 16-bit NHWC outputs."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x = rnd((b, cin, h, w), dev, 51)
    wt, bias = rnd((cout, cin, 3, 3), dev, 52, (9 * cin) ** -0.5), rnd((cout,), dev, 53)
    x_nhwc = qd(x).permute(0, 2, 3, 1).contiguous().to(dt)
    pw = ops.pack_conv3x3(wt, bias, dtype=dt)
    ref = F.conv2d(qd(x), qd(wt), bias, padding=1)
    outs = {}
    for algo in (0, 1):
        ops.FORCE_ALGO = algo
        try:
            o = ops.conv3x3(x_nhwc, pw, out_nchw_f32=(kind == "nchw"))
        finally:
            ops.FORCE_ALGO = 0
        outs[algo] = o if kind == "nchw" else o.permute(0, 3, 1, 2).float()
    tol = 1e-4 if kind == "nchw" else (6e-3 if dt == BF16 else 1e-3)
    assert outs[0].shape == ref.shape and rel_err(outs[0], ref) < tol and rel_err(outs[1], ref) < tol


def test_batched_gemm(dev):
    ops = _ops()
    z, m, n, k = 3, 256, 256, 512
    a, b = rnd((z, m, k), dev, 1), rnd((z, n, k), dev, 2)
    out = ops.gemm_nt_batched(a.to(BF16), b.to(BF16), torch.float32, alpha=0.5)
    ref = 0.5 * torch.einsum("zmk,znk->zmn", q(a), q(b))
    assert rel_err(out, ref) < 1e-4
    a1 = rnd((1, m, k), dev, 3)
    bias = rnd((m,), dev, 4)
    out = ops.gemm_nt_batched(a1.to(BF16), b.to(BF16), BF16, bias_rows=bias)
    ref = torch.einsum("mk,znk->zmn", q(a1[0]), q(b)) + bias[None, :, None]
    assert rel_err(out, ref) < 6e-3
    # the same batched problem on the 64-column tile through the C ABI (per-row bias: the general epilogue path)
    ab, bb = a1.to(BF16).contiguous(), b.to(BF16).contiguous()
    o64 = torch.empty(z, m, n, dtype=BF16, device=dev)
    g = ops.hip.GemmArgs()
    g.mode, g.A, g.lda, g.M, g.W, g.N, g.K = 0, ab.data_ptr(), k, m, bb.data_ptr(), n, k
    g.out, g.ld_out, g.out_kind, g.dtype, g.alpha, g.batch, g.split_k = o64.data_ptr(), n, 0, 0, 1.0, z, 1
    g.stride_a, g.stride_w, g.stride_out = 0, n * k, m * n
    g.bias, g.bias_per_row, g.tile_n, g.tile_m = bias.data_ptr(), 1, 64, 64
    assert ops.hip.lib().sdmi_gemm(C.byref(g), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(o64, out)


# ----------------------------------------------------------------------------- norms


@pytest.mark.parametrize("dt", [F16, BF16])
def test_dup_rows_store_is_a_bit_exact_copy(dev, dt, monkeypatch):
    """sdmi_gemm_args.dup_rows: every output row stored at m and m + M - what torch.cat([t, t]) of the plain launch
    gives, bit for bit - on every kernel whose fast 16-bit epilogue carries it: the tiled kernel (linear with bias +
    residual, 128- and 160-column tiles; conv3x3 with bias + embedding + residual, incl. the 8-channel conv_in shape), the
    256 x 256 x 32 tile and the A-panel-stationary kernel (K = 320 with a fused LayerNorm); refusals where the epilogue
    does not apply."""
    ops = _ops()
    from stablediffusion_amd import hip
    # linear, tiled kernel
    for m, n, k in ((4096, 320, 320), (1000, 640, 192), (8192, 1280, 1280)):
        x, w, b = rnd((m, k), dev, 1).to(dt), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
        res = rnd((m, n), dev, 4).to(dt)
        pw = ops.pack_linear(w, b, dtype=dt)
        one = ops.linear(x, pw, res=res)
        two = ops.linear(x, pw, res=res, dup_rows=True)
        assert two.shape == (2 * m, n) and torch.equal(two, torch.cat([one, one])), (m, n, k)
    # the 256 x 256 x 32 tile
    m, n, k = 16384, 1280, 2048
    x, w = rnd((m, k), dev, 5).to(dt), rnd((n, k), dev, 6, k ** -0.5)
    pw = ops.pack_linear(w, None, dtype=dt)
    assert pw.w.shape[0] % 256 == 0
    monkeypatch.setattr(ops, "FORCE_WIDE", 1)
    one, two = ops.linear(x, pw), ops.linear(x, pw, dup_rows=True)
    monkeypatch.setattr(ops, "FORCE_WIDE", 0)
    assert torch.equal(two, torch.cat([one, one]))
    # the panel kernel with its LayerNorm (K = 320, M >= 16384)
    m = 16384
    x, w, b = rnd((m, 320), dev, 7).to(dt), rnd((320, 320), dev, 8, 320 ** -0.5), rnd((320,), dev, 9)
    ln = (1 + 0.1 * rnd((320,), dev, 10), 0.1 * rnd((320,), dev, 11))
    pw = ops.pack_linear(w, b, dtype=dt)
    one, two = ops.linear(x, pw, ln=ln), ops.linear(x, pw, ln=ln, dup_rows=True)
    assert torch.equal(two, torch.cat([one, one]))
    # conv3x3: ResBlock-shaped (bias + embedding + residual) and the conv_in shape (8 input channels)
    for bsz, h, cin, cout in ((4, 32, 64, 320), (3, 64, 8, 320)):
        xi = rnd((bsz, h, h, cin), dev, 12).to(dt)
        wc, bc = rnd((cout, cin, 3, 3), dev, 13, (9 * cin) ** -0.5), rnd((cout,), dev, 14)
        emb, res = rnd((bsz, cout), dev, 15), rnd((bsz, h, h, cout), dev, 16).to(dt)
        pw = ops.pack_conv3x3(wc, bc, dtype=dt)
        one = ops.conv3x3(xi, pw, emb=emb, res=res)
        two = ops.conv3x3(xi, pw, emb=emb, res=res, dup_rows=True)
        assert two.shape == (2 * bsz, h, h, cout) and torch.equal(two, torch.cat([one, one]))
    # refused where the fast epilogue does not apply: fp32 output, GEGLU, a narrow conv
    x = rnd((256, 320), dev, 17).to(dt)
    pw = ops.pack_linear(rnd((320, 320), dev, 18), None, dtype=dt)
    with pytest.raises(hip.SdmiError):
        ops.linear(x, pw, out_dtype=torch.float32, dup_rows=True)
    pg = ops.pack_geglu(rnd((2 * 320, 320), dev, 19), rnd((2 * 320,), dev, 20), dtype=dt)
    with pytest.raises(hip.SdmiError):
        ops.linear(x, pg, dup_rows=True)
    pn = ops.pack_conv3x3(rnd((4, 64, 3, 3), dev, 21), None, dtype=dt)
    with pytest.raises(hip.SdmiError):
        ops.conv3x3(rnd((1, 16, 16, 64), dev, 22).to(dt), pn, dup_rows=True)


def test_bf16_operands_with_f16_residual_stream(dev):
    """The bf16 operand policy's f16 residual stream (sdmi_gemm_args.out_f16 / res_f16 / a_f16, SDMI_DTYPE_F16_TO_BF16):
    bf16 x bf16 GEMMs / convs that add an f16 residual and store f16, norms that read f16 and write the bf16 operand, the
    panel kernel whose fused LayerNorm reads the f16 stream - each against the fp32 formula on the rounded inputs, to the
    rounding of its OUTPUT type (f16: 2^-11, bf16: 2^-8 relative); and the refusals."""
    ops = _ops()
    from stablediffusion_amd import hip
    qb = lambda t: t.to(BF16).float()
    qh = lambda t: t.to(F16).float()
    # linear: tiled kernel (fast epilogue, the 4-column path, split-K reduce) and an N the fast path does not take
    for m, n, k in ((4096, 320, 320), (300, 324, 192), (256, 1280, 5120)):
        x, w, b, res = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3), rnd((m, n), dev, 4)
        pw = ops.pack_linear(w, b, dtype=BF16, out_dtype=F16)
        out = ops.linear(x.to(BF16), pw, res=res.to(F16))
        assert out.dtype == F16
        assert rel_err(out, F.linear(qb(x), qb(w), b) + qh(res)) < 1.5e-3, (m, n, k)
    # conv3x3 with embedding + f16 residual -> f16
    xi, wc, bc = rnd((2, 24, 24, 64), dev, 5), rnd((96, 64, 3, 3), dev, 6, 576 ** -0.5), rnd((96,), dev, 7)
    emb, res = rnd((2, 96), dev, 8), rnd((2, 24, 24, 96), dev, 9)
    pc = ops.pack_conv3x3(wc, bc, dtype=BF16, out_dtype=F16)
    out = ops.conv3x3(xi.to(BF16), pc, emb=emb, res=res.to(F16))
    ref = F.conv2d(qb(xi).permute(0, 3, 1, 2), qb(wc), bc, padding=1) + emb[:, :, None, None] + qh(res).permute(0, 3, 1, 2)
    assert out.dtype == F16 and rel_err(out.permute(0, 3, 1, 2), ref) < 1.5e-3
    # norms: f16 in -> bf16 out
    xs = (rnd((2, 40, 40, 320), dev, 10) * 1.5 + 0.3).to(F16)
    g, bta = 1 + 0.1 * rnd((320,), dev, 11), 0.1 * rnd((320,), dev, 12)
    for fused_hw in (xs, xs[:, :8]):               # two-launch and fused single-launch forms
        y = ops.groupnorm(fused_hw.contiguous(), g, bta, 1e-5, True, out_dtype=BF16)
        xf = fused_hw.float()
        nb_, hh_, ww_, c_ = xf.shape
        v = xf.reshape(nb_, hh_ * ww_, 32, c_ // 32)
        mu, var = v.mean(dim=(1, 3), keepdim=True), v.var(dim=(1, 3), unbiased=False, keepdim=True)
        yr = ((v - mu) * torch.rsqrt(var + 1e-5)).reshape(xf.shape) * g + bta
        assert y.dtype == BF16 and rel_err(y, F.silu(yr)) < 6e-3
    t = (rnd((1000, 640), dev, 13) * 2 + 0.5).to(F16)
    g, bta = 1 + 0.1 * rnd((640,), dev, 14), 0.1 * rnd((640,), dev, 15)
    y = ops.layernorm(t, g, bta, out_dtype=BF16)
    assert y.dtype == BF16 and rel_err(y, F.layer_norm(t.float(), (640,), g, bta)) < 6e-3
    # the panel kernel: LayerNorm of the f16 stream fused into a bf16 projection == separate launches, to bf16 rounding
    m = 16384
    t = (rnd((m, 320), dev, 16) * 2 + 0.5).to(F16)
    g, bta = 1 + 0.1 * rnd((320,), dev, 17), 0.1 * rnd((320,), dev, 18)
    w = rnd((960, 320), dev, 19, 320 ** -0.5)
    pw = ops.pack_linear(w, None, dtype=BF16)
    fused = ops.linear(t, pw, ln=(g, bta))
    ref = F.linear(qb(F.layer_norm(t.float(), (320,), g, bta)), qb(w))
    assert fused.dtype == BF16 and rel_err(fused, ref) < 8e-3
    # refusals: the flags are a bf16-operand matter; a GEGLU output is never the stream; a_f16 needs the LayerNorm
    a = hip.GemmArgs()
    x16 = torch.zeros(256, 320, dtype=F16, device=dev)
    o16 = torch.zeros(256, 320, dtype=F16, device=dev)
    pw16 = ops.pack_linear(rnd((320, 320), dev, 20), None, dtype=F16)
    a.mode, a.A, a.lda, a.M, a.N, a.K, a.W, a.out, a.ld_out = 0, x16.data_ptr(), 320, 256, 320, 320, pw16.w.data_ptr(), o16.data_ptr(), 320
    a.dtype, a.alpha, a.tile_n, a.batch, a.split_k, a.out_f16 = hip.DTYPE_F16, 1.0, 160, 1, 1, 1
    assert hip.lib().sdmi_gemm(C.byref(a), None) == -1
    a.dtype, a.out_f16, a.a_f16 = hip.DTYPE_BF16, 0, 1
    assert hip.lib().sdmi_gemm(C.byref(a), None) == -1
    with pytest.raises(hip.SdmiError):
        ops.groupnorm(xs.to(BF16), g[:320], bta[:320], 1e-5, True, out_dtype=F16)      # bf16 -> f16 does not exist
    torch.cuda.synchronize()


@pytest.mark.parametrize("b,hw,c1,c2,silu,eps", [
    (2, 1024, 320, 0, True, 1e-5), (2, 256, 640, 320, True, 1e-5), (1, 64, 1280, 1280, True, 1e-5),
    (2, 4096, 128, 0, True, 1e-6), (1, 100, 1280, 640, False, 1e-6), (3, 70000, 128, 0, True, 1e-6),
    (2, 16, 320, 0, False, 1e-6),
])
def test_groupnorm(dev, b, hw, c1, c2, silu, eps):
    ops = _ops()
    x1 = rnd((b, hw, c1), dev, 1) * 2 + 0.7
    x2 = rnd((b, hw, c2), dev, 2) * 0.5 - 1.0 if c2 else None
    c = c1 + c2
    gamma, beta = 1 + 0.2 * rnd((c,), dev, 3), 0.1 * rnd((c,), dev, 4)
    out = ops.groupnorm(x1.to(BF16), gamma, beta, eps, silu, x2=x2.to(BF16) if c2 else None)
    xin = q(x1) if not c2 else torch.cat([q(x1), q(x2)], -1)
    ref = F.group_norm(xin.permute(0, 2, 1), 32, gamma, beta, eps)   # [b, c, hw]
    if silu:
        ref = F.silu(ref)
    assert out.shape == (b, hw, c)
    assert rel_err(out.permute(0, 2, 1), ref) < 6e-3


@pytest.mark.parametrize("rows,c", [(1000, 320), (77, 640), (4096, 1280), (3, 1024)])
def test_layernorm(dev, rows, c):
    ops = _ops()
    x = rnd((rows, c), dev, 1) * 3 + 1
    gamma, beta = 1 + 0.2 * rnd((c,), dev, 2), 0.1 * rnd((c,), dev, 3)
    out = ops.layernorm(x.to(BF16), gamma, beta, 1e-5)
    ref = F.layer_norm(q(x), (c,), gamma, beta, 1e-5)
    assert rel_err(out, ref) < 6e-3


# ----------------------------------------------------------------------------- attention


def _attn_ref(qq, kk, vv, nb, heads, nq, nk, scale):
    def split(t, n):
        return t.reshape(nb, n, heads, 64).permute(0, 2, 1, 3)
    s = torch.einsum("bhid,bhjd->bhij", split(qq, nq), split(kk, nk)) * scale
    o = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), split(vv, nk))
    return o.permute(0, 2, 1, 3).reshape(nb * nq, heads * 64)


@pytest.mark.parametrize("nb,heads,nq,nk", [(2, 5, 1024, 1024), (2, 20, 64, 64), (2, 10, 256, 77), (1, 3, 16, 16),
                                            (1, 2, 144, 144), (1, 5, 4096, 77), (1, 1, 200, 333), (3, 3, 300, 200),
                                            (4, 4, 130, 64)])
def test_attention(dev, nb, heads, nq, nk):
    ops = _ops()
    c = heads * 64
    qq, kk, vv = rnd((nb * nq, c), dev, 1), rnd((nb * nk, c), dev, 2), rnd((nb * nk, c), dev, 3)
    out = ops.attention_d64(qq.to(BF16), kk.to(BF16), vv.to(BF16), heads, nb, nq, nk, 0.125)
    ref = _attn_ref(q(qq), q(kk), q(vv), nb, heads, nq, nk, 0.125)
    assert rel_err(out, ref) < 1e-2


def test_attention_fused_qkv_and_spike(dev):
    """q/k/v as column slices of one [M, 3C] buffer; one key spiked so the running max jumps mid-sequence
    (forces the online-softmax rescale path with a large factor)."""
    ops = _ops()
    nb, heads, n = 1, 2, 512
    c = heads * 64
    qkv = rnd((nb * n, 3 * c), dev, 1)
    qkv[300, c:2 * c] *= 12.0      # key 300 dominates for aligned queries
    qkv[5, :c] = qkv[300, c:2 * c] / 12.0
    buf = qkv.to(BF16)
    out = ops.attention_d64(buf[:, :c], buf[:, c:2 * c], buf[:, 2 * c:], heads, nb, n, n, 0.125)
    f = buf.float()
    ref = _attn_ref(f[:, :c], f[:, c:2 * c], f[:, 2 * c:], nb, heads, n, n, 0.125)
    assert rel_err(out, ref) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_device_packers_equal_host_packers_and_feed_gemm(dev, dtype):
    """sdmi_pack_* (device kernels) produce the same bits as sdmi_pack_*_host (pinned on the CPU against the torch
    reference, tests/test_host_cpu.py), and sdmi_gemm computes the right products from buffers packed entirely through
    the C ABI - no torch arithmetic between the fp32 weight and the kernel."""
    ops = _ops()
    same = lambda a, b: a.w.shape == b.w.shape and torch.equal(a.w.view(torch.int16).cpu(), b.w.view(torch.int16))
    w, b = rnd((332, 200), dev, 1), rnd((332,), dev, 2)
    assert same(ops.pack_linear(w, b, dtype=dtype), ops.pack_linear(w.cpu(), b.cpu(), dtype=dtype))
    ws = [rnd((320, 96), dev, 3), rnd((320, 96), dev, 4), rnd((17, 96), dev, 5)]
    assert same(ops.pack_linear_cat(ws, dtype=dtype), ops.pack_linear_cat([t.cpu() for t in ws], dtype=dtype))
    wg, bg = rnd((2 * 96, 72), dev, 6), rnd((2 * 96,), dev, 7)
    pg_d, pg_h = ops.pack_geglu(wg, bg, dtype=dtype), ops.pack_geglu(wg.cpu(), bg.cpu(), dtype=dtype)
    assert same(pg_d, pg_h) and torch.equal(pg_d.bias.cpu(), pg_h.bias)
    wc = rnd((40, 12, 3, 3), dev, 8)
    assert same(ops.pack_conv3x3(wc, dtype=dtype), ops.pack_conv3x3(wc.cpu(), dtype=dtype))
    # products from C-packed weights
    f32 = dtype == torch.float32
    cast = (lambda t: t) if f32 else (lambda t: t.to(dtype))
    qd = (lambda t: t) if f32 else (lambda t: t.to(dtype).float())
    tol = 2e-5 if f32 else 6e-3
    x = rnd((70, 200), dev, 9)
    out = ops.linear(cast(x), ops.pack_linear(w, b, dtype=dtype), out_dtype=torch.float32)
    assert rel_err(out, F.linear(qd(x), qd(w), b)) < tol
    xg = rnd((50, 72), dev, 10)
    og = ops.linear(cast(xg), pg_d, out_dtype=torch.float32)
    a_, g_ = F.linear(qd(xg), qd(wg), bg).chunk(2, dim=-1)
    assert rel_err(og, a_ * F.gelu(g_)) < tol
    xi = rnd((2, 12, 9, 7), dev, 11)
    xin = xi.permute(0, 2, 3, 1).contiguous()
    if not f32:
        xin = torch.cat([xin, torch.zeros(2, 9, 7, 4, device=dev)], dim=-1).to(dtype)     # Cin 12 -> the packed 16
    oc = ops.conv3x3(xin, ops.pack_conv3x3(wc, rnd((40,), dev, 12), dtype=dtype), out_nchw_f32=True)
    assert rel_err(oc, F.conv2d(qd(xi), qd(wc), rnd((40,), dev, 12), padding=1)) < tol


@pytest.mark.parametrize("dt", [BF16, torch.float16])
@pytest.mark.parametrize("m,n,bias,res", [(16384, 320, True, True), (16420, 320, False, False), (16384, 960, False, False),
                                          (17111, 640, True, True), (20000, 1280, True, False)])
def test_panel_gemm_k320(dev, dt, m, n, bias, res):
    """The A-panel-stationary kernel (sdmi_gemm_args.algo = 2; K = 320, every workgroup streams all of W past its
    resident rows of A) against F.linear on the rounded operands and against the tiled kernel (algo = 1), incl. a
    ragged last row panel and several column tiles."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x, w = rnd((m, 320), dev, 31), rnd((n, 320), dev, 32, 0.06)
    b = rnd((n,), dev, 33) if bias else None
    r = rnd((m, n), dev, 34) if res else None
    pw = ops.pack_linear(w, b, dtype=dt)
    outs = {}
    for algo in (2, 1):
        ops.FORCE_ALGO = algo
        try:
            outs[algo] = ops.linear(x.to(dt), pw, res=r.to(dt) if res else None)
        finally:
            ops.FORCE_ALGO = 0
    ref = F.linear(qd(x), qd(w), b) + (qd(r) if res else 0)
    tol = 8e-3 if dt == BF16 else 1.5e-3
    assert rel_err(outs[2], ref) < tol and rel_err(outs[1], ref) < tol
    assert rel_err(outs[2], outs[1].float()) < tol
    auto = ops.linear(x.to(dt), pw, res=r.to(dt) if res else None)       # what the library picks on its own: the tiled kernel
    assert torch.equal(auto, outs[1])                                    # (the panel kernel only when a LayerNorm rides along)


@pytest.mark.parametrize("dt", [BF16, torch.float16])
@pytest.mark.parametrize("m,inner", [(16384, 1280), (16500, 128)])
def test_panel_gemm_k320_geglu(dev, dt, m, inner):
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x, w, b = rnd((m, 320), dev, 35), rnd((2 * inner, 320), dev, 36, 0.06), rnd((2 * inner,), dev, 37)
    pw = ops.pack_geglu(w, b, dtype=dt)
    outs = {}
    for algo in (2, 1):
        ops.FORCE_ALGO = algo
        try:
            outs[algo] = ops.linear(x.to(dt), pw)
        finally:
            ops.FORCE_ALGO = 0
    a_, g_ = F.linear(qd(x), qd(w), b).chunk(2, dim=-1)
    ref = a_ * F.gelu(g_)
    tol = 8e-3 if dt == BF16 else 1.5e-3
    assert rel_err(outs[2], ref) < tol and rel_err(outs[1], ref) < tol


@pytest.mark.parametrize("dt", [BF16, torch.float16])
@pytest.mark.parametrize("kind,m,n", [("lin", 16384, 320), ("lin", 16500, 960), ("geglu", 16384, 1280)])
def test_panel_gemm_fused_layernorm(dev, dt, kind, m, n):
    """linear(..., ln=(gamma, beta)) on the panel kernel (the LayerNorm of attention.py:512-514 applied to the resident
    rows in LDS) against sdmi_layernorm + the same GEMM, and against torch on the rounded operands."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    x = (rnd((m, 320), dev, 41) * 1.7 + 0.4).to(dt)
    gamma, beta = 1.0 + 0.2 * rnd((320,), dev, 42), 0.3 * rnd((320,), dev, 43)
    if kind == "geglu":
        w, b = rnd((2 * n, 320), dev, 44, 0.06), rnd((2 * n,), dev, 45)
        pw = ops.pack_geglu(w, b, dtype=dt)
    else:
        w, b = rnd((n, 320), dev, 44, 0.06), rnd((n,), dev, 45)
        pw = ops.pack_linear(w, b, dtype=dt)
    fused = ops.linear(x, pw, ln=(gamma, beta))
    ops.FUSE_LN = False
    try:
        split = ops.linear(x, pw, ln=(gamma, beta))
    finally:
        ops.FUSE_LN = True
    y = qd(F.layer_norm(x.float(), (320,), gamma, beta, 1e-5))
    ref = F.linear(y, qd(w), b)
    if kind == "geglu":
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * F.gelu(g_)
    tol = 8e-3 if dt == BF16 else 1.5e-3
    assert rel_err(fused, ref) < tol and rel_err(split, ref) < tol
    assert rel_err(fused, split.float()) < (4e-3 if dt == BF16 else 6e-4)      # same arithmetic up to the order of the row sums


def test_fused_layernorm_falls_back_and_refuses(dev):
    """Where the panel kernel does not apply, ops.linear runs sdmi_layernorm itself; the C entry point refuses."""
    ops = _ops()
    x = rnd((4096, 640), dev, 46).to(BF16)
    gamma, beta = 1.0 + 0.2 * rnd((640,), dev, 47), 0.3 * rnd((640,), dev, 48)
    w = rnd((640, 640), dev, 49, 0.05)
    pw = ops.pack_linear(w)
    out = ops.linear(x, pw, ln=(gamma, beta))
    ref = F.linear(F.layer_norm(x.float(), (640,), gamma, beta, 1e-5).to(BF16).float(), w.to(BF16).float())
    assert rel_err(out, ref) < 8e-3
    a = ops.hip.GemmArgs()
    o = torch.empty(4096, 640, dtype=BF16, device=dev)
    a.mode, a.A, a.lda, a.M, a.W, a.N, a.K = 0, x.data_ptr(), 640, 4096, pw.w.data_ptr(), 640, 640
    a.out, a.ld_out, a.dtype, a.tile_n, a.alpha, a.batch, a.split_k = o.data_ptr(), 640, 0, pw.tile_n, 1.0, 1, 1
    a.ln_gamma, a.ln_beta, a.ln_eps = gamma.data_ptr(), beta.data_ptr(), 1e-5
    assert ops.hip.lib().sdmi_gemm_panel_applies(C.byref(a)) == 0
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == -1          # SDMI_ERR_BAD_ARG


@pytest.mark.parametrize("kind,m,k,n,res", [
    ("lin", 300, 320, 960, False),          # 128-row tile, ragged rows
    ("lin", 16384, 320, 320, True),         # 256-row tile + residual
    ("lin", 2100, 640, 1920, False),
    ("lin", 512, 1280, 1280, False),        # split-K: the row sums travel with the slabs
    ("lin", 4096, 1280, 3840, False),       # the 256 x 256 x 32 tile, plain epilogue
    ("geglu", 1000, 320, 1280, False),      # tiled GEGLU
    ("geglu", 16384, 640, 2560, False),     # the 256 x 256 x 32 tile, GEGLU epilogue
    ("geglu", 256, 1280, 5120, False),      # split-K + GEGLU
])
def test_layernorm_folded_into_the_projection(dev, kind, m, k, n, res):
    """pack_*(ln=...) with f16 operands: W * gamma is packed, the kernel multiplies the RAW input and applies
    rstd * (acc - mean * rowsum(W')) from the input's row statistics - sdmi_gemm_args.ln_colsum / ln_stats (here from
    sdmi_row_stats, which ops.linear runs when no producer handed them over).  Against torch in fp32 (LayerNorm of
    attention.py:512-514 + the projection), and against the unfolded path (sdmi_layernorm / the panel kernel + the same
    GEMM): the folded form drops the f16 rounding of the normalised operand, so it must not be further from fp32 than the
    unfolded one."""
    ops = _ops()
    qd = lambda t: t.to(F16).float()
    # rows with different means and scales, a residual-stream-like input
    x = (rnd((m, k), dev, 61) * (0.5 + 2.0 * torch.rand(m, 1, device=dev)) + 1.5 * rnd((m, 1), dev, 62)).to(F16)
    gamma, beta = 1.0 + 0.2 * rnd((k,), dev, 63), 0.3 * rnd((k,), dev, 64)
    ln = (gamma, beta, 1e-5)
    r = rnd((m, n), dev, 67).to(F16) if res else None
    if kind == "geglu":
        w, b = rnd((2 * n, k), dev, 65, 0.06), rnd((2 * n,), dev, 66)
        folded, plain = ops.pack_geglu(w, b, dtype=F16, ln=ln), ops.pack_geglu(w, b, dtype=F16)
    else:
        w, b = rnd((n, k), dev, 65, 0.06), rnd((n,), dev, 66)
        folded, plain = ops.pack_linear(w, b, dtype=F16, ln=ln), ops.pack_linear(w, b, dtype=F16)
    assert folded.ln_colsum is not None and plain.ln_colsum is None
    out = ops.linear(x, folded, ln=ln, res=r)
    old = ops.linear(x, plain, ln=ln, res=r)
    ref = F.linear(F.layer_norm(x.float(), (k,), gamma, beta, 1e-5), w, b)
    if kind == "geglu":
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * F.gelu(g_)
    if res:
        ref = ref + r.float()
    e_new, e_old = rel_err(out, ref), rel_err(old, ref)
    assert e_new < 1.5e-3, (e_new, e_old)
    assert e_new < 1.25 * e_old + 1e-4, (e_new, e_old)
    with pytest.raises(ops.hip.SdmiError):
        ops.linear(x, folded)                       # a folded weight needs the caller to say it knows (ln=...)


@pytest.mark.parametrize("m,k,n,res,dup", [
    (300, 320, 320, True, False),          # 128-row tiles, ragged rows: 2 column tiles x 2 wave columns = 4 slots
    (16384, 320, 320, True, True),         # 256-row tiles + the duplicated store of the guidance-pair prefix
    (2100, 2560, 640, True, False),
    (256, 5120, 1280, True, False),        # split-K: the row-wise reduce pass writes one slot
    (8192, 1280, 1280, False, False),      # the 256 x 256 x 32 tile: one slot per 64-column run
])
def test_row_statistics_travel_with_the_token_stream(dev, m, k, n, res, dup):
    """ops.linear(row_stats=True): the launch that writes the residual stream leaves per row and slot the partial
    (sum, sum of squares) of what it stores - sdmi_gemm_args.row_stats_out; the LayerNorm-folded projection that reads the
    stream next takes them instead of normalising (or re-reading) its input.  Checked: the slots add up to the sums of the
    stored rows, and the consumer's output equals the one computed from sdmi_row_stats of the same input."""
    ops = _ops()
    x0 = rnd((m, k), dev, 71).to(F16)
    w0, b0 = rnd((n, k), dev, 72, 0.05), rnd((n,), dev, 73)
    r = (rnd((m, n), dev, 74) * 2.0 + 0.7).to(F16) if res else None
    prod = ops.pack_linear(w0, b0, dtype=F16)
    t, (st, slots) = ops.linear(x0, prod, res=r, dup_rows=dup, row_stats=True)
    plain = ops.linear(x0, prod, res=r, dup_rows=dup)
    assert torch.equal(t, plain)                                   # the statistics change nothing in the output
    rows = 2 * m if dup else m
    assert st.shape == (rows, slots, 2) and t.shape == (rows, n)
    tot = st.double().sum(dim=1)
    tf = t.double()
    # (sums of the fp32 values before their rounding to f16: equal to the stored rows' up to that rounding)
    assert (tot[:, 0] - tf.sum(1)).abs().max() < 2e-3 * tf.abs().sum(1).max()
    assert (tot[:, 1] - (tf * tf).sum(1)).abs().max() < 2e-3 * (tf * tf).sum(1).max()
    if dup:
        assert torch.equal(st[:m], st[m:])
    gamma, beta = 1.0 + 0.2 * rnd((n,), dev, 75), 0.3 * rnd((n,), dev, 76)
    ln = (gamma, beta, 1e-5)
    w1 = rnd((3 * n, n), dev, 77, 0.05)
    cons = ops.pack_linear(w1, dtype=F16, ln=ln)
    out = ops.linear(t, cons, ln=ln, stats=(st, slots))
    out2 = ops.linear(t, cons, ln=ln)                               # statistics recomputed from the stored rows
    ref = F.linear(F.layer_norm(t.float(), (n,), gamma, beta, 1e-5), w1)
    assert rel_err(out, ref) < 1.5e-3 and rel_err(out2, ref) < 1.5e-3
    assert rel_err(out, out2.float()) < 1.2e-3                       # (one f16 rounding step of the largest output)
    with pytest.raises(ops.hip.SdmiError):
        ops.linear(t[:-1], cons, ln=ln, stats=(st, slots))          # statistics of other rows


@pytest.mark.parametrize("nb,n,nk", [(2, 1024, 77), (1, 128, 64), (3, 256, 128), (2, 384, 100), (16, 4096, 77),
                                      (2, 128, 1), (5, 128, 31), (2, 256, 32), (2, 256, 33), (1, 512, 65), (2, 128, 96),
                                      (3, 128, 97)])
def test_cross_attention_block_as_one_launch(dev, nb, n, nk):
    """sdmi_xattn_block: norm2 -> attn2.to_q -> attention over the prompt's keys -> attn2.to_out + residual
    (attention.py:518-528, 300-380) for C = 320 in one kernel, against the three-launch chain of the same ops and
    against torch in fp32 on the f16-rounded operands; with the row statistics a LayerNorm-folded consumer takes."""
    ops = _ops()
    c, heads = 320, 5
    x = (rnd((nb * n, c), dev, 81) * 1.3 + 0.2).to(F16)
    gamma, beta = 1.0 + 0.2 * rnd((c,), dev, 82), 0.2 * rnd((c,), dev, 83)
    ln = (gamma, beta, 1e-5)
    wq, wo, bo = rnd((c, c), dev, 84, 0.06), rnd((c, c), dev, 85, 0.06), rnd((c,), dev, 86)
    kv = rnd((nb * nk, 1000), dev, 87).to(F16)          # K | V as column slices of a wider matrix, as the UNet keeps them
    k, v = kv[:, 40:40 + c], kv[:, 400:400 + c]
    pq, po = ops.pack_linear(wq, dtype=F16), ops.pack_linear(wo, bo, dtype=F16)
    scale = 64 ** -0.5
    assert ops.xattn_block_applies(x, pq, po, k, v, nk, heads, n, policy=False)
    assert ops.xattn_block_applies(x, pq, po, k, v, nk, heads, n) == (ops.XATTN_BLOCK and nb * n >= ops.XATTN_MIN_ROWS)
    out, (st, slots) = ops.xattn_block(x, ln, pq, po, k, v, nk, heads, scale, n, row_stats=True)
    out2 = ops.xattn_block(x, ln, pq, po, k, v, nk, heads, scale, n)
    assert torch.equal(out, out2)
    # the three-launch chain
    qq = ops.linear(x, pq, ln=ln)
    aa = ops.attention_d64(qq, k, v, heads, nb, n, nk, scale)
    chain = ops.linear(aa, po, res=x)
    # torch, fp32 on rounded operands
    xf = x.float()
    qf = F.linear(F.layer_norm(xf, (c,), gamma, beta, 1e-5), wq.to(F16).float())
    qh = qf.reshape(nb, n, heads, 64).permute(0, 2, 1, 3)
    kh = k.float().reshape(nb, nk, heads, 64).permute(0, 2, 1, 3)
    vh = v.float().reshape(nb, nk, heads, 64).permute(0, 2, 1, 3)
    ah = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh
    ref = F.linear(ah.permute(0, 2, 1, 3).reshape(nb * n, c), wo.to(F16).float(), bo) + xf
    assert rel_err(out, ref) < 1.5e-3 and rel_err(chain, ref) < 1.5e-3
    assert rel_err(out, chain.float()) < 1.2e-3
    tot, of = st.double().sum(1), out.double()
    assert slots == 4 and (tot[:, 0] - of.sum(1)).abs().max() < 2e-3 * of.abs().sum(1).max()
    assert (tot[:, 1] - (of * of).sum(1)).abs().max() < 2e-3 * (of * of).sum(1).max()
    assert not ops.xattn_block_applies(x[:-1], pq, po, k, v, nk, heads, n, policy=False)          # ragged panel
    assert not ops.xattn_block_applies(x, pq, po, k, v, 129, heads, n, policy=False)              # too many keys


def test_cross_attention_block_strided_operands_through_the_c_abi(dev):
    """sdmi_xattn_block called as a binding would call it: x / out as column slices of wider matrices (ldx, ld_out != 320),
    no output bias, and out == x (in place: each lane reads exactly the elements it later writes)."""
    import ctypes
    from stablediffusion_amd import hip
    ops = _ops()
    nb, n, nk, c = 2, 256, 77, 320
    xw = (rnd((nb * n, 512), dev, 91) * 0.9).to(F16)
    x = xw[:, 64:64 + c]
    ow = torch.zeros(nb * n, 448, dtype=F16, device=dev)
    gamma, beta = 1.0 + 0.1 * rnd((c,), dev, 92), 0.1 * rnd((c,), dev, 93)
    wq, wo = rnd((c, c), dev, 94, 0.06), rnd((c, c), dev, 95, 0.06)
    kv = rnd((nb * nk, 2 * c), dev, 96).to(F16)
    k, v = kv[:, :c], kv[:, c:]
    pq, po = ops.pack_linear(wq, dtype=F16), ops.pack_linear(wo, dtype=F16)
    ref = ops.xattn_block(x.contiguous(), (gamma, beta, 1e-5), pq, po, k, v, nk, 5, 0.125, n)

    def call(xt, ot):
        a = hip.XattnArgs()
        a.x, a.ldx, a.M, a.rows_per_image, a.C, a.heads = xt.data_ptr(), xt.stride(0), nb * n, n, c, 5
        a.ln_gamma, a.ln_beta, a.ln_eps = gamma.data_ptr(), beta.data_ptr(), 1e-5
        a.wq, a.wo, a.bias_o = pq.w.data_ptr(), po.w.data_ptr(), None
        a.k, a.v, a.ldk, a.ldv, a.nk, a.scale = k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0), nk, 0.125
        a.out, a.ld_out, a.dtype = ot.data_ptr(), ot.stride(0), hip.DTYPE_F16
        assert hip.lib().sdmi_xattn_block_applies(ctypes.byref(a)) == 1
        hip.check(hip.lib().sdmi_xattn_block(ctypes.byref(a), torch.cuda.current_stream().cuda_stream), "sdmi_xattn_block")
        torch.cuda.synchronize()

    call(x, ow[:, 32:32 + c])
    assert torch.equal(ow[:, 32:32 + c], ref)
    assert not bool(ow[:, :32].any()) and not bool(ow[:, 32 + c:].any())          # nothing outside the slice
    xin = xw.clone()
    call(xin[:, 64:64 + c], xin[:, 64:64 + c])                                      # in place
    assert torch.equal(xin[:, 64:64 + c], ref)
    assert torch.equal(xin[:, :64], xw[:, :64]) and torch.equal(xin[:, 64 + c:], xw[:, 64 + c:])


def test_layernorm_fold_is_f16_only_and_checked(dev):
    ops = _ops()
    gamma, beta = 1.0 + 0.2 * rnd((320,), dev, 63), 0.3 * rnd((320,), dev, 64)
    w = rnd((320, 320), dev, 65, 0.06)
    assert ops.pack_linear(w, dtype=BF16, ln=(gamma, beta)).ln_colsum is None       # bf16 operands: the f16 stream is no operand
    pw = ops.pack_linear(w, dtype=F16, ln=(gamma, beta))
    x = rnd((256, 320), dev, 61).to(F16)
    a = ops.hip.GemmArgs()
    o = torch.empty(256, 320, dtype=F16, device=dev)
    a.mode, a.A, a.lda, a.M, a.W, a.N, a.K = 0, x.data_ptr(), 320, 256, pw.w.data_ptr(), 320, 320
    a.out, a.ld_out, a.dtype, a.tile_n, a.alpha, a.batch, a.split_k = o.data_ptr(), 320, 1, pw.tile_n, 1.0, 1, 1
    a.ln_colsum, a.ln_eps = pw.ln_colsum.data_ptr(), 1e-5
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == -1          # no row statistics
    st, slots = ops.row_stats_of(x)
    a.ln_stats, a.ln_stats_slots = st.data_ptr(), slots
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == 0
    a.dtype = 0
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == -1          # bf16 operands
    a.dtype, a.stages = 1, 2
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == -1          # the two-stage variant has no folded instantiation
    a.stages, a.ln_gamma, a.ln_beta = 0, gamma.data_ptr(), beta.data_ptr()
    assert ops.hip.lib().sdmi_gemm(C.byref(a), None) == -1          # folded and unfolded at once
    torch.cuda.synchronize()


def test_panel_gemm_refuses_what_it_is_not_built_for(dev):
    ops = _ops()
    x, pw = rnd((4096, 640), dev, 38).to(BF16), ops.pack_linear(rnd((640, 640), dev, 39))
    ops.FORCE_ALGO = 2
    try:
        with pytest.raises(ops.hip.SdmiError):
            ops.linear(x, pw)                     # K = 640
    finally:
        ops.FORCE_ALGO = 0


LN2, LOG2E = 0.6931471805599453, 1.4426950408889634


@pytest.mark.parametrize("dt", [torch.float16, BF16])
@pytest.mark.parametrize("case", ["random", "spike_late", "spike_early", "all_negative", "ragged"])
def test_attention_lazy_rescale_branches(dev, dt, case):
    """The running maximum only moves when a tile tops it by 2^8 (a rare, wave-uniform branch).  Cases force each
    branch: a key that dominates late (rescale by a large factor mid-sequence) or in the first tile (no later rescale,
    everything else underflows), logits that are all far below zero (the first tile must still anchor the maximum or
    f16 probabilities underflow), ragged Nk; each with the logit scale passed in and with the scale folded into q
    (scale = ln 2: logits arrive in log2 units)."""
    ops = _ops()
    nb, heads, nq, nk = (1, 2, 256, 512) if case != "ragged" else (2, 3, 200, 333)
    c = heads * 64
    qq, kk, vv = rnd((nb * nq, c), dev, 11), rnd((nb * nk, c), dev, 12), rnd((nb * nk, c), dev, 13)
    if case == "spike_late":
        kk[300] *= 12.0
        qq[5, :] = kk[300] / 12.0
    elif case == "spike_early":
        kk[3] *= 12.0
        qq[7, :] = kk[3] / 12.0
    elif case == "all_negative":
        kk[:] = kk[:1] * 3.0 + 0.05 * kk          # all keys nearly parallel ...
        qq[:] = -kk[:1] * 3.0 + 0.05 * qq         # ... and every query anti-parallel to them: logits ~ -70
    qs = (qq * (0.125 * LOG2E)).to(dt)              # the fold happens in fp32, then ONE rounding to the operand type
    out = ops.attention_d64(qs, kk.to(dt), vv.to(dt), heads, nb, nq, nk, LN2)
    ref = _attn_ref(qs.float(), kk.to(dt).float(), vv.to(dt).float(), nb, heads, nq, nk, LN2)
    assert bool(torch.isfinite(out.float()).all())
    assert rel_err(out, ref) < (1e-2 if dt == BF16 else 2e-3), case
    # and the general path (scale passed in, no fold) gives the same attention up to the operand rounding of q
    out2 = ops.attention_d64(qq.to(dt), kk.to(dt), vv.to(dt), heads, nb, nq, nk, 0.125)
    assert rel_err(out2, ref) < (3e-2 if dt == BF16 else 6e-3), case


@pytest.mark.parametrize("dt", [torch.float16, BF16])
@pytest.mark.parametrize("nb,n,d", [(2, 256, 128), (1, 1024, 512), (9, 64, 512), (1, 50, 64), (2, 330, 256), (1, 4096, 512)])
@pytest.mark.parametrize("mode", ["standard", "reference"])
def test_attention_vae(dev, dt, nb, n, d, mode):
    """sdmi_attention_vae (VAE AttnBlock, one head of dim d) against the fp32 formulas of model.py:514-525 on the
    rounded operands: standard = softmax(QK^T) V; reference = W^T V with W row-normalised over the key index.
    Shapes: ragged row blocks / contraction tiles (n = 50, 330), nb > 8 (XCD remap tail), the SD2 shape (4096 x 512);
    q | k | v as column slices of one buffer, as the model passes them."""
    ops = _ops()
    qkv = (rnd((nb * n, 3 * d), dev, 21) * 1.5).to(dt)
    qkv[n // 3, d:2 * d] *= 3.0           # one key with large logits: a peaked row of W
    out = ops.attention_vae(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], nb, n, d ** -0.5, mode)
    f = qkv.float().reshape(nb, n, 3 * d)
    qq, kk, vv = f[..., :d], f[..., d:2 * d], f[..., 2 * d:]
    w = (torch.einsum("bid,bjd->bij", qq, kk) * d ** -0.5).softmax(-1)
    ref = torch.einsum("bij,bjd->bid", w, vv) if mode == "standard" else torch.einsum("bij,bid->bjd", w, vv)
    assert out.shape == (nb * n, d) and bool(torch.isfinite(out.float()).all())
    assert rel_err(out.reshape(nb, n, d), ref) < (1.2e-2 if dt == BF16 else 2e-3)


@pytest.mark.parametrize("dt", [torch.float16, BF16])
@pytest.mark.parametrize("n,d", [(50, 64), (330, 512)])
def test_attention_vae_standard_ragged_tile_with_negative_logits(dev, dt, n, d):
    """Standard mode, N % 32 != 0, every logit of every row strongly negative (q = -k direction, large norm): the columns
    past N in the last contraction tile have ZERO logits, i.e. they sit far ABOVE the row maximum; their weight must still
    be exactly 0 (their V rows are zero), not exp2(-m) / l = inf (inf * 0 = NaN for the whole output row)."""
    ops = _ops()
    base = rnd((n, d), dev, 31)
    base = base / base.norm(dim=-1, keepdim=True)
    shared = torch.ones(d, device=dev) / d ** 0.5
    # q.k = -(a^2) + small: logits around -a^2 * d^-0.5 * ... chosen so that scale * q.k ~ -60 (2^-86 as a weight base)
    a = (60.0 * d ** 0.5) ** 0.5
    q = (-a * shared + 0.3 * base).to(dt)
    k = (a * shared + 0.3 * rnd((n, d), dev, 32) / d ** 0.5).to(dt)
    v = rnd((n, d), dev, 33).to(dt)
    out = ops.attention_vae(q.contiguous(), k.contiguous(), v.contiguous(), 1, n, d ** -0.5, "standard")
    w = ((q.float() @ k.float().T) * d ** -0.5)
    assert float(w.max()) < -30.0           # the case the clamp exists for
    ref = w.softmax(-1) @ v.float()
    assert bool(torch.isfinite(out.float()).all()), "ragged last tile produced inf * 0"
    assert rel_err(out, ref) < (1.2e-2 if dt == BF16 else 2e-3)


@pytest.mark.parametrize("mode", ["standard", "reference"])
@pytest.mark.parametrize("nb,n,d", [(2, 100, 64), (1, 333, 128), (2, 256, 512)])
def test_attention_vae_fp32_mode(dev, nb, n, d, mode):
    """sdmi_f32_attention_vae (the fp32 mode's AttnBlock) against the formulas of model.py:514-525 in float64: ragged
    row blocks and contraction tiles, an odd token count, q | k | v as column slices of one buffer."""
    ops = _ops()
    qkv = rnd((nb * n, 3 * d), dev, 23) * 1.5
    qkv[n // 3, d:2 * d] *= 3.0
    out = ops.attention_vae(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], nb, n, d ** -0.5, mode)
    f = qkv.double().reshape(nb, n, 3 * d)
    qq, kk, vv = f[..., :d], f[..., d:2 * d], f[..., 2 * d:]
    w = (torch.einsum("bid,bjd->bij", qq, kk) * d ** -0.5).softmax(-1)
    ref = torch.einsum("bij,bjd->bid", w, vv) if mode == "standard" else torch.einsum("bij,bid->bjd", w, vv)
    assert out.dtype == torch.float32 and out.shape == (nb * n, d)
    assert rel_err(out.reshape(nb, n, d).double(), ref) < 2e-6


def test_attention_vae_argument_checks(dev):
    import ctypes as C
    from stablediffusion_amd import hip
    lib = hip.lib()
    x = torch.zeros(64, 64, dtype=BF16, device=dev)
    ws = torch.zeros(1024, dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    call = lambda n, d, mode, wsb: lib.sdmi_attention_vae(p(x), p(x), p(x), p(x), 1, n, d, 64, 64, 64, 64, 0.125, mode, 0,
                                                          p(ws), wsb, None)
    assert call(64, 64, 0, 1024) == 0
    assert call(64, 96, 0, 1024) == -1 and call(64, 64, 2, 1024) == -1      # head dim / mode
    assert call(63, 64, 0, 1024) == -2                                        # odd token count
    assert call(64, 64, 0, 100) == -3                                         # workspace
    torch.cuda.synchronize()


def test_softmax_rows_cols(dev):
    ops = _ops()
    s = rnd((2, 300, 512), dev, 1) * 4
    p = ops.softmax_rows(s)
    assert rel_err(p, s.softmax(-1)) < 6e-3
    pt = ops.softmax_cols(s)
    assert rel_err(pt, s.softmax(1)) < 6e-3


# ----------------------------------------------------------------------------- edges


def test_timestep_embedding(dev):
    ops = _ops()
    t = torch.tensor([1, 21, 500, 981], device=dev, dtype=torch.int64)
    out = ops.timestep_embedding(t, 320)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / half)
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert (out.float() - ref).abs().max().item() < 1e-2   # bf16 storage of values in [-1, 1] + fp32 sin/cos of ~1e3 rad


@pytest.mark.parametrize("v_param,cfg", [(False, True), (True, True), (False, False)])
def test_ddim_step(dev, v_param, cfg):
    ops = _ops()
    x = rnd((2, 4, 8, 8), dev, 1)
    mo = rnd((4 if cfg else 2, 4, 8, 8), dev, 2)
    noise = rnd((2, 4, 8, 8), dev, 3)
    a_t, a_prev, sig = 0.5, 0.6, 0.1
    s1m = math.sqrt(1 - a_t)
    xp, p0 = ops.ddim_step(x, mo, a_t, a_prev, sig, s1m, guidance_scale=9.0, cfg=cfg, v_param=v_param, noise=noise,
                           want_pred_x0=True)
    o = mo[:2] + 9.0 * (mo[2:] - mo[:2]) if cfg else mo
    if v_param:
        e_t = math.sqrt(a_t) * o + s1m * x
        pred = math.sqrt(a_t) * x - s1m * o
    else:
        e_t = o
        pred = (x - s1m * e_t) / math.sqrt(a_t)
    ref = math.sqrt(a_prev) * pred + math.sqrt(1 - a_prev - sig ** 2) * e_t + sig * noise
    assert rel_err(xp, ref) < 1e-5
    assert rel_err(p0, pred) < 1e-5


@pytest.mark.parametrize("v_param", [False, True])
def test_ddim_step_full_size_repeated(dev, v_param):
    """The update on a batch-16 768^2 latent (thousands of workgroups, several per CU), thirty launches in a row, every
    element of every launch against the formula: the kernel holds the library's one packed multiply whose second source is
    half-swapped (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]) - a cousin of the packed-fma form that lost products under
    co-resident workgroups; this form has not."""
    ops = _ops()
    x, mo, noise = rnd((16, 4, 96, 96), dev, 1), rnd((32, 4, 96, 96), dev, 2), rnd((16, 4, 96, 96), dev, 3)
    a_t, a_prev, sig = 0.37, 0.52, 0.05
    s1m = math.sqrt(1 - a_t)
    o = mo[:16] + 7.5 * (mo[16:] - mo[:16])
    if v_param:
        e_t, pred = math.sqrt(a_t) * o + s1m * x, math.sqrt(a_t) * x - s1m * o
    else:
        e_t, pred = o, (x - s1m * o) / math.sqrt(a_t)
    ref = math.sqrt(a_prev) * pred + math.sqrt(1 - a_prev - sig ** 2) * e_t + sig * noise
    tol = 2e-5 * ref.abs().max().item()
    for _ in range(30):
        xp, p0 = ops.ddim_step(x, mo, a_t, a_prev, sig, s1m, guidance_scale=7.5, cfg=True, v_param=v_param, noise=noise,
                               want_pred_x0=True)
        assert (xp - ref).abs().max().item() < tol and (p0 - pred).abs().max().item() < 2e-5 * pred.abs().max().item()


def test_image_to_u8_and_cast(dev):
    ops = _ops()
    x = rnd((2, 3, 16, 16), dev, 1)
    out = ops.image_to_u8(x)
    ref = (torch.clamp((x + 1) / 2, 0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert (out.int() - ref.int()).abs().max().item() <= 1
    y = rnd((77 * 1024 + 3,), dev, 2)
    assert torch.equal(ops.to_bf16(y), y.to(BF16))


def test_errors_are_loud(dev):
    """Error behaviour of the boundary: Python exceptions, never a silent fallback."""
    ops = _ops()
    from stablediffusion_amd import hip
    x = rnd((8, 320), dev, 1)
    pw = ops.pack_linear(rnd((64, 320), dev, 2))
    with pytest.raises(hip.SdmiError):
        ops.linear(x, pw)                      # fp32 activations
    with pytest.raises(hip.SdmiError):
        ops.linear(x.cpu().to(BF16), pw)       # host tensor: no CPU fallback
    with pytest.raises(hip.SdmiError):
        ops.linear(rnd((8, 128), dev, 3).to(BF16), pw)   # K mismatch


def test_kernels_are_deterministic(dev):
    """Same inputs, two launches, bit-identical outputs (no float atomics anywhere on the path)."""
    ops = _ops()
    x = rnd((2, 1024, 320), dev, 1).to(BF16)
    gamma, beta = 1 + 0.2 * rnd((320,), dev, 3), 0.1 * rnd((320,), dev, 4)
    assert torch.equal(ops.groupnorm(x, gamma, beta, 1e-5, True), ops.groupnorm(x, gamma, beta, 1e-5, True))
    w = ops.pack_linear(rnd((1280, 11520), dev, 2, 0.01))
    a = rnd((128, 11520), dev, 5).to(BF16)
    assert torch.equal(ops.linear(a, w), ops.linear(a, w))          # split-K path
    qkv = rnd((1024, 3 * 320), dev, 6).to(BF16)
    f = lambda: ops.attention_d64(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 5, 1, 1024, 1024, 0.125)
    assert torch.equal(f(), f())


@pytest.mark.parametrize("dt", [F16])
def test_fp16_operand_type(dev, dt):
    """The same kernels instantiated for f16 operands (the reference's `use_fp16` policy): 8x finer mantissa."""
    ops = _ops()
    qd = lambda t: t.to(dt).float()
    m, n, k = 300, 320, 1280
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    res = rnd((m, n), dev, 4)
    out = ops.linear(x.to(dt), ops.pack_linear(w, b, dtype=dt), res=res.to(dt))
    assert out.dtype == dt
    assert rel_err(out, F.linear(qd(x), qd(w), b) + qd(res)) < 1e-3
    with pytest.raises(Exception):
        ops.linear(x.to(BF16), ops.pack_linear(w, b, dtype=dt))          # mixed operand types are rejected
    xi = rnd((2, 64, 16, 16), dev, 5)
    wt, bias = rnd((128, 64, 3, 3), dev, 6, 1 / 24), rnd((128,), dev, 7)
    out = ops.conv3x3(qd(xi).permute(0, 2, 3, 1).contiguous().to(dt), ops.pack_conv3x3(wt, bias, dtype=dt))
    assert rel_err(out.permute(0, 3, 1, 2), F.conv2d(qd(xi), qd(wt), bias, padding=1)) < 1e-3
    g = rnd((2, 256, 320), dev, 8) * 2 + 0.7
    gamma, beta = 1 + 0.2 * rnd((320,), dev, 9), 0.1 * rnd((320,), dev, 10)
    out = ops.groupnorm(g.to(dt), gamma, beta, 1e-5, True)
    assert rel_err(out.permute(0, 2, 1), F.silu(F.group_norm(qd(g).permute(0, 2, 1), 32, gamma, beta, 1e-5))) < 1e-3
    out = ops.layernorm(g.reshape(512, 320).to(dt), gamma, beta)
    assert rel_err(out, F.layer_norm(qd(g).reshape(512, 320), (320,), gamma, beta, 1e-5)) < 1e-3
    nb, heads, nq, nk = 2, 5, 256, 77
    c = heads * 64
    qq, kk, vv = rnd((nb * nq, c), dev, 11), rnd((nb * nk, c), dev, 12), rnd((nb * nk, c), dev, 13)
    out = ops.attention_d64(qq.to(dt), kk.to(dt), vv.to(dt), heads, nb, nq, nk, 0.125)
    assert rel_err(out, _attn_ref(qd(qq), qd(kk), qd(vv), nb, heads, nq, nk, 0.125)) < 2e-3
    assert torch.equal(ops.cast_f32(x, dt), x.to(dt))
    assert ops.nchw_to_nhwc(rnd((1, 4, 8, 8), dev, 14), dtype=dt).dtype == dt


# ----------------------------------------------------------------------------- fp32 mode


def test_fp32_split3_is_exact(dev):
    """x = x0 + x1 + x2 exactly (three bf16 pieces carry the 24 mantissa bits), planes laid out x0 x0 x1 x0 x1 x2,
    padded channels zero; the two-pointer form equals the concatenation."""
    ops = _ops()
    x1, x2 = rnd((37, 40), dev, 1, 3.0), rnd((37, 24), dev, 2, 1e-3)
    cp = 64
    out = ops.split3(x1, cp, x2).float().reshape(37, 6, cp)
    x = torch.cat([x1, x2], -1)
    assert torch.equal(out[:, 0], out[:, 1]) and torch.equal(out[:, 0], out[:, 3]) and torch.equal(out[:, 2], out[:, 4])
    assert torch.equal((out[:, 0] + out[:, 2] + out[:, 5])[:, :64], x)
    assert torch.equal(out[:, 0, :64], x.to(BF16).float())


def test_fp32_mode_ops(dev):
    """Every op of the fp32 mode against fp32 PyTorch: GEMM family through the split (<= 2e-6 of max|ref|, fp32
    accumulation order aside), norms / attention / activations in plain fp32."""
    ops = _ops()
    f32 = torch.float32
    m, k, n = 300, 320, 640
    x, w, b = rnd((m, k), dev, 1), rnd((n, k), dev, 2, k ** -0.5), rnd((n,), dev, 3)
    res = rnd((m, n), dev, 4)
    out = ops.linear(x, ops.pack_linear(w, b, dtype=f32), res=res)
    assert out.dtype == f32 and rel_err(out, F.linear(x, w, b) + res) < 2e-6
    out = ops.linear(x, ops.pack_linear(w, b, dtype=f32), act=1)
    assert rel_err(out, F.silu(F.linear(x, w, b))) < 2e-6
    x2 = rnd((m, 96), dev, 5)
    w2 = rnd((n, k + 96), dev, 6, 0.05)
    out = ops.linear(x, ops.pack_linear(w2, None, dtype=f32), x2=x2)
    assert rel_err(out, F.linear(torch.cat([x, x2], -1), w2)) < 2e-6
    wg, bg = rnd((8 * k, k), dev, 7, k ** -0.5), rnd((8 * k,), dev, 8)
    out = ops.linear(x, ops.pack_geglu(wg, bg, dtype=f32))
    a, gate = F.linear(x, wg, bg).chunk(2, dim=-1)
    assert rel_err(out, a * F.gelu(gate)) < 2e-6
    for stride, ups in ((1, False), (2, False), (1, True)):
        xi = rnd((2, 40, 12, 10), dev, 9)
        wt, bias = rnd((96, 40, 3, 3), dev, 10, 1 / 19), rnd((96,), dev, 11)
        emb = rnd((2, 96), dev, 12)
        out = ops.conv3x3(xi.permute(0, 2, 3, 1).contiguous(), ops.pack_conv3x3(wt, bias, dtype=f32), stride=stride,
                          upsample=ups, emb=emb)
        xin = F.interpolate(xi, scale_factor=2, mode="nearest") if ups else xi
        ref = F.conv2d(xin, wt, bias, stride=stride, padding=1) + emb[:, :, None, None]
        assert out.dtype == f32 and rel_err(out.permute(0, 3, 1, 2), ref) < 2e-6
    out = ops.conv3x3(xi.permute(0, 2, 3, 1).contiguous(), ops.pack_conv3x3(wt[:4], bias[:4], dtype=f32), out_nchw_f32=True)
    assert rel_err(out, F.conv2d(xi, wt[:4], bias[:4], padding=1)) < 2e-6
    # norms
    g1, g2 = rnd((2, 8, 8, 64), dev, 13, 2.0) + 0.5, rnd((2, 8, 8, 32), dev, 14)
    gam, bet = rnd((96,), dev, 15), rnd((96,), dev, 16)
    out = ops.groupnorm(g1, gam, bet, 1e-5, True, x2=g2)
    ref = F.silu(F.group_norm(torch.cat([g1, g2], -1).permute(0, 3, 1, 2), 32, gam, bet, 1e-5)).permute(0, 2, 3, 1)
    assert rel_err(out, ref) < 2e-6
    xl = rnd((77, 320), dev, 17, 3.0) + 1.0
    gl, bl = rnd((320,), dev, 18), rnd((320,), dev, 19)
    assert rel_err(ops.layernorm(xl, gl, bl, 1e-5), F.layer_norm(xl, (320,), gl, bl, 1e-5)) < 2e-6
    # attention (fused qkv buffer, ragged key count)
    nb, heads, nq, nk = 2, 3, 70, 77
    q = rnd((nb * nq, heads * 64), dev, 20)
    kv = rnd((nb * nk, 2 * heads * 64), dev, 21)
    out = ops.attention_d64(q, kv[:, :heads * 64], kv[:, heads * 64:], heads, nb, nq, nk, 0.125)
    qh = q.reshape(nb, nq, heads, 64).permute(0, 2, 1, 3)
    kh = kv[:, :heads * 64].reshape(nb, nk, heads, 64).permute(0, 2, 1, 3)
    vh = kv[:, heads * 64:].reshape(nb, nk, heads, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).permute(0, 2, 1, 3).reshape(nb * nq, heads * 64)
    assert rel_err(out, ref) < 2e-6
    # layout + timestep embedding
    xn = rnd((2, 4, 6, 5), dev, 22)
    assert torch.equal(ops.nchw_to_nhwc(xn, dtype=f32), xn.permute(0, 2, 3, 1).contiguous())
    t = torch.tensor([0, 17, 981], device=dev)
    te = ops.timestep_embedding(t, 320, dtype=f32)
    freqs = torch.exp(-math.log(10000) * torch.arange(160, device=dev, dtype=f32) / 160)
    args = t[:, None].float() * freqs[None]
    # arguments reach ~1e3 rad: one ulp of the fp32 frequency already moves cos/sin by ~6e-5
    assert rel_err(te, torch.cat([torch.cos(args), torch.sin(args)], -1)) < 2e-4
