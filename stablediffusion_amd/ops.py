"""Tensor-level wrappers over the C ABI (include/sdmi.h): torch supplies device memory and the
stream, the arithmetic happens in libsdmi_hip.so.  No op here has a PyTorch fallback.

Activation convention: 16-bit (bf16 or fp16 - the operand type of the MFMA kernels, chosen by the model's dtype
policy), channels-last.  An image batch is a contiguous [B, H, W, C] tensor,
which *is* the token matrix [B*H*W, C] the transformer blocks use (no permutes anywhere).
"""
import ctypes as C
import os
import functools
import math

import torch

from . import hip

BF16 = torch.bfloat16
F16 = torch.float16
HALF_TYPES = (BF16, F16)
_BN = 128   # default GEMM column tile (weight-row padding); 160 for widths that are multiples of 160
_BK = 64    # K padding (GEMM K tile)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SCRATCH = {}     # (device index, stream handle) -> grow-only byte buffer


def _scratch(nbytes, device):
    """Workspace for one launch sequence (split-K slabs, GroupNorm slab partials, attention row statistics): a static,
    grow-only arena per (device, stream) instead of an allocation per launch - every user writes its workspace before it
    reads it, and launches on one stream are ordered, so consecutive users may share the bytes.  Under stream capture
    the caching allocator's graph pool owns the lifetime, so a plain allocation is made there."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def release_scratch(device=None, stream=None):
    """Drop the scratch arenas (all of them, or those of one device / one stream handle).  They are re-created on
    demand; callers that ran a warm-up on a throw-away stream (graph.py) release that stream's arena here so that it
    does not stay pinned for the life of the process."""
    for key in [k for k in _SCRATCH if (device is None or k[0] == device) and (stream is None or k[1] == stream)]:
        del _SCRATCH[key]


MIXED_BF16_STREAM = True      # False: the bf16 policy stores its residual stream in bf16 as well (rounds 1 - 2; A/B, tests)


def stream_dtype(operand_dtype):
    """Storage type of the residual stream (the tensors carried from block to block: h, the token stream t, the skip
    connections) under an operand type.  bf16 operands keep it in f16: the same bytes, 11 significant bits instead of 8 -
    the CPU emulation (tools/precision_study.py) attributes 1.3-1.5x of the bf16 path's error over its format floor to
    bf16 storage of the stream, none of which is owed to the reference's bf16 policy."""
    return F16 if (operand_dtype == BF16 and MIXED_BF16_STREAM) else operand_dtype


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dt(dtype):
    if dtype == BF16:
        return hip.DTYPE_BF16
    if dtype == F16:
        return hip.DTYPE_F16
    raise hip.SdmiError(f"the HIP kernels take bf16 or fp16 operands, not {dtype}")


def _require(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise hip.SdmiError(f"{name}: the HIP path needs device tensors (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype:
        raise hip.SdmiError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise hip.SdmiError(f"{name}: tensor must be contiguous")


# ----------------------------------------------------------------------------- weight packing


def _round_up(x, m):
    return (x + m - 1) // m * m


class PackedWeight:
    """bf16 [Npad, Kpad] weight (K contiguous) + fp32 bias in the layout sdmi_gemm expects."""

    def __init__(self, w, bias, n_out, k, geglu=False, cin=None, tile_n=_BN, split_cp=0, out_dtype=None, ln_colsum=None,
                 ln_eps=None):
        # 16-bit type of the layer's OUTPUT when it differs from the operand type: the bf16 operand policy keeps its
        # residual stream in f16 (stream_dtype()), so a layer whose output IS the stream (conv_in, the ResBlock convs,
        # to_out, the feed-forward's second linear, proj_out ...) has bf16 weights and an f16 output
        self.out_dtype = out_dtype if out_dtype is not None else w.dtype
        self.split_cp = split_cp  # fp32 mode: > 0 = weights hold six bf16 planes of this width per K group (split3)
        self.w = w              # bf16|fp16 [Npad, Kpad]
        self.tile_n = tile_n    # column tile the rows are padded for (128 | 160)
        self.dtype = w.dtype
        self.bias = bias        # fp32 [n_w] or None
        self.n_out = n_out      # output columns (after gating)
        self.k = k              # padded K
        self.geglu = geglu
        self.cin = cin          # conv3x3: (padded) input channels
        # LayerNorm folded into the projection (sdmi_gemm_args.ln_colsum): `w` holds W * gamma, `bias` holds
        # bias + W @ beta, ln_colsum [Npad] fp32 the row sums of the packed (rounded) weights; linear() then takes the RAW
        # input and the kernel normalises through two numbers per row
        self.ln_colsum = ln_colsum
        self.ln_eps = ln_eps
        self.plain = None       # optional twin without the fold (pack_*(keep_plain=True)): fold_applies() picks per launch


_SPLIT_PAD = 32       # fp32 mode: plane width padding (6 * 32 = 3 GEMM k-tiles, and a conv tap never straddles a k-tile)


def split3(x, cp, x2=None):
    """fp32 [..., C] (optionally the concat [x | x2]) -> bf16 [..., 6*cp] piece planes (sdmi_f32_split3)."""
    _require(x, torch.float32, "x")
    _require(x2, torch.float32, "x2")
    c1 = x.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    rows = x.numel() // c1
    out = torch.empty(*x.shape[:-1], 6 * cp, dtype=BF16, device=x.device)
    hip.check(hip.lib().sdmi_f32_split3(_p(x), _p(x2), c1, c2, rows, _p(out), cp, _stream()), "sdmi_f32_split3")
    return out


# The packers below are thin callers of the C ABI's sdmi_pack_* (csrc/sdmi_pack.hip): a device kernel for weights that
# live on the GPU, the same arithmetic compiled for the host (sdmi_pack_*_host) for CPU tensors.  torch only supplies
# the memory.


def _pack_dt(dtype):
    """(sdmi dtype code, torch dtype of the packed buffer)."""
    if dtype == torch.float32:
        return hip.DTYPE_F32_PLANES, BF16
    return _dt(dtype), dtype


def _src(t):
    """fp32, contiguous, detached view / copy of a weight (no arithmetic)."""
    return t.detach().to(torch.float32).contiguous()


def _pack_alloc(rows_total, kp, tile_n, tdt, device):
    nbytes = hip.lib().sdmi_pack_bytes(rows_total, kp, tile_n)
    out = torch.empty(nbytes // (2 * kp), kp, dtype=tdt, device=device)
    return out


def _pack_rows(ws, out, code, tile_n):
    """Pack the [N_i, K] fp32 matrices `ws` one below the other into `out` (sdmi_pack_linear / _host)."""
    lib = hip.lib()
    rows_total = sum(w.shape[0] for w in ws)
    off = 0
    for w in ws:
        n, k = w.shape
        if w.is_cuda:
            st = lib.sdmi_pack_linear(_p(w), n, k, k, off, rows_total, _p(out), code, tile_n, _stream())
        else:
            st = lib.sdmi_pack_linear_host(_p(w), n, k, k, off, rows_total, _p(out), code, tile_n)
        hip.check(st, "sdmi_pack_linear")
        off += n


# False (env SDMI_FOLD_LN=0): pack_*(ln=...) keeps the plain weights and linear(ln=...) normalises as rounds 1-2 did (tests, A/B)
FOLD_LN = os.environ.get("SDMI_FOLD_LN", "1") != "0"


def _ln_folds(dtype, ln):
    """Whether a projection behind a LayerNorm is packed in the folded form: f16 operands only - the kernels that gather
    the row sums are f16 instantiations, and under bf16 operands the raw input is the f16 stream, not an MFMA operand."""
    return bool(ln is not None and FOLD_LN and FUSE_LN and dtype == F16)


def _fold_ln(w, b, ln):
    """(W * gamma, bias + W @ beta) of a projection behind LayerNorm(gamma, beta): LayerNorm(x) @ W^T + bias =
    rstd * (x @ (W * gamma)^T - mean * rowsum(W * gamma)) + (bias + W @ beta)."""
    gamma, beta = _src(ln[0]).to(w.device), _src(ln[1]).to(w.device)
    extra = (w.double() @ beta.double()).float()
    return (w * gamma[None, :]).contiguous(), (extra if b is None else (b + extra)).contiguous()


def _ln_colsum(packed):
    """Row sums of the PACKED weights (the rounded values the MFMA multiplies, in packed row order, 0 for padding rows)."""
    return packed.double().sum(dim=1).float().contiguous()


# Rows from which a LayerNorm-consuming projection with a plain twin runs UNfolded (tools/bench_lnfold.py, MI355X): the fold
# costs one exposed round trip for the row statistics per workgroup (2.5-3 us in the epilogue), the LayerNorm launch it
# saves costs 11-14 us per layer.  The GEGLU projections (the most workgroup rounds per launch) lose from 4096 rows on
# (16384 x 5120 x 640: 133 us folded, 131 LayerNorm + GEMM; 4096 x 10240 x 1280: 129 / 125), and where the panel kernel
# carries the LayerNorm itself (K = 320, >= 16384 rows) it beats the fold on the 3 x 320 and GEGLU projections (82 against
# 88 us, 164 against 185).  Everything else is faster folded at every size (640 x 640 at 16384 rows: 22 against 34 us).
FOLD_MAX_ROWS_GEGLU = 4096
FOLD_MAX_ROWS_PANEL = 16384


def fold_applies(pw, rows):
    """Whether linear() runs `pw` in its LayerNorm-folded form on `rows` rows (else: its plain twin + the LayerNorm)."""
    if pw.ln_colsum is None:
        return False
    if pw.plain is None:
        return True
    if pw.geglu and rows >= FOLD_MAX_ROWS_GEGLU and not (pw.k == 320 and rows <= 8192):
        # (the level-0 GEGLU projection at UNet batch 2 - 8192 x 2560 x 320 - is below the panel kernel's 16384 rows, so its
        # unfolded form pays a LayerNorm launch: 29.4 + 8.2 us and a launch boundary against 32.2 us folded)
        return False
    if pw.k == 320 and rows >= FOLD_MAX_ROWS_PANEL and (pw.geglu or pw.n_out == 960):
        return False
    return True


def pack_linear(weight, bias=None, dtype=BF16, tile_n=None, out_dtype=None, ln=None, keep_plain=False):
    """nn.Linear / conv1x1 weight [N, K(,1,1)] fp32 -> PackedWeight."""
    return pack_linear_cat([weight], None if bias is None else [bias], dtype=dtype, tile_n=tile_n, out_dtype=out_dtype,
                           ln=ln, keep_plain=keep_plain)


def pack_linear_cat(weights, biases=None, dtype=BF16, tile_n=None, out_dtype=None, ln=None, keep_plain=False):
    """Several [N_i, K] linears sharing an input, fused along N (q|k|v, the 22 emb projections).
    ln = (gamma, beta[, eps]) of the nn.LayerNorm the projection's input goes through: with f16 operands the norm is folded
    into the packed weights (PackedWeight.ln_colsum) and linear() takes the un-normalised input."""
    ws = [_src(x.reshape(x.shape[0], -1)) for x in weights]
    n = sum(w.shape[0] for w in ws)
    k = ws[0].shape[1]
    b = torch.cat([_src(x) for x in biases]).contiguous() if biases is not None else None
    fold = _ln_folds(dtype, ln)
    plain = pack_linear_cat(weights, biases, dtype=dtype, tile_n=tile_n, out_dtype=out_dtype) if fold and keep_plain else None
    if fold:
        wcat, b = _fold_ln(torch.cat(ws), b, ln)
        ws = [wcat]
    code, tdt = _pack_dt(dtype)
    tn = tile_n or hip.lib().sdmi_pack_tile_n(n, 0)
    kp = hip.lib().sdmi_pack_linear_k(k, code)
    out = _pack_alloc(n, kp, tn, tdt, ws[0].device)
    _pack_rows(ws, out, code, tn)
    pw = PackedWeight(out, b, n, kp, tile_n=tn, split_cp=_round_up(k, _SPLIT_PAD) if dtype == torch.float32 else 0,
                      out_dtype=out_dtype if dtype != torch.float32 else None,
                      ln_colsum=_ln_colsum(out) if fold else None, ln_eps=(ln[2] if len(ln) > 2 else 1e-5) if fold else None)
    pw.plain = plain
    return pw


def pack_geglu(weight, bias, dtype=BF16, ln=None, keep_plain=False):
    """GEGLU projection [2*inner, K]: rows [0,inner) = value, [inner,2*inner) = gate (attention.py:89).
    Packed row r holds value (half 0) or gate (half 1) column blk*16+i with blk=r//32, half=(r%32)//16, i=r%16.
    ln: as pack_linear_cat."""
    w = _src(weight)
    inner, k = w.shape[0] // 2, w.shape[1]
    assert inner % 16 == 0
    code, tdt = _pack_dt(dtype)
    kp = hip.lib().sdmi_pack_linear_k(k, code)
    out = _pack_alloc(2 * inner, kp, _BN, tdt, w.device)
    b = _src(bias) if bias is not None else None
    fold = _ln_folds(dtype, ln)
    plain = pack_geglu(weight, bias, dtype=dtype) if fold and keep_plain else None
    if fold:
        w, b = _fold_ln(w, b, ln)
    bo = torch.empty_like(b) if b is not None else None
    if w.is_cuda:
        st = hip.lib().sdmi_pack_geglu(_p(w), _p(b), inner, k, k, _p(out), _p(bo), code, _stream())
    else:
        st = hip.lib().sdmi_pack_geglu_host(_p(w), _p(b), inner, k, k, _p(out), _p(bo), code)
    hip.check(st, "sdmi_pack_geglu")
    pw = PackedWeight(out, bo, inner, kp, geglu=True, split_cp=_round_up(k, _SPLIT_PAD) if dtype == torch.float32 else 0,
                      ln_colsum=_ln_colsum(out) if fold else None, ln_eps=(ln[2] if len(ln) > 2 else 1e-5) if fold else None)
    pw.plain = plain
    return pw


def pack_conv3x3(weight, bias=None, dtype=BF16, tile_n=None, out_dtype=None):
    """Conv2d weight [Cout, Cin, 3, 3] -> [Cout, (ky, kx, ci)] with ci padded to a multiple of 8 (fp32 mode: six planes
    of ci padded to 32 inside every tap)."""
    w = _src(weight)
    cout, cin = w.shape[0], w.shape[1]
    code, tdt = _pack_dt(dtype)
    tn = tile_n or hip.lib().sdmi_pack_tile_n(cout, 0)
    kp = hip.lib().sdmi_pack_conv3x3_k(cin, code)
    cin_p = hip.lib().sdmi_pack_conv3x3_cin(cin, code)
    out = _pack_alloc(cout, kp, tn, tdt, w.device)
    if w.is_cuda:
        st = hip.lib().sdmi_pack_conv3x3(_p(w), cout, cin, _p(out), code, tn, _stream())
    else:
        st = hip.lib().sdmi_pack_conv3x3_host(_p(w), cout, cin, _p(out), code, tn)
    hip.check(st, "sdmi_pack_conv3x3")
    b = _src(bias) if bias is not None else None
    return PackedWeight(out, b, cout, kp, cin=cin_p, tile_n=tn, split_cp=cin_p // 6 if dtype == torch.float32 else 0,
                        out_dtype=out_dtype if dtype != torch.float32 else None)


# ----------------------------------------------------------------------------- GEMM family


FORCE_ALGO = 0        # development aid / tests: sdmi_gemm_args.algo (0 auto, 1 tiled kernel, 2 panel kernel)
FORCE_TILE_M = 0      # development aid (tools/bench_gemm.py): 0 = the library picks the row tile, else 128 | 256
FORCE_WIDE = 0        # development aid: 1 = the 256 x 256 x 32 tile wherever it applies
FORCE_TILE_N = 0      # development aid (tools/bench_gemm.py): 64 = the small tile (64 columns, FORCE_TILE_M = 64 | 128 rows) where it applies
SMALL_TILE = os.environ.get("SDMI_SMALL_TILE", "1") != "0"      # False: _plan_gemm never picks the 64-column tile (A/B, tests)
WIDE_TILE = True      # False: _plan_gemm never picks the 256 x 256 x 32 tile (A/B, tests)

_CUS = 256            # MI355X compute units; the GEMM kernel runs one 8-wave block per CU at a time


def _wide_wins(m, n_w, nk, tile_n):
    """Whether the 256 x 256 x 32 tile (sdmi_gemm_wide.h) beats both row tiles of the tiled kernel, unsplit.  Times in us
    from the measured step costs (tools/bench_gemm.py WIDE=0|1): 0.97 per 32-deep step for the wide tile, 1.12 | 1.07
    per 64-deep step of the 256 x 160 | 256 x 128 tile, 0.67 | 0.63 for the 128-row tiles, and the fixed per-block
    costs those sweeps show for an unsplit launch.  It needs a launch that fills the chip with its own, fewer tiles:
    4096 x 3840 x 1280 55 -> 43 us, 8192^3 1087 -> 990, the decoder's 512- and 256-channel convs +7..9 %; below ~200
    tiles (M = 4096 x N = 1280: 80) it loses to split-K on the smaller tiles and is not considered."""
    tiles_w = math.ceil(m / 256) * math.ceil(n_w / 256)
    if tiles_w < 200 or nk < 16:      # (short K: not measured)
        return False
    t_wide = math.ceil(tiles_w / _CUS) * (2 * nk + 10) * 0.97
    cols = math.ceil(n_w / tile_n)
    t_256 = math.ceil(math.ceil(m / 256) * cols / _CUS) * (nk + 9) * (1.12 if tile_n == 160 else 1.07)
    t_128 = math.ceil(math.ceil(m / 128) * cols / _CUS) * (nk + 7.6) * (0.67 if tile_n == 160 else 0.63)
    return t_wide < 0.95 * min(t_256, t_128)


@functools.lru_cache(maxsize=None)
def _plan_gemm(m, n_w, k, tile_n=_BN, wide_ok=False, geglu=False, small_ok=False, conv=False):
    """(tile_m, split_k, stages, tile_n) from a small occupancy model, in microseconds.  A launch of B blocks puts
    ceil(B/256) blocks on the busiest CU one after the other; a block costs nk/s k-tile steps plus a fixed
    prologue/epilogue; splitting K adds the fp32 slab write + the reduce pass.  Step times fitted to
    tools/bench_gemm.py sweeps on MI355X (predicted vs measured within ~10% on the UNet conv shapes).
    wide_ok: the 256 x 256 x 32 tile (tile_n 256) may be chosen - the weight is padded to 256 rows and the epilogue is
    one it has; _wide_wins decides."""
    nk = k // _BK
    if wide_ok and not geglu and _wide_wins(m, n_w, nk, tile_n):      # (GEGLU: the measured rule below, not the model)
        return (256, 1, 3, 256)
    # GEGLU projections with a short K (levels 0 and 1 of the UNet: K = 320 | 640, many row tiles): half as many tiles,
    # i.e. half as many prologues / epilogues, on the 256 x 256 x 32 tile - measured (tools/bench_gemm.py geglu, WIDE=0|1)
    # 65536 x 2560 x 320 193 -> 170 us, 16384 x 5120 x 640 142 -> 132; at K = 1280 (4096 x 10240) it loses 3 % and stays
    # on the 256 x 128 tile
    if wide_ok and geglu and k <= 640 and math.ceil(m / 256) * math.ceil(n_w / 256) >= 1024:
        return (256, 1, 3, 256)
    # (Rounds 1 - 2 sent short K with many tiles to the two-stage 128-row variant - two blocks per CU, one block's epilogue
    # under the other's main loop.  Re-measured at the end of round 2 the 256-row tile beats it on every such shape:
    # 65536 x 320 x 320 37.4 -> 27.6 us, 16384 x 640 x 640 30.6 -> 23.9, 16384 x 1920 x 640 77 -> 57, GEGLU
    # 16384 x 5120 x 640 141 -> 137; the variant stays reachable through sdmi_gemm_args.stages = 2.)
    best, best_cost = (128, 1, 3, tile_n), None
    for tile_m, step_us in ((256, 1.12 * tile_n / 160), (128, 0.67 * tile_n / 160)):
        tiles = math.ceil(m / tile_m) * math.ceil(n_w / tile_n)
        for s in range(1, 17):
            if s > 1 and nk // s < 2:
                break
            blocks = tiles * s
            cost = math.ceil(blocks / _CUS) * (math.ceil(nk / s) + 4) * step_us
            if blocks < _CUS:
                cost *= 0.8 + 0.2 * blocks / _CUS       # a partly idle chip clocks higher and contends less
            if s > 1:
                cost += (2 * s + 0.5) * m * n_w * 4 / 12.0e6 + 6.0   # slab write+read (mostly L2/MALL) + the extra launch
            if best_cost is None or cost < best_cost * 0.97:
                best, best_cost = (tile_m, s, 3, tile_n), cost
    if small_ok and not conv:
        # The 64-column tile (round 3): 64 x 64 or 128 x 64 blocks for launches whose larger tiles leave most CUs without a
        # block (a few hundred rows: the UNet at batch 1 - 2, the 8 x 8 level at any batch).  Unsplit only - every split
        # measured slower than the unsplit small tile (tools/small_tile_check.py, hipGraph replay, in the units of the
        # costs above = measured - 4.8 us of launch): 64 x 64 costs 0.30 us a k-tile, 128 x 64 0.39, with a far shorter
        # prologue / epilogue than the 160-column tiles; beyond one block per CU the co-resident blocks (three / two per
        # CU) share the CU's fill rate.  512 x 1280 x 1280 18.0 -> 9.7 us, 2048 x 640 x 640 12.7 -> 8.3, 128 x 1280 x 1280
        # 14.4 -> 9.4, 2048 x 640 x 2560 25.0 (split 4) -> 20.4.
        for tile_m, step_us, fixed in ((64, 0.30, -1.0), (128, 0.39, -0.9)):
            rounds = math.ceil(math.ceil(m / tile_m) * math.ceil(n_w / 64) / _CUS)
            cost = max(step_us * nk + fixed, 0.5) + (rounds - 1) * (step_us * nk + 1.5)
            if cost < best_cost * 0.95:
                best, best_cost = (tile_m, 1, 3, 64), cost
    if small_ok and conv and m <= 8192:
        # conv3x3 on the 128 x 64 block (tools/small_conv_check.py, hipGraph replay): 0.35 us a k-tile while every block has
        # a CU to itself, x 1.6 with two co-resident blocks per CU; the split costs are those of the larger tiles.  It wins
        # where the 160-column tiles leave CUs idle even with their best split: 8 x 8 x 1280 at UNet batch 2 (128 rows)
        # 21.8 (split 14) -> 17.5 us (split 12), 64 x 64 x 320 (8192 rows) 33.7 (split 2 + reduce) -> 29.5 unsplit; it loses
        # from 512 rows x 1280 columns on, where the model below keeps the larger tiles.
        tiles = math.ceil(m / 128) * math.ceil(n_w / 64)
        for s in range(1, 25 if m <= 256 else 2):      # (split only where it was measured to win: one or two row tiles)
            if s > 1 and nk // s < 2:
                break
            blocks = tiles * s
            cost = 0.35 * math.ceil(nk / s) * (1.0 if blocks <= _CUS else 0.8 * math.ceil(blocks / _CUS))
            if s > 1:
                cost += (2 * s + 0.5) * m * n_w * 4 / 12.0e6 + 6.0
            if cost < best_cost * 0.95:
                best, best_cost = (128, s, 3, 64), cost
    return best


def _auto_split_k(m, n_w, k, tile_n=_BN):
    return _plan_gemm(m, n_w, k, tile_n)[1]


_AUTO_SPLIT_K = _auto_split_k


# Optional per-launch instrumentation (bench.py's roofline leg): when TRACE is a list, every sdmi_gemm
# launch appends (kind, algorithmic_flops, start_event, end_event) recorded on the launch stream.
TRACE = None


def _gemm_flops(a):
    n_w = a.N * (2 if a.geglu else 1)
    k = 9 * a.Cin if a.mode == 1 else (a.k_valid or a.K)
    return 2 * a.M * n_w * k * max(a.batch, 1)


def _traced(kind, work, shape, launch):
    """Run `launch()`; under TRACE bracket it with HIP events on the launch stream and log (kind, algorithmic work -
    FLOPs for the MFMA families, bytes for the HBM-bound norms -, events, 0, shape)."""
    if TRACE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = launch()
    e1.record()
    TRACE.append((kind, work, e0, e1, 0, shape))
    return r


def _run_gemm(a, what):
    if TRACE is None:
        hip.check(hip.lib().sdmi_gemm(C.byref(a), _stream()), what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(hip.lib().sdmi_gemm(C.byref(a), _stream()), what)
    e1.record()
    TRACE.append(("conv3x3" if a.mode == 1 else "linear", _gemm_flops(a), e0, e1, int(a.split_k),
                  (int(a.M), int(a.N) * (2 if a.geglu else 1), int(a.K), int(a.stride), int(a.upsample),
                   int(a.geglu), 1 if a.res else 0, int(a.out_kind != 0), max(int(a.batch), 1))))


_NARROW_MAX_CIN = 512     # csrc/sdmi_conv_narrow.hip NARROW_MAX_CIN


def _narrow_conv_applies(a, pw):
    """The conditions of sdmi_conv_narrow_launch that the caller controls (csrc/sdmi_conv_narrow.hip): conv3x3, stride 1,
    <= 16 output channels, plain epilogue, Cin a multiple of 32 up to 512."""
    return bool(a.mode == 1 and not FORCE_ALGO and pw.n_out <= 16 and a.stride == 1 and not a.upsample and not a.asym_pad
                and not a.emb and not a.res and not a.act and not pw.geglu and a.Cin % 32 == 0 and a.Cin <= _NARROW_MAX_CIN
                and a.H == a.Ho and a.W_in == a.Wo)


def _fill_common(a, pw, out, bias_on, emb, rows_per_batch, res, act, out_kind, alpha, m, allow_small=True):
    a.W = pw.w.data_ptr()
    a.dtype = _dt(pw.dtype)
    a.bias = pw.bias.data_ptr() if (bias_on and pw.bias is not None) else None
    a.N = pw.n_out
    a.K = pw.k
    a.geglu = 1 if pw.geglu else 0
    a.act = act
    a.out_kind = out_kind
    a.alpha = alpha
    a.batch = 1
    a.out = out.data_ptr()
    if out_kind == 0 and out.dtype != pw.dtype:
        if not (pw.dtype == BF16 and out.dtype == F16):
            raise hip.SdmiError(f"16-bit output {out.dtype} under {pw.dtype} operands")
        a.out_f16 = 1
    if out_kind == 2:
        a.ld_out = 0
    else:
        a.ld_out = out.stride(-2)
    a.rows_per_batch = rows_per_batch or 0
    if emb is not None:
        # may be a column slice of a wider table: only the innermost stride must be 1
        if not emb.is_cuda or emb.dtype != torch.float32 or emb.stride(-1) != 1:
            raise hip.SdmiError("emb: expected a device fp32 matrix with unit inner stride")
        a.emb = emb.data_ptr()
        a.ld_emb = emb.stride(0)
    if res is not None:
        if not pw.split_cp and pw.dtype == BF16 and res.dtype == F16:
            _require(res, F16, "res")            # the f16 residual stream of the bf16 operand policy
            a.res_f16 = 1
        else:
            _require(res, torch.float32 if pw.split_cp else pw.dtype, "res")
        a.res_f32 = 1 if pw.split_cp else 0
        a.res = res.data_ptr()
        a.ld_res = res.shape[-1]
    n_w = pw.n_out * (2 if pw.geglu else 1)
    a.tile_n = pw.tile_n
    a.algo = FORCE_ALGO
    # the 256 x 256 x 32 tile: only where the weight buffer has the rows for it and the library has the epilogue (asked
    # with split_k = 1 - the stricter question - so the answer holds whatever split is pinned below)
    wide_ok = bool(WIDE_TILE and not FORCE_TILE_M and not FORCE_ALGO and pw.w.shape[0] % 256 == 0)
    if wide_ok:
        a.tile_n, a.split_k = 256, 1
        wide_ok = bool(hip.lib().sdmi_gemm_wide_applies(C.byref(a)))
    # the 64-column tile (mode 0, no GEGLU): the packed weight must hold whole 64-row tiles
    small_ok = bool(allow_small and not pw.geglu and not FORCE_ALGO and pw.w.shape[0] >= _round_up(n_w, 64))
    a.tile_m, sk, a.stages, a.tile_n = _plan_gemm(m, n_w, pw.k, pw.tile_n, wide_ok, bool(pw.geglu),
                                                  small_ok and SMALL_TILE and not FORCE_TILE_M, conv=a.mode == 1)
    if FORCE_WIDE and wide_ok:
        a.tile_m, sk, a.tile_n = 256, 1, 256
    if _auto_split_k is not _AUTO_SPLIT_K or a.tile_n not in (256, 64):
        sk = _auto_split_k(m, n_w, pw.k, pw.tile_n)      # (separate hook: tests and tools pin it)
    if FORCE_TILE_M:
        a.tile_m = FORCE_TILE_M & ~1
        a.stages = 2 if FORCE_TILE_M & 1 else 3         # 129 = 128 rows, two stages
    if FORCE_TILE_N == 64 and small_ok:
        a.tile_n, a.stages = 64, 3
        a.tile_m = FORCE_TILE_M if FORCE_TILE_M in (64, 128) else 0
    if a.dup_rows:
        sk = 1                  # the duplicated store lives in the fast 16-bit epilogue, not in the split-K reduce pass
    if _narrow_conv_applies(a, pw):
        # <= 16 output channels: sdmi_gemm takes the byte-bound narrow kernel, which has no split-K form - a split
        # picked for the tiled kernel's occupancy (the UNet `out` conv at batch 2: split 4) would route around it
        sk = 1
    if sk > 1:
        nbytes = hip.lib().sdmi_gemm_workspace_bytes(m, n_w, sk)
        ws = _scratch(nbytes, out.device)
        a.workspace = ws.data_ptr()
        a.workspace_bytes = nbytes
        a.split_k = sk
        return ws
    a.split_k = 1
    return None


FUSE_LN = True      # tests / tools: False runs the LayerNorm of linear(..., ln=...) as its own launch everywhere


def row_stats_of(x):
    """(stats fp32 [rows, 1, 2], 1): per row (sum, sum of squares) of a 16-bit matrix, in the layout a LayerNorm-folded
    projection reads (sdmi_row_stats) - for inputs that no launch with row_stats=True produced."""
    c = x.shape[-1]
    rows = x.numel() // c
    _require(x, x.dtype, "x")
    st = torch.empty(rows, 1, 2, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().sdmi_row_stats(_p(x), _p(st), rows, c, _dt(x.dtype), _stream()), "sdmi_row_stats")
    return st, 1


def linear(x, pw, x2=None, emb=None, rows_per_batch=None, res=None, act=0, out_dtype=None, alpha=1.0, bias=True,
           out_nchw_f32=None, ln=None, dup_rows=False, stats=None, row_stats=False):
    """out[M, N] = epilogue(x[M, K] @ W^T).  x2: second source for the trailing K columns (channel concat).
    ln = (gamma, beta[, eps]): out = epilogue(LayerNorm(x) @ W^T) - fused into the GEMM where the A-panel-stationary
    kernel applies (sdmi_gemm_panel_applies), a sdmi_layernorm launch in front of it elsewhere.
    out_dtype: None = the operand type, or torch.float32.  fp32 mode (pw.split_cp): x (and x2, res) are fp32, the
    input is split into bf16 piece planes first and the output is fp32.
    dup_rows: the output has 2 M rows, rows [M, 2M) a copy of rows [0, M) (sdmi_gemm_args.dup_rows: the guidance-pair
    duplication of the UNet written by the producing launch).
    A weight packed behind a LayerNorm (pack_*(ln=...), f16) takes the RAW input plus its row statistics `stats` =
    (tensor, slots) as a launch with row_stats=True returned them for its output (None: computed here, one more launch).
    row_stats=True: returns (out, stats) - stats = None where the launch cannot produce them (bf16 / fp32 operands)."""
    if dup_rows and (pw.split_cp or out_nchw_f32 is not None or (out_dtype is not None and out_dtype not in HALF_TYPES)):
        raise hip.SdmiError("linear: dup_rows needs a 16-bit row-major output")
    if pw.ln_colsum is not None and not fold_applies(pw, x.numel() // x.shape[-1]):
        pw = pw.plain           # this size runs faster unfolded: the plain twin + the LayerNorm (panel kernel or its own launch)
    if pw.ln_colsum is not None:
        # the LayerNorm lives in the packed weights (pack_*(ln=...)): the kernel takes the raw input
        if ln is None or x2 is not None or emb is not None:
            raise hip.SdmiError("linear: this weight was packed behind a LayerNorm (ln=...): pass ln, no x2, no emb")
        ln = None
    if ln is not None and (pw.split_cp or x2 is not None or not FUSE_LN):
        x, ln = layernorm(x, *ln, out_dtype=None if pw.split_cp else pw.dtype), None
    if pw.split_cp:
        x = split3(x, pw.split_cp, x2)
        x2 = None
        out_dtype = torch.float32
    stream_in = ln is not None and not pw.split_cp and pw.dtype == BF16 and x.dtype == F16
    _require(x, F16 if stream_in else pw.dtype, "x")      # (stream_in: the LayerNorm turns the f16 stream into the bf16 operand)
    out_dtype = out_dtype or pw.out_dtype
    k1 = x.shape[-1]
    m = x.numel() // k1
    a = hip.GemmArgs()
    a.mode = 0
    a.A = x.data_ptr()
    a.lda = k1
    a.M = m
    if pw.ln_colsum is not None:
        if stats is None:
            stats = row_stats_of(x)
        st_t, st_slots = stats
        if st_t.shape[0] != m or st_t.dtype != torch.float32 or not st_t.is_contiguous():
            raise hip.SdmiError("linear: row statistics do not match the input's rows")
        a.ln_colsum, a.ln_eps = pw.ln_colsum.data_ptr(), pw.ln_eps
        a.ln_stats, a.ln_stats_slots = st_t.data_ptr(), st_slots
    if x2 is not None:
        _require(x2, pw.dtype, "x2")
        a.A2 = x2.data_ptr()
        a.lda2 = x2.shape[-1]
        a.k_split = k1
        if k1 + x2.shape[-1] != pw.k:
            raise hip.SdmiError("linear: concat width does not match the packed weight")
    elif k1 != pw.k:
        if _round_up(k1, _BK) != pw.k or k1 % 8:
            raise hip.SdmiError(f"linear: K mismatch ({k1} vs packed {pw.k})")
        a.k_valid = k1          # K padding of the packed weight: columns past k1 read as zero
    if out_nchw_f32 is not None:          # (B, H, W): rows are pixels of B images; write fp32 [B, N, H, W]
        nb_, hh_, ww_ = out_nchw_f32
        out = torch.empty(nb_, pw.n_out, hh_, ww_, dtype=torch.float32, device=x.device)
        keep = _fill_common(a, pw, out, bias, emb, hh_ * ww_, res, act, 2, alpha, m)
    else:
        if dup_rows:
            a.dup_rows = 1
            out = torch.empty(2 * m, pw.n_out, dtype=out_dtype, device=x.device)
        else:
            out = torch.empty(*x.shape[:-1], pw.n_out, dtype=out_dtype, device=x.device)
        # (row statistics on the 64-column tile: one slot per 32 columns, and a consumer reads at most 48)
        keep = _fill_common(a, pw, out, bias, emb, rows_per_batch, res, act, 0 if out_dtype in HALF_TYPES else 1, alpha, m,
                            allow_small=not (row_stats and _round_up(pw.n_out, 64) // 32 > 48))
    out_stats = None
    if row_stats and pw.dtype == F16 and not pw.split_cp and out_nchw_f32 is None and out_dtype == F16 and not pw.geglu \
            and x2 is None and emb is None:
        # the row sums of what this launch stores, for the LayerNorm-folded projection that reads it next
        a.row_stats_slots = hip.lib().sdmi_gemm_row_stats_slots(C.byref(a))
        st_o = torch.empty(out.numel() // pw.n_out, a.row_stats_slots, 2, dtype=torch.float32, device=x.device)
        a.row_stats_out = st_o.data_ptr()
        out_stats = (st_o, int(a.row_stats_slots))
    if ln is not None:
        gamma, beta = ln[0], ln[1]
        _require(gamma, torch.float32, "ln gamma")
        _require(beta, torch.float32, "ln beta")
        if gamma.numel() != k1 or beta.numel() != k1:
            raise hip.SdmiError("linear: LayerNorm parameters do not match K")
        if hip.lib().sdmi_gemm_panel_applies(C.byref(a)):
            a.ln_gamma, a.ln_beta, a.ln_eps = gamma.data_ptr(), beta.data_ptr(), ln[2] if len(ln) > 2 else 1e-5
            a.a_f16 = 1 if stream_in else 0
            if a.tile_n in (256, 64):       # the plan's 256 x 256 x 32 (or 64-column) tile gives way to the kernel that carries the LayerNorm
                a.tile_n, a.tile_m = pw.tile_n, 0
        else:
            xn = layernorm(x, *ln, out_dtype=pw.dtype)
            a.A = xn.data_ptr()
    _run_gemm(a, "sdmi_gemm(linear)")
    del keep
    return (out, out_stats) if row_stats else out


XATTN_BLOCK = os.environ.get("SDMI_XATTN_BLOCK", "1") != "0"      # False: the three-launch form everywhere (A/B, tests)
# One workgroup owns a 128-row panel for the whole chain and fills a CU alone (144 KiB of LDS), so the launch needs rows
# for every CU before it beats three launches that tile finer.  UNet step, same process A/B (tools/ab_flag.py XATTN_BLOCK;
# 5 such blocks a forward): 8192 rows + 0.014 ms, 16384 - 0.003, 32768 - 0.100, 65536 - 0.137.
XATTN_MIN_ROWS = 16384


def xattn_block_applies(x, pw_q, pw_o, k, v, nk, heads, rows_per_image, policy=True):
    """Whether the cross-attention half of a transformer block runs as the one-launch sdmi_xattn_block: f16, C = 320
    (5 heads of 64), plain (unfolded) to_q weight packed for the 160-column tile, a context of <= 128 keys, whole 128-row
    panels per image - and (policy=True) enough rows for it to pay, the switch on."""
    c = x.shape[-1]
    m = x.numel() // c
    if policy and not (XATTN_BLOCK and m >= XATTN_MIN_ROWS):
        return False
    return bool(x.dtype == F16 and c == 320 and heads == 5 and pw_q.dtype == F16 and pw_o.dtype == F16
                and pw_q.ln_colsum is None and not pw_q.geglu and pw_q.n_out == 320 and pw_q.k == 320 and pw_q.tile_n == 160
                and pw_o.n_out == 320 and pw_o.k == 320 and pw_o.tile_n == 160 and pw_o.out_dtype == F16
                and pw_q.bias is None and nk <= 128 and m % 128 == 0 and rows_per_image % 128 == 0
                and k.dtype == F16 and v.dtype == F16 and k.stride(-1) == 1 and v.stride(-1) == 1
                and k.stride(0) % 8 == 0 and v.stride(0) % 8 == 0 and k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0)


def xattn_block(x, ln, pw_q, pw_o, k, v, nk, heads, scale, rows_per_image, row_stats=False):
    """out = to_out(softmax(scale * to_q(LayerNorm(x)) K^T) V) + bias + x in one launch (sdmi_xattn_block; conditions:
    xattn_block_applies).  x: f16 [M, 320] token stream; k, v: f16 [images * nk, >= 320] (column slices allowed).
    row_stats=True: -> (out, statistics of out) for a LayerNorm-folded consumer."""
    _require(x, F16, "x")
    m = x.numel() // 320
    out = torch.empty_like(x)
    a = hip.XattnArgs()
    a.x, a.ldx, a.M, a.rows_per_image, a.C, a.heads = x.data_ptr(), 320, m, rows_per_image, 320, heads
    gamma, beta = ln[0], ln[1]
    _require(gamma, torch.float32, "ln gamma")
    _require(beta, torch.float32, "ln beta")
    a.ln_gamma, a.ln_beta, a.ln_eps = gamma.data_ptr(), beta.data_ptr(), ln[2] if len(ln) > 2 else 1e-5
    a.wq, a.wo = pw_q.w.data_ptr(), pw_o.w.data_ptr()
    a.bias_o = pw_o.bias.data_ptr() if pw_o.bias is not None else None
    a.k, a.v, a.ldk, a.ldv, a.nk = k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0), nk
    a.scale = scale
    a.out, a.ld_out = out.data_ptr(), 320
    a.dtype = hip.DTYPE_F16
    st_o = None
    if row_stats:
        st_o = torch.empty(m, 4, 2, dtype=torch.float32, device=x.device)
        a.row_stats_out, a.row_stats_slots = st_o.data_ptr(), 4
    # (one MFMA family for bench.py's table: the launch is counted with the linears - 2 of its 3 contractions are)
    flops = 2 * m * 320 * 320 * 2 + 4 * m * nk * 320
    _traced("linear", flops, (m, 320, 320, 0, 0, 0, 0, 0, 1), lambda: hip.check(hip.lib().sdmi_xattn_block(C.byref(a), _stream()), "sdmi_xattn_block"))
    return (out, (st_o, 4) if row_stats else None) if row_stats else out


def conv3x3(x, pw, stride=1, upsample=False, emb=None, res=None, out_nchw_f32=False, asym_pad=False, dup_rows=False):
    """3x3 conv, pad 1, on an NHWC bf16 image [B, H, W, Cin]; optional fused nearest-x2 input gather,
    time-embedding add (emb fp32 [B, Cout]) and residual add.  Returns NHWC 16-bit, or fp32 NCHW.
    fp32 mode (pw.split_cp): x and res are fp32 NHWC, the output is fp32 NHWC."""
    if pw.split_cp:
        x = split3(x, pw.split_cp)
    _require(x, pw.dtype, "x")
    b, h, w, cin = x.shape
    if cin != pw.cin:
        raise hip.SdmiError(f"conv3x3: Cin mismatch ({cin} vs packed {pw.cin})")
    hl, wl = (2 * h, 2 * w) if upsample else (h, w)
    padsum = 1 if asym_pad else 2        # asym_pad: zero pad bottom/right only (VAE encoder DownSample)
    ho, wo = (hl + padsum - 3) // stride + 1, (wl + padsum - 3) // stride + 1
    a = hip.GemmArgs()
    a.mode = 1
    a.A = x.data_ptr()
    a.H, a.W_in, a.Cin, a.Ho, a.Wo = h, w, cin, ho, wo
    a.stride = stride
    a.upsample = 1 if upsample else 0
    a.asym_pad = 1 if asym_pad else 0
    m = b * ho * wo
    a.M = m
    if dup_rows:            # images [b, 2b) of the output = a copy of images [0, b) (sdmi_gemm_args.dup_rows)
        if out_nchw_f32 or pw.split_cp:
            raise hip.SdmiError("conv3x3: dup_rows needs a 16-bit NHWC output")
        a.dup_rows = 1
        out = torch.empty(2 * b, ho, wo, pw.n_out, dtype=pw.out_dtype, device=x.device)
    elif out_nchw_f32:
        out = torch.empty(b, pw.n_out, ho, wo, dtype=torch.float32, device=x.device)
    else:
        out = torch.empty(b, ho, wo, pw.n_out, dtype=torch.float32 if pw.split_cp else pw.out_dtype, device=x.device)
    keep = _fill_common(a, pw, out, True, emb, ho * wo, res, 0, 2 if out_nchw_f32 else (1 if pw.split_cp else 0), 1.0, m)
    _run_gemm(a, "sdmi_gemm(conv3x3)")
    del keep
    return out


def gemm_nt_batched(a_mat, b_mat, out_dtype, alpha=1.0, bias_rows=None):
    """Batched out[z] = alpha * a[z] @ b[z]^T (+ bias per ROW).  a: [Z, M, K], b: [Z, N, K] bf16 (Z may be
    1-broadcast for a).  Used for the VAE single-head attention products."""
    if a_mat.dtype not in HALF_TYPES:
        raise hip.SdmiError("gemm_nt_batched: operands must be bf16 or fp16")
    _require(a_mat, a_mat.dtype, "a")
    _require(b_mat, a_mat.dtype, "b")
    z = b_mat.shape[0]
    m, k = a_mat.shape[-2], a_mat.shape[-1]
    n = b_mat.shape[-2]
    if k % _BK or b_mat.shape[-1] != k or n % _BN:
        raise hip.SdmiError("gemm_nt_batched: K must be a multiple of 64 and N a multiple of 128")
    out = torch.empty(z, m, n, dtype=out_dtype, device=a_mat.device)
    g = hip.GemmArgs()
    g.mode = 0
    g.A, g.lda, g.M = a_mat.data_ptr(), k, m
    g.W, g.N, g.K = b_mat.data_ptr(), n, k
    g.out, g.ld_out = out.data_ptr(), n
    g.out_kind = 0 if out_dtype in HALF_TYPES else 1
    g.dtype = _dt(a_mat.dtype)
    g.alpha = alpha
    g.batch = z
    g.split_k = 1
    g.stride_a = 0 if a_mat.shape[0] == 1 else m * k
    g.stride_w = n * k
    g.stride_out = m * n
    if bias_rows is not None:
        _require(bias_rows, torch.float32, "bias_rows")
        g.bias = bias_rows.data_ptr()
        g.bias_per_row = 1
    _run_gemm(g, "sdmi_gemm(batched)")
    return out


# ----------------------------------------------------------------------------- norms


def _norm_dt(in_dtype, out_dtype):
    """sdmi_dtype code of a norm kernel for (input, output) types: same type, or f16 stream -> bf16 operand."""
    if out_dtype is None or out_dtype == in_dtype:
        return _dt(in_dtype)
    if in_dtype == F16 and out_dtype == BF16:
        return hip.DTYPE_F16_TO_BF16
    raise hip.SdmiError(f"norm kernels convert f16 -> bf16 only, not {in_dtype} -> {out_dtype}")


def groupnorm(x1, gamma, beta, eps, silu, x2=None, out_dtype=None):
    """GroupNorm(32)(+SiLU) over NHWC 16-bit (or fp32 in fp32 mode); x2 = second half of a channel concat (read in
    place).  out_dtype: the operand type of the consuming GEMM when it differs from the input's (bf16 policy: the input
    is the f16 residual stream, the output a bf16 operand)."""
    if x1.dtype == torch.float32:
        _require(x1, torch.float32, "x1")
        _require(x2, torch.float32, "x2")
        b, c1 = x1.shape[0], x1.shape[-1]
        c2 = x2.shape[-1] if x2 is not None else 0
        out = torch.empty(*x1.shape[:-1], c1 + c2, dtype=torch.float32, device=x1.device)
        hip.check(hip.lib().sdmi_f32_groupnorm(_p(x1), _p(x2), c1, c2, _p(gamma), _p(beta), _p(out), b,
                                               x1.numel() // (b * c1), eps, 1 if silu else 0, _stream()), "sdmi_f32_groupnorm")
        return out
    dt = _norm_dt(x1.dtype, out_dtype)
    _require(x1, x1.dtype, "x1")
    _require(x2, x1.dtype, "x2")
    _require(gamma, torch.float32, "gamma")
    _require(beta, torch.float32, "beta")
    b = x1.shape[0]
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    hw = x1.numel() // (b * c1)
    out = torch.empty(*x1.shape[:-1], c1 + c2, dtype=out_dtype or x1.dtype, device=x1.device)
    nbytes = hip.lib().sdmi_groupnorm_workspace_bytes(b, hw)
    ws = _scratch(nbytes, x1.device)
    _traced("groupnorm", 2 * out.numel() * out.element_size(), (b, hw, c1 + c2),
            lambda: hip.check(hip.lib().sdmi_groupnorm(_p(x1), _p(x2), c1, c2, _p(gamma), _p(beta), _p(out), b, hw, eps,
                                                       1 if silu else 0, dt, _p(ws), nbytes, _stream()), "sdmi_groupnorm"))
    return out


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=None):
    if x.dtype == torch.float32:
        _require(x, torch.float32, "x")
        out = torch.empty_like(x)
        c = x.shape[-1]
        hip.check(hip.lib().sdmi_f32_layernorm(_p(x), _p(gamma), _p(beta), _p(out), x.numel() // c, c, eps, _stream()),
                  "sdmi_f32_layernorm")
        return out
    dt = _norm_dt(x.dtype, out_dtype)
    _require(x, x.dtype, "x")
    c = x.shape[-1]
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    _traced("layernorm", 2 * out.numel() * out.element_size(), (x.numel() // c, c),
            lambda: hip.check(hip.lib().sdmi_layernorm(_p(x), _p(gamma), _p(beta), _p(out), x.numel() // c, c, eps, dt,
                                                       _stream()), "sdmi_layernorm"))
    return out


# ----------------------------------------------------------------------------- attention


def attention_d64(q, k, v, heads, nb, nq, nk, scale, causal=False):
    """q: [nb*nq, >=heads*64] (may be a column slice of a fused qkv buffer), k/v: [nb*nk, ...].
    Row strides are taken from the tensors; the innermost stride must be 1.  causal (fp32, nq == nk): query i sees
    keys 0..i - the text tower's attn_mask."""
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if t.dtype != q.dtype or not t.is_cuda or t.stride(-1) != 1:
            raise hip.SdmiError(f"attention: bad {name}")
    out = torch.empty(nb * nq, heads * 64, dtype=q.dtype, device=q.device)
    if causal:
        if q.dtype != torch.float32 or nq != nk:
            raise hip.SdmiError("attention: the causal form is fp32 self-attention only")
        hip.check(hip.lib().sdmi_f32_attention_d64_causal(_p(q), _p(k), _p(v), _p(out), nb, heads, nq, q.stride(0),
                                                          k.stride(0), v.stride(0), out.stride(0), scale, _stream()),
                  "sdmi_f32_attention_d64_causal")
        return out
    if q.dtype == torch.float32:
        hip.check(hip.lib().sdmi_f32_attention_d64(_p(q), _p(k), _p(v), _p(out), nb, heads, nq, nk, q.stride(0), k.stride(0),
                                                   v.stride(0), out.stride(0), scale, _stream()), "sdmi_f32_attention_d64")
        return out
    dt = _dt(q.dtype)
    _traced("attention", 4 * nb * heads * nq * nk * 64, (nb, heads, nq, nk),
            lambda: hip.check(hip.lib().sdmi_attention_d64(_p(q), _p(k), _p(v), _p(out), nb, heads, nq, nk, q.stride(0),
                                                           k.stride(0), v.stride(0), out.stride(0), scale, dt, _stream()),
                              "sdmi_attention_d64"))
    return out


def attention_vae(q, k, v, nb, n, scale, mode):
    """Single-head attention over [nb*n, D] 16-bit (or, in fp32 mode, fp32) matrices (row strides from the tensors, inner stride 1):
    mode "standard" = softmax(QK^T) V, "reference" = the reference's contraction over the query index
    (model.py:519-523).  Two launches, no [nb, n, n] tensor (sdmi_attention_vae)."""
    d = q.shape[-1]
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if t.dtype != q.dtype or not t.is_cuda or t.stride(-1) != 1 or t.shape[-1] != d or t.dim() != 2:
            raise hip.SdmiError(f"attention_vae: bad {name}")
    if mode not in ("standard", "reference"):
        raise ValueError(f"vae_attn must be 'reference' or 'standard', got {mode!r}")
    out = torch.empty(nb * n, d, dtype=q.dtype, device=q.device)
    nbytes = hip.lib().sdmi_attention_vae_workspace_bytes(nb, n)
    ws = _scratch(nbytes, q.device)
    if q.dtype == torch.float32:          # fp32 mode of the path
        hip.check(hip.lib().sdmi_f32_attention_vae(_p(q), _p(k), _p(v), _p(out), nb, n, d, q.stride(0), k.stride(0),
                                                   v.stride(0), out.stride(0), scale, 1 if mode == "reference" else 0,
                                                   _p(ws), nbytes, _stream()), "sdmi_f32_attention_vae")
        return out
    dt = _dt(q.dtype)
    hip.check(hip.lib().sdmi_attention_vae(_p(q), _p(k), _p(v), _p(out), nb, n, d, q.stride(0), k.stride(0), v.stride(0),
                                           out.stride(0), scale, 1 if mode == "reference" else 0, dt, _p(ws), nbytes,
                                           _stream()), "sdmi_attention_vae")
    return out


def embed_tokens(ids, table, pos):
    """ids int64 [B, T] -> fp32 [B, T, C] = table[ids] + pos  (token + positional embedding)."""
    _require(table, torch.float32, "table")
    _require(pos, torch.float32, "pos")
    if ids.dtype != torch.int64 or not ids.is_cuda or not ids.is_contiguous() or ids.dim() != 2:
        raise hip.SdmiError("embed_tokens: ids must be a contiguous device int64 [B, T] matrix")
    b, t = ids.shape
    vocab, c = table.shape
    if pos.shape != (t, c):
        raise hip.SdmiError(f"embed_tokens: positional table {tuple(pos.shape)} does not match [T={t}, C={c}]")
    out = torch.empty(b, t, c, dtype=torch.float32, device=ids.device)
    hip.check(hip.lib().sdmi_f32_embed_tokens(_p(ids), _p(table), _p(pos), _p(out), b * t, t, c, vocab, _stream()),
              "sdmi_f32_embed_tokens")
    return out


def softmax_rows(s, out_dtype=BF16):
    """fp32 [R, Ccols] -> 16-bit row softmax."""
    _require(s, torch.float32, "s")
    rows, cols = s.numel() // s.shape[-1], s.shape[-1]
    out = torch.empty(s.shape, dtype=out_dtype, device=s.device)
    hip.check(hip.lib().sdmi_softmax_rows(_p(s), _p(out), rows, cols, cols, cols, _dt(out_dtype), _stream()),
              "sdmi_softmax_rows")
    return out


def softmax_cols(st, out_dtype=BF16):
    """fp32 [Z, R, Ccols] -> 16-bit, softmax over the ROW index of every column."""
    _require(st, torch.float32, "st")
    z, rows, cols = st.shape
    out = torch.empty(st.shape, dtype=out_dtype, device=st.device)
    nbytes = hip.lib().sdmi_softmax_cols_scratch_bytes(z, cols)
    ws = _scratch(nbytes, st.device)
    hip.check(hip.lib().sdmi_softmax_cols(_p(st), _p(out), _p(ws), z, rows, cols, _dt(out_dtype), _stream()),
              "sdmi_softmax_cols")
    return out


# ----------------------------------------------------------------------------- edges


def nchw_to_nhwc(x, cpad=None, scale=1.0, dtype=BF16):
    """fp32 [B, C, H, W] -> 16-bit [B, H, W, Cpad] (fp32 mode: fp32 [B, H, W, C], no padding)."""
    _require(x, torch.float32, "x")
    b, c, h, w = x.shape
    if dtype == torch.float32:
        out = torch.empty(b, h, w, c, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().sdmi_f32_nchw_to_nhwc(_p(x), _p(out), b, c, h * w, scale, _stream()), "sdmi_f32_nchw_to_nhwc")
        return out
    cpad = cpad or _round_up(c, 8)
    out = torch.empty(b, h, w, cpad, dtype=dtype, device=x.device)
    hip.check(hip.lib().sdmi_nchw_f32_to_nhwc(_p(x), _p(out), b, c, cpad, h * w, scale, _dt(dtype), _stream()),
              "sdmi_nchw_f32_to_nhwc")
    return out


def cast_f32(x, dtype=BF16):
    _require(x, torch.float32, "x")
    if dtype == torch.float32:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    hip.check(hip.lib().sdmi_cast_f32(_p(x), _p(out), x.numel(), _dt(dtype), _stream()), "sdmi_cast_f32")
    return out


def to_bf16(x):
    return cast_f32(x, BF16)


def timestep_embedding(t, dim, max_period=10000.0, dtype=BF16):
    _require(t, torch.int64, "t")
    out = torch.empty(t.shape[0], dim, dtype=dtype, device=t.device)
    if dtype == torch.float32:
        hip.check(hip.lib().sdmi_f32_timestep_embedding(_p(t), _p(out), t.shape[0], dim, max_period, _stream()),
                  "sdmi_f32_timestep_embedding")
        return out
    hip.check(hip.lib().sdmi_timestep_embedding(_p(t), _p(out), t.shape[0], dim, max_period, _dt(dtype), _stream()),
              "sdmi_timestep_embedding")
    return out


def ddim_step(x, model_out, a_t, a_prev, sigma_t, sqrt_one_minus_a_t, guidance_scale=1.0, cfg=False, v_param=False,
              noise=None, want_pred_x0=False):
    _require(x, torch.float32, "x")
    _require(model_out, torch.float32, "model_out")
    _require(noise, torch.float32, "noise")
    n = x.numel()
    if model_out.numel() != (2 * n if cfg else n):
        raise hip.SdmiError("ddim_step: model_out has the wrong size")
    x_prev = torch.empty_like(x)
    pred = torch.empty_like(x) if want_pred_x0 else None
    hip.check(hip.lib().sdmi_ddim_step(_p(x), _p(model_out), _p(noise), _p(x_prev), _p(pred), n, 1 if cfg else 0,
                                       guidance_scale, a_t, a_prev, sigma_t, sqrt_one_minus_a_t, 1 if v_param else 0,
                                       _stream()), "sdmi_ddim_step")
    return x_prev, pred


def lincomb(terms):
    """sum_i w_i * x_i over up to four fp32 tensors of one shape: terms = [(w, x), ...]."""
    if not 1 <= len(terms) <= 4:
        raise hip.SdmiError("lincomb takes one to four terms")
    xs = [t[1] for t in terms]
    for x in xs:
        _require(x, torch.float32, "x")
        if x.shape != xs[0].shape:
            raise hip.SdmiError("lincomb: shapes differ")
    out = torch.empty_like(xs[0])
    ptr = [_p(x) for x in xs] + [None] * (4 - len(xs))
    w = [float(t[0]) for t in terms] + [0.0] * (4 - len(terms))
    hip.check(hip.lib().sdmi_lincomb4(*ptr, *w, _p(out), out.numel(), _stream()), "sdmi_lincomb4")
    return out


def diag_gaussian(parameters, deterministic=False):
    """fp32 [B, 2C, h, w] moments -> (mean, log_var clamped to [-30, 20], std, var), each fp32 [B, C, h, w]."""
    _require(parameters, torch.float32, "parameters")
    b, c2, h, w = parameters.shape
    c = c2 // 2
    outs = [torch.empty(b, c, h, w, dtype=torch.float32, device=parameters.device) for _ in range(4)]
    hip.check(hip.lib().sdmi_diag_gaussian(_p(parameters), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), b, c, h * w,
                                           1 if deterministic else 0, _stream()), "sdmi_diag_gaussian")
    return tuple(outs)


def lincomb_mul_add(a, b, c):
    """a + b * c elementwise (fp32): the reparameterised sample mean + std * noise."""
    for t, n in ((a, "a"), (b, "b"), (c, "c")):
        _require(t, torch.float32, n)
    out = torch.empty_like(a)
    hip.check(hip.lib().sdmi_mul_add(_p(a), _p(b), _p(c), _p(out), a.numel(), _stream()), "sdmi_mul_add")
    return out


def image_to_u8(x):
    """fp32 NCHW in [-1, 1] -> uint8 NHWC."""
    _require(x, torch.float32, "x")
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().sdmi_image_to_u8(_p(x), _p(out), b, c, h * w, _stream()), "sdmi_image_to_u8")
    return out
