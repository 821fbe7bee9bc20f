"""Oracle: fp32 CPU restatement of the UNet and VAE-decoder forward passes.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional and state-dict driven:
every function takes ``sd`` (a dict of tensors keyed with the *reference's* parameter
names, e.g. ``input_blocks.1.0.in_layers.2.weight``) and a key prefix ``p``.  Upstream
(Stability-AI) spellings are accepted as aliases where the reference renamed them
(SURVEY.md §0.5).  Repairs applied to the reference's visible structure are the B#
rows of SURVEY.md §8(a-bugs) and are marked inline.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- helpers

# Operand-precision emulation.  None (default) = the reference's plain fp32 arithmetic.  Set to
# torch.bfloat16 / torch.float16 (use `operand_precision(...)`) to round the two operands of every
# contraction (conv / linear weights and inputs, attention q, k, v and probabilities) to that type
# while still accumulating in fp32 - i.e. exactly what an MFMA with 16-bit operands and fp32
# accumulators computes.  This separates "the kernel is wrong" from "the operand format is coarse".
OPERAND_DTYPE = None


class operand_precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global OPERAND_DTYPE
        self.prev, OPERAND_DTYPE = OPERAND_DTYPE, self.dtype

    def __exit__(self, *exc):
        global OPERAND_DTYPE
        OPERAND_DTYPE = self.prev


def _q(x):
    return x if OPERAND_DTYPE is None else x.to(OPERAND_DTYPE).to(x.dtype)


# Storage-precision emulation (precision studies only; None = off).  The HIP path keeps the residual stream - the
# tensors that are not operands of the next contraction but inputs of a norm or of a later residual add - in a 16-bit
# type; `storage_precision(dtype)` rounds exactly those points (UNet only) so that the cost of that choice can be
# separated from the operand format's.
STORAGE_DTYPE = None


class storage_precision(operand_precision):
    def __enter__(self):
        global STORAGE_DTYPE
        self.prev, STORAGE_DTYPE = STORAGE_DTYPE, self.dtype

    def __exit__(self, *exc):
        global STORAGE_DTYPE
        STORAGE_DTYPE = self.prev


def _s(x):
    return x if STORAGE_DTYPE is None else x.to(STORAGE_DTYPE).to(x.dtype)



def _get(sd, *names):
    for n in names:
        if n in sd:
            return sd[n]
    raise KeyError(names[0])


def _has(sd, *names):
    return any(n in sd for n in names)


def _conv(sd, p, x, stride=1):
    w = sd[p + "weight"]
    return F.conv2d(_q(x), _q(w), sd.get(p + "bias"), stride=stride, padding=w.shape[-1] // 2)


def _lin(sd, p, x):
    return F.linear(_q(x), _q(sd[p + "weight"]), sd.get(p + "bias"))


def _gn(sd, p, x, eps):
    # ldm/utils.py:276-278 (fp32-cast GroupNorm32) and model.py:24-31 / attention.py:45
    return F.group_norm(x.float(), 32, sd[p + "weight"], sd[p + "bias"], eps).type(x.dtype)


# ----------------------------------------------------------------------------- UNet pieces


def time_step_embedding(t, dim, max_period=10000):
    """ldm/utils.py:292-328 with repairs B1 (index ``time_steps``, not the function)
    and B2 (zero-pad only when ``dim`` is odd).  cos first, then sin (utils.py:314)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def res_block(sd, p, x, emb):
    """ResBlock._forward, openai_model.py:262-304 (plain branch: no up/down, no scale-shift)."""
    h = _conv(sd, p + "in_layers.2.", F.silu(_gn(sd, p + "in_layers.0.", x, 1e-5)))
    e = _lin(sd, _pfx(sd, p, "embedding_layers.1.", "emb_layers.1."), F.silu(emb)).type(h.dtype)
    h = _s(h + e[:, :, None, None])
    h = _conv(sd, p + "out_layers.3.", F.silu(_gn(sd, p + "out_layers.0.", h, 1e-5)))
    if p + "skip_connection.weight" in sd:       # openai_model.py:241-248
        x = _s(_conv(sd, p + "skip_connection.", x))
    return _s(x + h)


def _pfx(sd, p, *alts):
    for a in alts:
        if (p + a + "weight") in sd:
            return p + a
    raise KeyError(p + alts[0] + "weight")


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, attention.py:300-380 (mask=None path; fp32 logits)."""
    ctx = x if context is None else context
    q, k, v = _lin(sd, p + "to_q.", x), _lin(sd, p + "to_k.", ctx), _lin(sd, p + "to_v.", ctx)
    b, n, inner = q.shape
    d = inner // heads

    def split(t):  # 'b n (h d) -> (b h) n d', attention.py:315
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

    q, k, v = _q(split(q)), _q(split(k)), _q(split(v))
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * (d ** -0.5)   # attention.py:323-325
    sim = sim.softmax(dim=-1)
    o = torch.einsum("bij,bjd->bid", _q(sim), v)
    o = o.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)   # attention.py:378
    return _lin(sd, p + "to_out.0.", o)


def feed_forward(sd, p, x):
    """FeedForward(glu=True) + GEGLU, attention.py:72-121: value half first, gate second,
    exact (erf) GELU."""
    a, gate = _lin(sd, p + "net.0.proj.", x).chunk(2, dim=-1)
    return _lin(sd, p + "net.2.", a * F.gelu(gate))


def basic_transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:518-528 (disable_self_attn=False)."""
    ff = _pfx(sd, p, "feed_forward.net.2.", "ff.net.2.")[: -len("net.2.")]

    def ln(name, t):
        return F.layer_norm(t, (t.shape[-1],), sd[p + name + ".weight"], sd[p + name + ".bias"], 1e-5)

    x = _s(cross_attention(sd, p + "attn1.", ln("norm1", x), None, heads) + x)
    x = _s(cross_attention(sd, p + "attn2.", ln("norm2", x), context, heads) + x)
    x = _s(feed_forward(sd, ff, ln("norm3", x)) + x)
    return x


def spatial_transformer(sd, p, x, context, heads):
    """SpatialTransformer.forward, attention.py:228-265, with repair B3 (one name for
    project_in / proj_in).  Plain nn.GroupNorm eps 1e-6 (attention.py:45)."""
    b, c, h, w = x.shape
    x_in = x
    pin = _pfx(sd, p, "project_in.", "proj_in.")
    use_linear = sd[pin + "weight"].dim() == 2
    x = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    if not use_linear:
        x = _conv(sd, pin, x)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
    if use_linear:
        x = _lin(sd, pin, x)
    x = _s(x)
    d = 0
    while _has(sd, f"{p}transformer_blocks.{d}.norm1.weight"):
        x = basic_transformer_block(sd, f"{p}transformer_blocks.{d}.", x, context, heads)
        d += 1
    if use_linear:
        x = _lin(sd, p + "proj_out.", x)
    x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = _conv(sd, p + "proj_out.", x)
    return _s(x + x_in)


def down_sample(sd, p, x):
    """DownSample (use_conv=True), openai_model.py:119-148: conv3x3 stride 2 pad 1."""
    return _s(_conv(sd, _pfx(sd, p, "operation.", "op."), x, stride=2))


def up_sample(sd, p, x):
    """UpSample, openai_model.py:67-116: nearest x2 then conv3x3."""
    return _s(_conv(sd, p + "conv.", F.interpolate(x, scale_factor=2, mode="nearest")))


def unet_layout(cfg):
    """Block structure implied by UNetModel.__init__ (openai_model.py:457-938) for
    use_spatial_transformer=True.  Returns (input_blocks, middle, output_blocks) where each
    block is a list of ('conv'|'res'|'st'|'down'|'up', ...) tuples."""
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    attn_res = list(cfg["attention_resolutions"])
    nhc = cfg.get("num_head_channels", -1)
    nh = cfg.get("num_heads", -1)

    def heads_of(ch):  # openai_model.py:686-693
        return ch // nhc if nhc != -1 else nh

    inp = [[("conv",)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb[level]):
            layers = [("res",)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("st", heads_of(ch)))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down",)])
            chans.append(ch)
            ds *= 2
    mid = [("res",), ("st", heads_of(ch)), ("res",)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb[level] + 1):
            chans.pop()
            layers = [("res",)]
            ch = mc * m
            if ds in attn_res:
                layers.append(("st", heads_of(ch)))
            if level and i == nrb[level]:
                layers.append(("up",))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _run_block(sd, p, layers, h, emb, context):
    # TimestepEmbeddingSequential.forward, openai_model.py:46-64
    for j, layer in enumerate(layers):
        q = f"{p}{j}."
        kind = layer[0]
        if kind == "conv":
            h = _s(_conv(sd, q, h))
        elif kind == "res":
            h = res_block(sd, q, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, q, h, context, layer[1])
        elif kind == "down":
            h = down_sample(sd, q, h)
        elif kind == "up":
            h = up_sample(sd, q, h)
    return h


def unet_forward(sd, cfg, x, t, context):
    """UNetModel.forward, openai_model.py:940-1018, with repairs B1/B2 (time embedding)
    and B4 (final conv model_channels -> out_channels)."""
    inp, mid, out = unet_layout(cfg)
    te = _pfx(sd, "", "time_embedding.0.", "time_embed.0.")[:-2]
    emb = time_step_embedding(t, cfg["model_channels"])
    emb = _lin(sd, te + "2.", F.silu(_lin(sd, te + "0.", emb)))
    hs = []
    h = x
    for i, layers in enumerate(inp):
        h = _run_block(sd, f"input_blocks.{i}.", layers, h, emb, context)
        hs.append(h)
    h = _run_block(sd, "middle_block.", mid, h, emb, context)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)          # openai_model.py:1010
        h = _run_block(sd, f"output_blocks.{i}.", layers, h, emb, context)
    h = h.type(x.dtype)
    return _conv(sd, "out.2.", F.silu(_gn(sd, "out.0.", h, 1e-5)))


# ----------------------------------------------------------------------------- VAE decoder


def swish(x):  # model.py:34-40
    return x * torch.sigmoid(x)


def vae_resnet_block(sd, p, x):
    """ResnetBlock.forward, model.py:450-476 with time_embedding=None (model.py:379)."""
    h = _conv(sd, p + "conv1.", swish(_gn(sd, p + "norm1.", x, 1e-6)))
    h = _conv(sd, p + "conv2.", swish(_gn(sd, p + "norm2.", h, 1e-6)))
    if p + "nin_shortcut.weight" in sd:
        x = _conv(sd, p + "nin_shortcut.", x)
    elif p + "conv_shortcut.weight" in sd:
        x = _conv(sd, p + "conv_shortcut.", x)
    return x + h


def vae_attn_block(sd, p, x, mode="reference"):
    """AttnBlock.forward, model.py:496-528.

    mode="reference": literally what the reference computes (model.py:519-523): W is
    softmax-normalised over the key index j but V is contracted against the *query*
    index i:  out[c, j] = sum_i v[c, i] * W[i, j]   (defect B7, kept as the parity default).
    mode="standard": upstream's softmax(QK^T) V."""
    b, c, hh, ww = x.shape
    h_ = _gn(sd, p + "norm.", x, 1e-6)
    q = _conv(sd, p + "q.", h_).reshape(b, c, hh * ww).permute(0, 2, 1)   # b i c
    k = _conv(sd, p + "k.", h_).reshape(b, c, hh * ww)                    # b c j
    v = _conv(sd, p + "v.", h_).reshape(b, c, hh * ww)                    # b c n
    q, k, v = _q(q), _q(k), _q(v)
    w_ = torch.einsum("bic,bcj->bij", q, k) * (int(c) ** (-0.5))
    w_ = _q(F.softmax(w_, dim=-1))
    if mode == "reference":
        o = torch.einsum("bci,bij->bcj", v, w_)
    elif mode == "standard":
        o = torch.einsum("bcj,bij->bci", v, w_)
    else:
        raise ValueError(mode)
    o = _conv(sd, p + "proj_out.", o.reshape(b, c, hh, ww))
    return x + o


def vae_up_sample(sd, p, x):
    """UpSample, model.py:110-142."""
    return _conv(sd, p + "conv.", F.interpolate(x, scale_factor=2.0, mode="nearest"))


def decoder_forward(sd, ddcfg, z, p="", vae_attn="reference"):
    """Decoder.forward, model.py:371-411 with repairs B5 (norm_out is a GroupNorm) and
    B6 (one name for up_sample / upsample).  attn_resolutions is [] for SD2 (yaml:57)."""
    ch_mult = list(ddcfg["ch_mult"])
    nrb = ddcfg["num_res_blocks"]
    h = _conv(sd, p + "conv_in.", z)
    h = vae_resnet_block(sd, p + "mid.block_1.", h)
    h = vae_attn_block(sd, p + "mid.attn_1.", h, vae_attn)
    h = vae_resnet_block(sd, p + "mid.block_2.", h)
    for lvl in reversed(range(len(ch_mult))):
        for i in range(nrb + 1):
            h = vae_resnet_block(sd, f"{p}up.{lvl}.block.{i}.", h)
            if f"{p}up.{lvl}.attn.{i}.norm.weight" in sd:
                h = vae_attn_block(sd, f"{p}up.{lvl}.attn.{i}.", h, vae_attn)
        if lvl != 0:
            up = _pfx(sd, f"{p}up.{lvl}.", "up_sample.conv.", "upsample.conv.")[: -len("conv.")]
            h = vae_up_sample(sd, up, h)
    h = swish(_gn(sd, p + "norm_out.", h, 1e-6))
    h = _conv(sd, p + "conv_out.", h)
    if ddcfg.get("tanh_out", False):
        h = torch.tanh(h)
    return h


def vae_decode(sd, ddcfg, z, vae_attn="reference"):
    """AutoencoderKL.decode (absent from the reference, autoencoder.py:54 only builds
    post_quant_conv): z -> post_quant_conv (1x1) -> decoder."""
    return decoder_forward(sd, ddcfg, _conv(sd, "post_quant_conv.", z), p="decoder.", vae_attn=vae_attn)


# ----------------------------------------------------------------------------- VAE encoder


def vae_down_sample(sd, p, x):
    """DownSample.forward with with_conv=True, model.py:76-106: zero pad (left 0, right 1, top 0, bottom 1), then
    conv3x3 stride 2 without padding."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    w, b = _q(sd[p + "conv.weight"].float()), sd[p + "conv.bias"].float()
    return F.conv2d(_q(x), w, b, stride=2, padding=0)


def encoder_forward(sd, ddcfg, x, p="", vae_attn="reference"):
    """Encoder.forward, model.py:240-279: conv_in -> per level [ResnetBlock (+Attn)] x num_res_blocks, DownSample
    between levels -> mid (Res, Attn, Res) -> GroupNorm -> swish -> conv_out (2*z_channels moments)."""
    ch_mult = list(ddcfg["ch_mult"])
    nrb = ddcfg["num_res_blocks"]
    h = _conv(sd, p + "conv_in.", x)
    for lvl in range(len(ch_mult)):
        for i in range(nrb):
            h = vae_resnet_block(sd, f"{p}down.{lvl}.block.{i}.", h)
            if f"{p}down.{lvl}.attn.{i}.norm.weight" in sd:
                h = vae_attn_block(sd, f"{p}down.{lvl}.attn.{i}.", h, vae_attn)
        if lvl != len(ch_mult) - 1:
            h = vae_down_sample(sd, f"{p}down.{lvl}.downsample.", h)
    h = vae_resnet_block(sd, p + "mid.block_1.", h)
    h = vae_attn_block(sd, p + "mid.attn_1.", h, vae_attn)
    h = vae_resnet_block(sd, p + "mid.block_2.", h)
    h = swish(_gn(sd, p + "norm_out.", h, 1e-6))
    return _conv(sd, p + "conv_out.", h)


def diagonal_gaussian(parameters, noise=None):
    """DiagonalGaussianDistribution, distribution.py:10-43: (mean, log_var clamped to [-30, 20], std, var) and the
    reparameterised sample mean + std * noise (mode = mean)."""
    mean, log_var = torch.chunk(parameters, 2, dim=1)
    log_var = torch.clamp(log_var, -30.0, 20.0)
    std, var = torch.exp(0.5 * log_var), torch.exp(log_var)
    sample = mean + std * noise if noise is not None else None
    return dict(mean=mean, log_var=log_var, std=std, var=var, sample=sample)


def vae_encode(sd, ddcfg, x, vae_attn="reference"):
    """AutoencoderKL.encode (absent from the reference, autoencoder.py builds encoder + quant_conv only):
    x -> encoder -> quant_conv (1x1) -> moments."""
    return _conv(sd, "quant_conv.", encoder_forward(sd, ddcfg, x, p="encoder.", vae_attn=vae_attn))
