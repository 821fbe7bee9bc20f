"""UNetModel on the HIP path - host-side mirror of the reference's
ldm/modules/diffusion/openai_model.py (UNetModel :457-1018, ResBlock :151-304, DownSample :119-148,
UpSample :67-116) and ldm/modules/attention.py (SpatialTransformer :170-265, BasicTransformerBlock
:462-537, CrossAttention :268-380, FeedForward/GEGLU :72-121).

Same constructor keywords, same parameter names (state-dict compatible with the reference; upstream
spellings are accepted on load), same ``forward(x, time_steps, context)`` contract.  The module tree
only *holds* the fp32 master parameters; ``forward`` runs a static list of HIP kernel calls over
bf16 NHWC activations and pre-packed bf16 weights:

  * every conv / linear is one ``sdmi_gemm`` launch with its bias, time-embedding add, GEGLU and
    residual add fused in the epilogue; UpSample's nearest-x2 and the skip-connection
    ``torch.cat`` (openai_model.py:1010) are read in place (gather / two pointers);
  * q|k|v projections of self-attention are one GEMM; cross-attention K/V of all 16 transformer
    blocks are one GEMM on the context, cached per context tensor (they depend on the prompt only,
    attention.py:312-313); the 22 ResBlock time-embedding projections are one GEMM per call;
  * NCHW<->(hw)c rearranges of SpatialTransformer (attention.py:249,260) vanish: NHWC *is* [B*HW, C].
"""
import torch
import torch.nn as nn

from . import ops

BF16 = torch.bfloat16


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    """Parameter holder for the reference's fp32-cast GroupNorm (ldm/utils.py:267-278)."""


# ------------------------------------------------------------------ parameter containers (no compute)


class ResBlock(nn.Module):
    def __init__(self, channels, embedding_channels, dropout, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.embedding_layers = nn.Sequential(nn.SiLU(), nn.Linear(embedding_channels, self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)


class DownSample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.operation = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=1)


class UpSample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=1)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim if context_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head, dropout)
        self.feed_forward = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_linear=False):
        super().__init__()
        if context_dim is not None and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim] * depth
        self.in_channels = in_channels
        self.n_heads, self.d_head = n_heads, d_head
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        if use_linear:
            self.project_in = nn.Linear(in_channels, inner)
            self.proj_out = zero_module(nn.Linear(in_channels, inner))   # sic: argument order as in attention.py:224
        else:
            self.project_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = zero_module(nn.Conv2d(inner, in_channels, 1))
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim[d]) for d in range(depth)])


@torch.no_grad()
def pack_spatial_transformer(st, cd, kv_off=0):
    """Packed weights of one SpatialTransformer for UNetModel._transformer: -> (entry, [to_k, to_v weights of its blocks, in
    order], kv_off advanced).  The cross-attention K / V projections are returned unpacked because the UNet concatenates
    those of all its transformers into ONE weight (one GEMM on the context); `kv_off` is this transformer's first column
    in that concatenation.  project_in / proj_out may be nn.Linear (use_linear=True) or 1x1 nn.Conv2d (attention.py:213-226):
    on NHWC activations both are the same [C_out, C_in] GEMM."""
    f32 = lambda t: t.detach().float().contiguous()
    sd = ops.stream_dtype(cd) if cd in ops.HALF_TYPES else None      # the outputs that ARE the residual stream (t, h)
    e = {"norm": (f32(st.norm.weight), f32(st.norm.bias)),
         "pin": ops.pack_linear(st.project_in.weight, st.project_in.bias, dtype=cd, out_dtype=sd),
         # proj_out reads the token stream itself (attention.py:257-262: no norm in front of it): a stream-typed layer
         "pout": ops.pack_linear(st.proj_out.weight, st.proj_out.bias, dtype=sd or cd), "blocks": []}
    kv_w = []
    for tb in st.transformer_blocks:
        c = tb.attn1.to_q.weight.shape[0]
        be = {
            "ln1": (f32(tb.norm1.weight), f32(tb.norm1.bias)),
            "ln2": (f32(tb.norm2.weight), f32(tb.norm2.bias)),
            "ln3": (f32(tb.norm3.weight), f32(tb.norm3.bias)),
            # (ln=: with f16 operands the three LayerNorms are folded into the projections they feed - ops.pack_linear_cat)
            "qkv1": ops.pack_linear_cat([tb.attn1.to_q.weight, tb.attn1.to_k.weight, tb.attn1.to_v.weight], dtype=cd,
                                        ln=(tb.norm1.weight, tb.norm1.bias, tb.norm1.eps), keep_plain=c == 320),
            "o1": ops.pack_linear(tb.attn1.to_out[0].weight, tb.attn1.to_out[0].bias, dtype=cd, out_dtype=sd),
            # (plain twin at 320 channels: ops.xattn_block normalises the token panel itself)
            "q2": ops.pack_linear(tb.attn2.to_q.weight, dtype=cd, ln=(tb.norm2.weight, tb.norm2.bias, tb.norm2.eps),
                                  keep_plain=c == 320),
            "o2": ops.pack_linear(tb.attn2.to_out[0].weight, tb.attn2.to_out[0].bias, dtype=cd, out_dtype=sd),
            "geglu": ops.pack_geglu(tb.feed_forward.net[0].proj.weight, tb.feed_forward.net[0].proj.bias, dtype=cd,
                                    ln=(tb.norm3.weight, tb.norm3.bias, tb.norm3.eps), keep_plain=True),
            "ff2": ops.pack_linear(tb.feed_forward.net[2].weight, tb.feed_forward.net[2].bias, dtype=cd, out_dtype=sd),
            "kv_off": kv_off, "c": c, "heads": tb.attn1.heads, "scale": tb.attn1.scale,
        }
        if tb.attn1.dim_head != 64:
            raise NotImplementedError("sdmi_attention_d64 is built for head dim 64 (SD2: num_head_channels=64)")
        kv_w += [tb.attn2.to_k.weight, tb.attn2.to_v.weight]
        kv_off += 2 * c
        e["blocks"].append(be)
    return e, kv_w, kv_off


def _linear_st(x, pw, want_stats, **kw):
    """ops.linear -> (out, row statistics of out | None)."""
    if want_stats:
        return ops.linear(x, pw, row_stats=True, **kw)
    return ops.linear(x, pw, **kw), None


def run_spatial_transformer(e, h, kv, nk, pair_prefix=False):
    """SpatialTransformer.forward (attention.py:228-265) on an NHWC 16-bit image `h` [B, H, W, C] with the packed entry
    `e` of pack_spatial_transformer; kv = the context's K | V projections ([B * nk, ...], this transformer's columns at
    be["kv_off"]).  pair_prefix: the second half of `h`'s images is a copy of the first (one classifier-free-guidance pair
    per image of the first half, UNetModel._forward_cfg_pairs); everything in front of the first cross-attention - the first
    op that sees the context - runs on the first half only."""
    b, hh, ww, c = h.shape
    n = hh * ww
    h_full = h
    if pair_prefix:
        # `h` arrives with its rows already duplicated (images [b/2, b) are a copy of [0, b/2), written by the launch that
        # produced it: ops.conv3x3(dup_rows=True)); what precedes the first cross-attention works on the first half
        b //= 2
        h = h[:b]
    xn = ops.groupnorm(h, e["norm"][0], e["norm"][1], 1e-6, False, out_dtype=e["pin"].dtype)
    # `st`: the row statistics of the token stream `t`, left by the launch that wrote it (ops.linear(row_stats=True)) for
    # the LayerNorm-folded projection that reads it next (f16 operands; None elsewhere, and ops.linear ignores it)
    # (ops.fold_applies: a projection with a plain twin runs unfolded at the sizes where that is faster)
    blocks = e["blocks"]
    t, st = _linear_st(xn.reshape(b * n, c), e["pin"], bool(blocks) and ops.fold_applies(blocks[0]["qkv1"], b * n))
    for bi, be in enumerate(blocks):
        ci, heads = be["c"], be["heads"]
        qkv = ops.linear(t, be["qkv1"], ln=be["ln1"], stats=st)
        a = ops.attention_d64(qkv[:, :ci], qkv[:, ci:2 * ci], qkv[:, 2 * ci:], heads, b, n, n, be["scale"])
        dup = pair_prefix and bi == 0
        # (dup: the output projection of the first self-attention stores every row twice - the duplication of the
        # guidance pairs where the context enters, without a copy kernel)
        ko = be["kv_off"]
        k2, v2 = kv[:, ko:ko + ci], kv[:, ko + ci:ko + 2 * ci]
        q2p = be["q2"].plain if be["q2"].plain is not None else be["q2"]
        # the cross-attention half as one launch where ops.xattn_block is built for it (f16, 320 channels, <= 128 keys)
        fused = t.dtype == torch.float16 and ops.xattn_block_applies(t if not dup else h_full.reshape(-1, c), q2p, be["o2"],
                                                                     k2, v2, nk, heads, n)
        if dup and t.dtype not in ops.HALF_TYPES:       # fp32 mode (built for the 1e-5 gate, not for speed): a plain copy
            t, st = ops.linear(a, be["o1"], res=t), None
            t = torch.cat([t, t])
        else:
            t, st = _linear_st(a, be["o1"], not fused and ops.fold_applies(be["q2"], b * n * (2 if dup else 1)), res=t,
                               dup_rows=dup)
        if dup:
            h, b = h_full, 2 * b
        if fused:
            want = ops.fold_applies(be["geglu"], b * n)
            t = ops.xattn_block(t, be["ln2"], q2p, be["o2"], k2, v2, nk, heads, be["scale"], n, row_stats=want)
            t, st = t if want else (t, None)
        else:
            q2 = ops.linear(t, be["q2"], ln=be["ln2"], stats=st)
            a = ops.attention_d64(q2, k2, v2, heads, b, n, nk, be["scale"])
            t, st = _linear_st(a, be["o2"], ops.fold_applies(be["geglu"], b * n), res=t)
        g = ops.linear(t, be["geglu"], ln=be["ln3"], stats=st)
        t, st = _linear_st(g, be["ff2"], bi + 1 < len(blocks) and ops.fold_applies(blocks[bi + 1]["qkv1"], b * n), res=t)
    out = ops.linear(t, e["pout"], res=h.reshape(b * n, c))
    return out.reshape(b, hh, ww, c)


class TimestepEmbeddingSequential(nn.Sequential):
    """Container only (openai_model.py:38-64); dispatch happens in UNetModel._run_block."""


# Upstream (Stability-AI) parameter spellings -> the reference's
_UPSTREAM_RENAMES = [
    ("time_embed.", "time_embedding."), ("label_emb.", "label_embedding."), (".emb_layers.", ".embedding_layers."),
    (".op.", ".operation."), (".proj_in.", ".project_in."), (".ff.net.", ".feed_forward.net."),
]


def rename_upstream_keys(state_dict):
    out = {}
    for k, v in state_dict.items():
        for a, b in _UPSTREAM_RENAMES:
            if a in k or k.startswith(a):
                k = k.replace(a, b)
        out[k] = v
    return out


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, use_bf16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_up_sample=-1, use_scale_shift_norm=False, res_block_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, disable_self_attentions=None,
                 num_attention_blocks=None, disable_middle_self_attn=False, use_linear_in_transformer=False,
                 adm_in_channels=None):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None, "use_spatial_transformer needs context_dim"
        if context_dim is not None:
            assert use_spatial_transformer, "context_dim needs use_spatial_transformer"
            if not isinstance(context_dim, (int, list, tuple)):   # OmegaConf ListConfig (openai_model.py:539-542)
                context_dim = list(context_dim)
        if num_heads == -1:
            assert num_head_channels != -1, "set num_heads or num_head_channels"
        if num_head_channels == -1:
            assert num_heads != -1, "set num_heads or num_head_channels"
        unsupported = {
            "dims != 2": dims != 2, "num_classes": num_classes is not None,
            "use_scale_shift_norm": use_scale_shift_norm, "res_block_updown": res_block_updown,
            "not use_spatial_transformer": not use_spatial_transformer, "not conv_resample": not conv_resample,
            "n_embed": n_embed is not None, "disable_self_attentions": disable_self_attentions is not None,
            "num_attention_blocks": num_attention_blocks is not None, "disable_middle_self_attn": disable_middle_self_attn,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"HIP UNetModel: options outside the SD2 hot path are not built: {bad}")

        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = len(channel_mult) * [num_res_blocks] if isinstance(num_res_blocks, int) \
            else list(num_res_blocks)
        if len(self.num_res_blocks) != len(channel_mult):
            raise ValueError("num_res_blocks must be an int or have one entry per channel_mult level")
        self.attention_resolutions = list(attention_resolutions)
        self.dropout = dropout
        self.channel_mult = tuple(channel_mult)
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint      # read by callers (txt2img.py:373); a no-op for inference
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.dtype = torch.bfloat16 if use_bf16 else self.dtype
        # The reference's dtype policy (openai_model.py:602-603) selects the 16-bit MFMA operand / storage type of
        # the HIP kernels (same MFMA rate for both); neither flag = the fp32 mode (fp32 activations, contractions
        # through the exact 3 x bf16 split, csrc/sdmi_f32.hip).
        self.compute_dtype = self.dtype
        self.num_heads, self.num_head_channels = num_heads, num_head_channels

        ted = model_channels * 4
        self.time_embedding = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def make_st(ch):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                                      use_linear=use_linear_in_transformer)

        self.input_blocks = nn.ModuleList(
            [TimestepEmbeddingSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(make_st(ch))
                self.input_blocks.append(TimestepEmbeddingSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbeddingSequential(DownSample(ch, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbeddingSequential(ResBlock(ch, ted, dropout), make_st(ch),
                                                        ResBlock(ch, ted, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(make_st(ch))
                if level and i == self.num_res_blocks[level]:
                    layers.append(UpSample(ch, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbeddingSequential(*layers))
        # repair B4: the final conv maps model_channels -> the constructor's out_channels
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))
        self._pack = None
        self._ctx_cache = None
        self._generation = 0     # bumped whenever the packed weights are dropped (graph.py keys its captures on it)

    # ------------------------------------------------------------------ state dict / packing

    def _drop_packed(self):
        self._pack = None
        self._ctx_cache = None   # cross-attention K/V were projected with the old to_k / to_v
        self._generation = getattr(self, "_generation", 0) + 1

    def load_state_dict(self, state_dict, strict=True, **kw):
        self._drop_packed()
        return super().load_state_dict(rename_upstream_keys(state_dict), strict=strict, **kw)

    def _apply(self, fn, *a, **kw):
        self._drop_packed()      # .cuda() / .to(): packed weights follow the parameters
        return super()._apply(fn, *a, **kw)

    def invalidate_packed(self):
        """Call after mutating parameters in place (packed 16-bit copies are rebuilt on the next forward)."""
        self._drop_packed()

    def _res_blocks(self):
        for blk in list(self.input_blocks) + [self.middle_block] + list(self.output_blocks):
            for layer in blk:
                if isinstance(layer, ResBlock):
                    yield layer

    def _transformers(self):
        for blk in list(self.input_blocks) + [self.middle_block] + list(self.output_blocks):
            for layer in blk:
                if isinstance(layer, SpatialTransformer):
                    yield layer

    @torch.no_grad()
    def _build_pack(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise ops.hip.SdmiError("HIP UNetModel.forward needs the model on a GPU (model.cuda()); "
                                    "there is no CPU fallback")
        # dtype policy of the reference (openai_model.py:602-603): fp16 / bf16 MFMA operands, or - neither flag set -
        # the fp32 mode: fp32 activations, contractions through the 3 x bf16 split (csrc/sdmi_f32.hip)
        cd = self.compute_dtype
        if cd not in ops.HALF_TYPES and cd != torch.float32:
            raise NotImplementedError(f"HIP UNetModel: unsupported compute dtype {cd}")
        pk = {}
        f32 = lambda t: t.detach().float().contiguous()
        # residual-stream storage (h, the skips, the token stream): the operand type, except that bf16 operands keep it
        # in f16 (ops.stream_dtype).  Layers whose OUTPUT is the stream get out_dtype=sd; the few whose INPUT is the raw
        # stream (DownSample / UpSample convs, the 1x1 skip_connection) are packed in the stream type and run as GEMMs
        # of that type - under the bf16 policy they are f16 layers.
        sd = ops.stream_dtype(cd) if cd in ops.HALF_TYPES else None
        te = self.time_embedding
        pk["te0"] = ops.pack_linear(te[0].weight, te[0].bias, dtype=cd)
        pk["te2"] = ops.pack_linear(te[2].weight, te[2].bias, dtype=cd)
        # all ResBlock embedding projections as one [sum Cout, 4*mc] GEMM
        rbs = list(self._res_blocks())
        pk["emb_all"] = ops.pack_linear_cat([r.embedding_layers[1].weight for r in rbs],
                                            [r.embedding_layers[1].bias for r in rbs], dtype=cd)
        off = 0
        for r in rbs:
            e = {"emb_off": off}
            off += r.out_channels
            e["gn1"] = (f32(r.in_layers[0].weight), f32(r.in_layers[0].bias))
            e["conv1"] = ops.pack_conv3x3(r.in_layers[2].weight, r.in_layers[2].bias, dtype=cd, out_dtype=sd)
            e["gn2"] = (f32(r.out_layers[0].weight), f32(r.out_layers[0].bias))
            e["conv2"] = ops.pack_conv3x3(r.out_layers[3].weight, r.out_layers[3].bias, dtype=cd, out_dtype=sd)
            if isinstance(r.skip_connection, nn.Conv2d):
                if r.skip_connection.kernel_size != (1, 1):
                    raise NotImplementedError("3x3 skip_connection (use_conv=True) is outside the SD2 path")
                e["skip"] = ops.pack_linear(r.skip_connection.weight, r.skip_connection.bias, dtype=sd or cd)
            pk[id(r)] = e
        # cross-attention K/V of every transformer block, one GEMM on the context
        kv_w = []
        kv_off = 0
        for st in self._transformers():
            pk[id(st)], w, kv_off = pack_spatial_transformer(st, cd, kv_off)
            kv_w += w
        pk["kv_all"] = ops.pack_linear_cat(kv_w, dtype=cd)
        for blk in list(self.input_blocks) + list(self.output_blocks):
            for layer in blk:
                if isinstance(layer, nn.Conv2d):          # conv_in: operand = the converted latents, output = the stream
                    pk[id(layer)] = ops.pack_conv3x3(layer.weight, layer.bias, dtype=cd, out_dtype=sd)
                elif isinstance(layer, DownSample):
                    pk[id(layer)] = ops.pack_conv3x3(layer.operation.weight, layer.operation.bias, dtype=sd or cd)
                elif isinstance(layer, UpSample):
                    pk[id(layer)] = ops.pack_conv3x3(layer.conv.weight, layer.conv.bias, dtype=sd or cd)
        pk["out_gn"] = (f32(self.out[0].weight), f32(self.out[0].bias))
        pk["out_conv"] = ops.pack_conv3x3(self.out[2].weight, self.out[2].bias, dtype=cd)
        self._pack = pk
        return pk

    # ------------------------------------------------------------------ forward pieces

    def _context_kv(self, pk, context):
        key = (context.data_ptr(), tuple(context.shape), context._version, context.dtype)
        if self._ctx_cache is not None and self._ctx_cache[0] == key:
            return self._ctx_cache[1]
        cd = self.compute_dtype
        ctx = context if context.dtype == cd else ops.cast_f32(context.float().contiguous(), cd)
        kv = ops.linear(ctx.reshape(-1, ctx.shape[-1]), pk["kv_all"])     # [B*Nk, sum 2C]
        self._ctx_cache = (key, kv, context)   # keep `context` alive so data_ptr stays unique
        return kv

    def refresh_context_kv(self, context, kv):
        """The cross-attention K / V of `context` written INTO the existing tensor `kv` (shape and type of what
        _context_kv returns), which becomes the cache entry for `context`: a captured hipGraph that holds pointers into `kv`
        (graph.py) serves a new prompt without a re-capture - one projection launch and one copy instead of three forwards."""
        pk = self._pack or self._build_pack()
        self._ctx_cache = None
        new = self._context_kv(pk, context)
        if new.shape != kv.shape or new.dtype != kv.dtype:
            raise ValueError("refresh_context_kv: the context's K/V do not have the shape of the tensor to refresh")
        kv.copy_(new)
        self._ctx_cache = ((context.data_ptr(), tuple(context.shape), context._version, context.dtype), kv, context)

    def _res_block(self, pk, r, h, skip, emb_all, dup_rows=False):
        e = pk[id(r)]
        emb = emb_all[:, e["emb_off"]:e["emb_off"] + r.out_channels]
        n1 = ops.groupnorm(h, e["gn1"][0], e["gn1"][1], 1e-5, True, x2=skip, out_dtype=e["conv1"].dtype)
        h1 = ops.conv3x3(n1, e["conv1"], emb=emb)
        n2 = ops.groupnorm(h1, e["gn2"][0], e["gn2"][1], 1e-5, True, out_dtype=e["conv2"].dtype)
        if "skip" in e:
            sk = ops.linear(h, e["skip"], x2=skip)
        else:
            assert skip is None
            sk = h
        return ops.conv3x3(n2, e["conv2"], res=sk, dup_rows=dup_rows)

    def _transformer(self, pk, st, h, kv, nk, pair_prefix=False):
        return run_spatial_transformer(pk[id(st)], h, kv, nk, pair_prefix)

    def _run_block(self, pk, blk, h, skip, emb_all, kv, nk):
        for layer in blk:
            if isinstance(layer, ResBlock):
                h = self._res_block(pk, layer, h, skip, emb_all)
                skip = None
            elif isinstance(layer, SpatialTransformer):
                h = self._transformer(pk, layer, h, kv, nk)
            elif isinstance(layer, DownSample):
                h = ops.conv3x3(h, pk[id(layer)], stride=2)
            elif isinstance(layer, UpSample):
                h = ops.conv3x3(h, pk[id(layer)], upsample=True)
            elif isinstance(layer, nn.Conv2d):
                h = ops.conv3x3(h, pk[id(layer)])
            else:
                raise NotImplementedError(type(layer))
        return h

    @torch.no_grad()
    def _cfg_prefix_blocks(self):
        """The blocks in front of the first cross-attention, if the model starts the way SD's does: conv_in, then
        [ResBlock, SpatialTransformer]."""
        if len(self.input_blocks) < 2:
            return None
        b0, b1 = list(self.input_blocks[0]), list(self.input_blocks[1])
        if len(b0) == 1 and isinstance(b0[0], nn.Conv2d) and len(b1) == 2 and isinstance(b1[0], ResBlock) and \
                isinstance(b1[1], SpatialTransformer):
            return b0, b1
        return None

    def _forward_cfg_pairs(self, pk, x, emb_all, kv, nk):
        """The first two input blocks for a classifier-free-guidance batch x = cat([z, z]), t = cat([t, t]),
        context = cat([uc, c]) (what DDIMSampler / PLMSSampler build, ddim.py:184-186 upstream): rows b and b + B/2 are the
        same latent at the same time step and differ in the context only, so everything in front of the first
        cross-attention - conv_in, the first ResBlock, GroupNorm + proj_in, the first self-attention and its output
        projection - is the same computation twice.  It runs once, on B/2 rows, and is duplicated where the context
        enters.  Exact: the duplicated rows are what the full batch would have computed, up to the fp32 summation order
        of a GEMM plan that depends on the row count.  Returns (h, [skip tensors of the two blocks]) at full batch."""
        b0, b1 = self._cfg_prefix_blocks()
        half = x.shape[0] // 2
        emb_h = emb_all[:half]
        h = ops.nchw_to_nhwc(x[:half].float().contiguous(), dtype=self.compute_dtype)
        # conv_in and the ResBlock's last conv store every image twice (sdmi_gemm_args.dup_rows): the full-batch tensors the
        # skip connection and the transformer's residual need, without a copy kernel; the prefix itself reads the first half
        if self.compute_dtype in ops.HALF_TYPES:
            h0 = ops.conv3x3(h, pk[id(b0[0])], dup_rows=True)
            r = self._res_block(pk, b1[0], h0[:half], None, emb_h, dup_rows=True)
        else:                                           # fp32 mode: plain copies
            h0 = ops.conv3x3(h, pk[id(b0[0])])
            r = self._res_block(pk, b1[0], h0, None, emb_h)
            h0, r = torch.cat([h0, h0]), torch.cat([r, r])
        h1 = self._transformer(pk, b1[1], r, kv, nk, pair_prefix=True)
        return h1, [h0, h1]

    @torch.no_grad()
    def _time_path(self, pk, t):
        """int64 [n] -> fp32 [n, sum C_out]: sinusoid -> Linear -> SiLU -> Linear -> (SiLU, shared by all ResBlocks) ->
        the 22 embedding projections as one GEMM (utils.py:292-328; openai_model.py:956-964, 280)."""
        te = ops.timestep_embedding(t, self.model_channels, dtype=self.compute_dtype)
        e1 = ops.linear(te, pk["te0"], act=1)
        e2 = ops.linear(e1, pk["te2"], act=1)
        return ops.linear(e2, pk["emb_all"], out_dtype=torch.float32)

    @torch.no_grad()
    def time_embedding_table(self, time_steps):
        """The whole timestep path for a list of steps at once: int64 [S] -> fp32 [S, sum C_out], row s = what forward()
        computes for time_steps == time_steps[s].  It depends on t only (openai_model.py:956-964, 280), and a sampler knows
        all S steps of its schedule up front: it builds the table once per schedule and hands forward() one row per step
        (`time_emb`), which removes the four launches of the time path from every step."""
        pk = self._pack or self._build_pack()
        dev = next(self.parameters()).device
        t = torch.as_tensor(time_steps).to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
        return self._time_path(pk, t)

    @torch.no_grad()
    def forward(self, x, time_steps=None, context=None, label=None, timesteps=None, y=None, cfg_pairs=False,
                time_emb=None, **kwargs):
        """x [B, C, h, w] (fp32/fp16/bf16 NCHW), time_steps int64 [B], context [B, Nk, context_dim]
        -> eps / v prediction, same shape and dtype as x (openai_model.py:940-1018).
        cfg_pairs=True is the caller's promise that x[B/2:] == x[:B/2] and time_steps[B/2:] == time_steps[:B/2] (a
        classifier-free-guidance batch): the part of the network in front of the first cross-attention then runs on
        B/2 rows (_forward_cfg_pairs).  The promise covers x and time_steps only; it is ignored when a label is given.
        time_emb: the row of time_embedding_table() that belongs to time_steps (all rows of the batch at the same step), or
        [B, sum C_out] rows - the time path is then not run."""
        time_steps = time_steps if time_steps is not None else timesteps
        label = label if label is not None else y
        assert (label is not None) == (self.num_classes is not None), \
            "must specify label if and only if the model is class-conditional"
        assert time_steps is not None and context is not None
        pk = self._pack or self._build_pack()
        bsz = x.shape[0]
        t = time_steps.to(device=x.device, dtype=torch.int64).contiguous()
        if t.dim() == 0:
            t = t.expand(bsz).contiguous()
        if time_emb is not None:
            # a row (or B rows) of time_embedding_table(): the caller's statement of what the time path yields for
            # `time_steps`.  One row serves the whole batch through a zero batch stride (the conv epilogue reads
            # emb[b * ld_emb + c]).
            if not time_emb.is_cuda or time_emb.dtype != torch.float32 or time_emb.shape[-1] != pk["emb_all"].n_out:
                raise ops.hip.SdmiError("time_emb: expected a device fp32 row of time_embedding_table()")
            emb_all = time_emb.unsqueeze(0).expand(bsz, -1) if time_emb.dim() == 1 else time_emb
            if emb_all.shape[0] != bsz or emb_all.stride(-1) != 1:
                raise ops.hip.SdmiError("time_emb: expected [sum C_out] or [B, sum C_out] with unit inner stride")
        else:
            emb_all = self._time_path(pk, t)
        kv = self._context_kv(pk, context)
        nk = context.shape[1]

        # (label is None: with class / adm conditioning the two halves of a guidance pair may carry different y, so their
        # time + label embeddings differ and the shared prefix would be wrong for rows B/2 onwards)
        if cfg_pairs and label is None and bsz % 2 == 0 and bsz >= 2 and self._cfg_prefix_blocks() is not None:
            h, hs = self._forward_cfg_pairs(pk, x, emb_all, kv, nk)
            rest = list(self.input_blocks)[2:]
        else:
            h = ops.nchw_to_nhwc(x.float().contiguous(), dtype=self.compute_dtype)
            hs, rest = [], self.input_blocks
        for blk in rest:
            h = self._run_block(pk, blk, h, None, emb_all, kv, nk)
            hs.append(h)
        h = self._run_block(pk, self.middle_block, h, None, emb_all, kv, nk)
        for blk in self.output_blocks:
            h = self._run_block(pk, blk, h, hs.pop(), emb_all, kv, nk)
        n = ops.groupnorm(h, pk["out_gn"][0], pk["out_gn"][1], 1e-5, True, out_dtype=pk["out_conv"].dtype)
        out = ops.conv3x3(n, pk["out_conv"], out_nchw_f32=True)
        return out.type(x.dtype)
