"""AutoencoderKL decode path on the HIP kernels - host-side mirror of the reference's
ldm/models/autoencoder.py (AutoencoderKL :14-93) and ldm/modules/diffusion/model.py (Decoder :282-411,
ResnetBlock :414-476, AttnBlock :479-528, UpSample :110-142).

Constructor keywords accept both the reference's spellings (``dd_config/loss_config/embedding_dim``,
``channels/out_channels``) and the YAML's / upstream's (``ddconfig/lossconfig/embed_dim``, ``ch/out_ch``)
- repair B12.  Parameter names follow the reference (``up.N.up_sample.conv``; upstream's ``upsample`` is
accepted on load).  ``decode`` is restated (absent from the reference): post_quant_conv -> Decoder.

``vae_attn`` selects what the single-head mid attention computes:
  "reference": literally model.py:519-523 - W = softmax_j(q_i.k_j/sqrt(C)) but V contracted over the
               QUERY index: out[:, j] = sum_i v[:, i] W[i, j]   (the parity default)
  "standard":  upstream's softmax(QK^T) V   (what real SD weights need)
"""
import torch
import torch.nn as nn

from . import ops
from .util import instantiate_from_config

BF16 = torch.bfloat16


def _gn(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = _gn(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = _gn(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = _gn(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)


class UpSample(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)


class DownSample(nn.Module):
    """model.py:76-106 with with_conv=True: F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 without padding (the HIP conv
    folds the one-sided padding into its bounds check: ops.conv3x3(..., stride=2, asym_pad=True))."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, stride=2, padding=0)


class _VaeNet(nn.Module):
    """What Encoder and Decoder share on the HIP path: weight packing, ResnetBlock, AttnBlock, dtype policy."""

    vae_attn = "reference"
    compute_dtype = torch.bfloat16

    def _apply(self, fn, *a, **kw):
        self._pack = None
        return super()._apply(fn, *a, **kw)

    def invalidate_packed(self):
        self._pack = None

    def set_compute_dtype(self, dtype):
        """bf16 or fp16 storage / MFMA operand type, or torch.float32 = the fp32 mode of the path (fp32 activations,
        contractions on exact three-way bf16 splits; LatentDiffusion aligns it with the UNet's policy)."""
        if dtype not in ops.HALF_TYPES and dtype != torch.float32:
            raise NotImplementedError("HIP VAE: compute dtype must be torch.bfloat16, torch.float16 or torch.float32")
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._pack = None

    @torch.no_grad()
    def _build_pack(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise ops.hip.SdmiError("HIP VAE forward needs the model on a GPU; there is no CPU fallback")
        f32 = lambda t: t.detach().float().contiguous()
        pk = {}
        cd = self.compute_dtype
        # residual-stream storage (ops.stream_dtype: f16 under bf16 operands): layers whose output is the stream get
        # out_dtype=sd, layers whose INPUT is the raw stream (nin_shortcut, the Up / DownSample convs) run in the stream type
        sd = ops.stream_dtype(cd) if cd in ops.HALF_TYPES else None
        for m in self.modules():
            if isinstance(m, ResnetBlock):
                e = {"gn1": (f32(m.norm1.weight), f32(m.norm1.bias)),
                     "conv1": ops.pack_conv3x3(m.conv1.weight, m.conv1.bias, dtype=cd, out_dtype=sd),
                     "gn2": (f32(m.norm2.weight), f32(m.norm2.bias)),
                     "conv2": ops.pack_conv3x3(m.conv2.weight, m.conv2.bias, dtype=cd, out_dtype=sd)}
                if hasattr(m, "nin_shortcut"):
                    e["skip"] = ops.pack_linear(m.nin_shortcut.weight, m.nin_shortcut.bias, dtype=sd or cd)
                pk[id(m)] = e
            elif isinstance(m, AttnBlock):
                c = m.in_channels
                pk[id(m)] = {
                    "gn": (f32(m.norm.weight), f32(m.norm.bias)),
                    "qkv": ops.pack_linear_cat([m.q.weight, m.k.weight, m.v.weight], [m.q.bias, m.k.bias, m.v.bias], dtype=cd),
                    "proj": ops.pack_linear(m.proj_out.weight, m.proj_out.bias, dtype=cd, out_dtype=sd),
                }
            elif isinstance(m, (UpSample, DownSample)):
                pk[id(m)] = ops.pack_conv3x3(m.conv.weight, m.conv.bias, dtype=sd or cd)
        pk["conv_in"] = ops.pack_conv3x3(self.conv_in.weight, self.conv_in.bias, dtype=cd, out_dtype=sd)
        pk["norm_out"] = (f32(self.norm_out.weight), f32(self.norm_out.bias))
        pk["conv_out"] = ops.pack_conv3x3(self.conv_out.weight, self.conv_out.bias, dtype=cd)
        self._pack = pk
        return pk

    def _resnet(self, pk, m, h):
        e = pk[id(m)]
        n1 = ops.groupnorm(h, e["gn1"][0], e["gn1"][1], 1e-6, True, out_dtype=e["conv1"].dtype)
        h1 = ops.conv3x3(n1, e["conv1"])
        n2 = ops.groupnorm(h1, e["gn2"][0], e["gn2"][1], 1e-6, True, out_dtype=e["conv2"].dtype)
        sk = ops.linear(h, e["skip"]) if "skip" in e else h
        return ops.conv3x3(n2, e["conv2"], res=sk)

    def _attn(self, pk, m, h):
        """AttnBlock.forward (model.py:496-528): GroupNorm -> q | k | v (one GEMM) -> single-head attention (two
        launches, no [B, N, N] tensor) -> proj_out + residual."""
        e = pk[id(m)]
        b, hh, ww, c = h.shape
        n = hh * ww
        if self.vae_attn not in ("reference", "standard"):
            raise ValueError(f"vae_attn must be 'reference' or 'standard', got {self.vae_attn!r}")
        if c not in (64, 128, 256, 512) or (n % 2 and self.compute_dtype != torch.float32):
            raise NotImplementedError("VAE attention is built for 64 / 128 / 256 / 512 channels and an even token count")
        hn = ops.groupnorm(h, e["gn"][0], e["gn"][1], 1e-6, False, out_dtype=e["qkv"].dtype).reshape(b * n, c)
        qkv = ops.linear(hn, e["qkv"])
        o = ops.attention_vae(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, n, float(int(c) ** (-0.5)), self.vae_attn)
        out = ops.linear(o, e["proj"], res=h.reshape(b * n, c))
        return out.reshape(b, hh, ww, c)


class Encoder(_VaeNet):
    """VAE Encoder on the HIP kernels (model.py:145-279): image -> 2*z_channels moments."""

    def __init__(self, *, channels=None, out_channels=None, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True,
                 use_linear_attn=False, attn_type="vanilla", ch=None, out_ch=None, vae_attn="reference",
                 compute_dtype=torch.bfloat16, **ignore_kwargs):
        super().__init__()
        self.compute_dtype = compute_dtype
        channels = channels if channels is not None else ch
        if channels is None:
            raise TypeError("Encoder needs channels/ch")
        if use_linear_attn or attn_type not in ("vanilla",) or not resamp_with_conv:
            raise NotImplementedError("HIP Encoder: only the SD2 configuration (vanilla attention, conv resample) is built")
        self.channels = channels
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.vae_attn = vae_attn
        self.conv_in = nn.Conv2d(in_channels, channels, 3, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = channels
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = channels * in_ch_mult[i_level]
            block_out = channels * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = DownSample(block_in)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)
        self._pack = None

    @torch.no_grad()
    def forward(self, x):
        """x: fp32 NCHW image [B, in_channels, H, W] -> 16-bit NHWC moments [B, H/f, W/f, 2*z_channels] (model.py:240-279)."""
        pk = self._pack or self._build_pack()
        h = ops.nchw_to_nhwc(x.float().contiguous(), dtype=self.compute_dtype)       # channels padded 3 -> 8
        h = ops.conv3x3(h, pk["conv_in"])
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = self._resnet(pk, lvl.block[i_block], h)
                if len(lvl.attn) > 0:
                    h = self._attn(pk, lvl.attn[i_block], h)
            if i_level != self.num_resolutions - 1:
                h = ops.conv3x3(h, pk[id(lvl.downsample)], stride=2, asym_pad=True)
        h = self._resnet(pk, self.mid.block_1, h)
        h = self._attn(pk, self.mid.attn_1, h)
        h = self._resnet(pk, self.mid.block_2, h)
        n = ops.groupnorm(h, pk["norm_out"][0], pk["norm_out"][1], 1e-6, True, out_dtype=pk["conv_out"].dtype)
        return ops.conv3x3(n, pk["conv_out"])


class DiagonalGaussianDistribution(object):
    """distribution.py:10-43 on device tensors: parameters fp32 NCHW [B, 2C, h, w] -> mean / log_var (clamped to
    [-30, 20]) / std / var views computed by sdmi_diag_gaussian; sample() = mean + std * N(0, 1); mode() = mean."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic
        self.mean, self.log_var, self.std, self.var = ops.diag_gaussian(parameters, deterministic)

    def sample(self, noise=None):
        noise = torch.randn_like(self.mean) if noise is None else noise
        return ops.lincomb_mul_add(self.mean, self.std, noise)

    def mode(self):
        return self.mean


class Decoder(_VaeNet):
    def __init__(self, *, channels=None, out_channels=None, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False,
                 tanh_out=False, use_linear_attn=False, attn_type="vanilla", ch=None, out_ch=None,
                 vae_attn="reference", compute_dtype=torch.bfloat16, **ignorekwargs):
        super().__init__()
        self.compute_dtype = compute_dtype
        channels = channels if channels is not None else ch
        out_channels = out_channels if out_channels is not None else out_ch
        if channels is None or out_channels is None:
            raise TypeError("Decoder needs channels/ch and out_channels/out_ch")
        if use_linear_attn or attn_type not in ("vanilla",) or not resamp_with_conv or give_pre_end or tanh_out:
            raise NotImplementedError("HIP Decoder: only the SD2 configuration (vanilla attention, conv resample) is built")
        self.channels = channels
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.vae_attn = vae_attn
        block_in = channels * ch_mult[-1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = channels * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.up_sample = UpSample(block_in)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = _gn(block_in)              # repair B5
        self.conv_out = nn.Conv2d(block_in, out_channels, 3, padding=1)
        self._pack = None

    @torch.no_grad()
    def forward(self, z, scale=1.0, nhwc=None):
        """z: fp32 NCHW latent [B, z_channels, h, w] (or, nhwc=True, the channels-last activation decode() hands over:
        16-bit, or fp32 in fp32 mode) -> fp32 NCHW image [B, out_channels, 8h, 8w] (model.py:371-411)."""
        pk = self._pack or self._build_pack()
        if nhwc is None:
            nhwc = z.dtype in ops.HALF_TYPES
        if nhwc:
            h = z
        else:
            h = ops.nchw_to_nhwc(z.float().contiguous(), scale=scale, dtype=self.compute_dtype)
        h = ops.conv3x3(h, pk["conv_in"])
        h = self._resnet(pk, self.mid.block_1, h)
        h = self._attn(pk, self.mid.attn_1, h)
        h = self._resnet(pk, self.mid.block_2, h)
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = self._resnet(pk, lvl.block[i_block], h)
                if len(lvl.attn) > 0:
                    h = self._attn(pk, lvl.attn[i_block], h)
            if i_level != 0:
                h = ops.conv3x3(h, pk[id(lvl.up_sample)], upsample=True)
        n = ops.groupnorm(h, pk["norm_out"][0], pk["norm_out"][1], 1e-6, True, out_dtype=pk["conv_out"].dtype)
        return ops.conv3x3(n, pk["conv_out"], out_nchw_f32=True)


class AutoencoderKL(nn.Module):
    def __init__(self, dd_config=None, loss_config=None, embedding_dim=None, ckpt_path=None, ignore_keys=(),
                 image_key="image", colorize_nlabels=None, monitor=None, ema_decay=None, learn_log_var=False,
                 ddconfig=None, lossconfig=None, embed_dim=None, vae_attn="reference", compute_dtype=torch.bfloat16):
        super().__init__()
        dd_config = dict(dd_config if dd_config is not None else ddconfig)
        loss_config = loss_config if loss_config is not None else lossconfig
        embedding_dim = embedding_dim if embedding_dim is not None else embed_dim
        assert dd_config["double_z"], "double_z must be set"
        self.image_key = image_key
        self.learn_log_var = learn_log_var
        self.embedding_dim = embedding_dim
        # encoder half: img2img / inpainting; autoencoder.py:44-54
        self.encoder = Encoder(vae_attn=vae_attn, compute_dtype=compute_dtype, **dd_config)
        self.decoder = Decoder(vae_attn=vae_attn, compute_dtype=compute_dtype, **dd_config)
        self.loss = instantiate_from_config(loss_config) if loss_config is not None else None
        self.quant_conv = nn.Conv2d(2 * dd_config["z_channels"], 2 * embedding_dim, 1)
        self.post_quant_conv = nn.Conv2d(embedding_dim, dd_config["z_channels"], 1)
        if monitor is not None:
            self.monitor = monitor
        self._pq = self._q = None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=list(ignore_keys))

    def init_from_ckpt(self, path, ignore_keys=()):
        sd = torch.load(path, map_location="cpu", weights_only=True)["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print(f"Deleting key {k} from state_dict.")
                del sd[k]
        self.load_state_dict(sd, strict=False)

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k.replace(".upsample.", ".up_sample."): v for k, v in state_dict.items()}
        self._pq = self._q = None
        self.decoder._pack = None
        self.encoder._pack = None
        return super().load_state_dict(sd, strict=strict, **kw)

    def _apply(self, fn, *a, **kw):
        self._pq = self._q = None
        return super()._apply(fn, *a, **kw)

    def set_compute_dtype(self, dtype):
        self.decoder.set_compute_dtype(dtype)
        self.encoder.set_compute_dtype(dtype)
        self._pq = self._q = None

    @torch.no_grad()
    def encode(self, x):
        """Encoder then quant_conv (1x1) -> DiagonalGaussianDistribution over fp32 NCHW moments (restated: the
        reference builds encoder + quant_conv, autoencoder.py:44-53, but has no encode method)."""
        if self._q is None:
            self._q = ops.pack_linear(self.quant_conv.weight, self.quant_conv.bias, dtype=self.encoder.compute_dtype)
        h = self.encoder(x)                                            # [B, h, w, 2*z] 16-bit NHWC
        b, hh, ww, c = h.shape
        moments = ops.linear(h.reshape(b * hh * ww, c), self._q, out_nchw_f32=(b, hh, ww))
        return DiagonalGaussianDistribution(moments)

    @torch.no_grad()
    def decode(self, z, scale=1.0):
        """post_quant_conv (1x1, autoencoder.py:54) then Decoder.  `scale` multiplies z first
        (decode_first_stage passes 1/scale_factor so the division is fused into the layout kernel)."""
        if self.decoder.compute_dtype == torch.float32:      # fp32 mode: no channel padding (split3 pads per plane)
            if self._pq is None:
                w = self.post_quant_conv.weight
                self._pq = ops.pack_linear(w.reshape(w.shape[0], w.shape[1]), self.post_quant_conv.bias, dtype=torch.float32)
            h = ops.nchw_to_nhwc(z.float().contiguous(), scale=scale, dtype=torch.float32)        # [B, h, w, embed_dim]
            return self.decoder(ops.linear(h, self._pq), nhwc=True)
        if self._pq is None:
            w = self.post_quant_conv.weight.detach().float()
            zc, ed = w.shape[0], w.shape[1]
            zp = (zc + 7) // 8 * 8
            wp = torch.zeros(zp, (ed + 7) // 8 * 8, device=w.device)
            wp[:zc, :ed] = w.reshape(zc, ed)
            bp = torch.zeros(zp, device=w.device)
            bp[:zc] = self.post_quant_conv.bias.detach().float()
            self._pq = ops.pack_linear(wp, bp, dtype=self.decoder.compute_dtype)     # output channels padded to 8 (zeros) for conv_in's 16-byte loads
        h = ops.nchw_to_nhwc(z.float().contiguous(), cpad=self._pq.k, scale=scale,
                             dtype=self.decoder.compute_dtype)                        # [B, h, w, 64], zero padded
        h = ops.linear(h, self._pq)                                                    # [B, h, w, 8]
        return self.decoder(h, nhwc=True)

    def forward(self, z):
        return self.decode(z)
