"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

Every tensor is a pure function of (its state-dict key, its shape): the RNG seed is crc32(key), so the
same weights can be rebuilt anywhere without shipping them (the full SD2 UNet is 3.5 GB in fp32).
Zero-initialised tensors of the reference (zero_module: second ResBlock conv, proj_out, final conv -
SURVEY.md §0.7) get random values like every other tensor, so no parity check is vacuous.
"""
import zlib

import torch


def synth_tensor(key, shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) == 0:
        return torch.zeros(shape, dtype=dtype)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "bias":
        return (0.05 * torch.randn(shape, generator=g)).to(dtype)
    if len(shape) == 1:                      # norm scales: around 1
        return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return (torch.randn(shape, generator=g) * fan_in ** -0.5).to(dtype)


def synth_state_dict(shapes):
    """shapes: {key: shape} (e.g. from a meta-device module's state_dict())."""
    return {k: synth_tensor(k, tuple(s)) for k, s in shapes.items()}


def synth_input(name, shape, scale=1.0):
    g = torch.Generator().manual_seed(zlib.crc32(("input:" + name).encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g) * scale


def init8d_state_dict(cls, cfg, seed=0):
    """SURVEY.md §8(d) weight recipe: ``torch.manual_seed(seed)`` + the module's default ``nn`` initialisation, then
    every tensor the reference zero-initialises (``zero_module``: second ResBlock conv, SpatialTransformer.proj_out,
    the final conv - SURVEY §0.7), weight and bias, re-randomised from ONE ``Generator(1)`` stream in state-dict
    order as ``randn * 0.02``.  ``cls(**cfg)`` must build the module tree with default init on the CPU."""
    prev = torch.get_rng_state()
    torch.manual_seed(seed)
    m = cls(**cfg)
    torch.set_rng_state(prev)
    sd = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    zeroed = set()
    for name, mod in m.named_modules():
        w = getattr(mod, "weight", None)
        if w is not None and w.dim() >= 2 and float(w.detach().abs().max()) == 0.0:
            zeroed.add(name)
    for k in sd:
        if k.rsplit(".", 1)[0] in zeroed:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    return sd


SD2_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, legacy=False,
                use_checkpoint=True, use_fp16=True)
SD2_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
# small configurations with the same structure (head dim 64 so the HIP attention kernel applies)
TINY_UNET = dict(image_size=16, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2, 1],
                 num_res_blocks=1, channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True,
                 use_linear_in_transformer=True, transformer_depth=1, context_dim=96, legacy=False,
                 use_checkpoint=True, use_fp16=True)
TINY_DDCONFIG = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=64,
                     ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
