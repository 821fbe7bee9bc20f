#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own layer code on seeded inputs.

Runs only in the authoring container (needs /root/reference, read-only; nothing from it is copied:
this file is loader code + a list of which reference classes to call).  The reference package cannot
be imported as shipped, so its files are loaded one by one with
importlib.util.spec_from_file_location under a synthetic parent package whose attributes satisfy the
`from ...ldm import ...` lines; `pytorch_lightning` and `omegaconf` (not installed) are stubbed with
empty stand-ins that execute nothing.

Leaf modules run UNMODIFIED.  Composites that crash as shipped are run with the in-memory repairs of
SURVEY.md §8(a-bugs) (B1/B2 time_step_embedding, B3 proj_in alias, B4 final conv, B5/B6 decoder
norm_out/upsample) so that the goldens still come from the reference's own layer code.

Weights: tests/synth.py - every tensor a pure function of its state-dict key.

usage: python tools/gen_golden.py [--full]      (--full also writes the SD2-size h=32 checksums)
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from tests import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m
        return m

    R = pkg("R", REF)
    ldm = pkg("R.ldm", f"{REF}/ldm")
    modules = pkg("R.ldm.modules", f"{REF}/ldm/modules")
    pkg("R.ldm.modules.diffusion", f"{REF}/ldm/modules/diffusion")
    pkg("R.ldm.models", f"{REF}/ldm/models")
    pkg("R.ldm.models.diffusion", f"{REF}/ldm/models/diffusion")
    R.ldm = ldm
    # absent third-party packages: inert stand-ins
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    sys.modules["pytorch_lightning"] = pl
    oc = types.ModuleType("omegaconf")
    ocl = types.ModuleType("omegaconf.listconfig")
    ocl.ListConfig = type("ListConfig", (list,), {})
    oc.listconfig = ocl
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = ocl

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, f"{REF}/{rel}")
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    utils = load("R.ldm.utils", "ldm/utils.py")
    for n in ("instantiate_from_config", "checkpoint", "conv_nd", "group_norm", "avg_pool_nd", "zero_module",
              "time_step_embedding", "count_params", "default"):
        setattr(ldm, n, getattr(utils, n))
    attention = load("R.ldm.modules.attention", "ldm/modules/attention.py")
    modules.SpatialTransformer = attention.SpatialTransformer
    modules.MemoryEfficientCrossAttention = attention.MemoryEfficientCrossAttention
    ema = load("R.ldm.modules.ema", "ldm/modules/ema.py")
    modules.LitEma = ema.LitEma
    ldm.LitEma = ema.LitEma
    util = load("R.ldm.modules.diffusion.util", "ldm/modules/diffusion/util.py")
    for n in ("make_ddim_timesteps", "make_ddim_sampling_parameters", "make_beta_schedule"):
        setattr(modules, n, getattr(util, n))
        setattr(ldm, n, getattr(util, n))
    model = load("R.ldm.modules.diffusion.model", "ldm/modules/diffusion/model.py")
    modules.Encoder, modules.Decoder = model.Encoder, model.Decoder
    openai = load("R.ldm.modules.diffusion.openai_model", "ldm/modules/diffusion/openai_model.py")
    ddim = load("R.ldm.models.diffusion.ddim", "ldm/models/diffusion/ddim.py")
    distribution = load("R.ldm.modules.distribution.distribution", "ldm/modules/distribution/distribution.py")
    return types.SimpleNamespace(utils=utils, attention=attention, util=util, model=model, openai=openai, ddim=ddim,
                                 distribution=distribution)


def seed_module(m, prefix):
    """Overwrite every parameter/buffer with synth_tensor(prefix + key)."""
    sd = m.state_dict()
    new = {k: synth.synth_tensor(prefix + k, v.shape) for k, v in sd.items()}
    m.load_state_dict(new)
    return new


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
        else:
            flat[k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def repaired_time_step_embedding(time_steps, dim, max_period=10000, repeat_only=False):
    """B1/B2 repair applied to the reference's function body (utils.py:292-328)."""
    import math
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(time_steps.device)
    args = time_steps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def build_ref_unet(ref, cfg):
    ref.openai.time_step_embedding = repaired_time_step_embedding                      # B1/B2
    ref.attention.SpatialTransformer.proj_in = property(lambda s: s.project_in)       # B3
    u = ref.openai.UNetModel(**cfg)
    u.out[2] = nn.Conv2d(cfg["model_channels"], cfg["out_channels"], 3, padding=1)    # B4
    return u.eval()


def build_ref_decoder(ref, dd):
    kw = dict(dd)
    kw["channels"] = kw.pop("ch")
    kw["out_channels"] = kw.pop("out_ch")
    orig_swish = ref.model.swish
    ref.model.swish = lambda x: ref.model.group_norm(x) if isinstance(x, int) else orig_swish(x)   # B5
    try:
        d = ref.model.Decoder(**kw)
    finally:
        ref.model.swish = orig_swish
    for lvl in d.up:                                                                   # B6
        if hasattr(lvl, "up_sample"):
            lvl.__dict__["upsample"] = lvl.up_sample     # plain attribute: no second state-dict entry
    return d.eval()


@torch.no_grad()
def gen_leaves(ref):
    A, O, Mo = ref.attention, ref.openai, ref.model
    x_tok = synth.synth_input("tok", (2, 10, 64))
    ctx = synth.synth_input("ctx", (2, 7, 24))
    cases = {}

    def run(name, module, *inputs, call=None):
        sd = seed_module(module, name + ".")
        out = (call or module)(*inputs)
        cases[name] = dict(sd={k: v.numpy() for k, v in sd.items()}, out=out.numpy())

    run("cross_attention", A.CrossAttention(query_dim=64, context_dim=24, heads=2, dim_head=32).eval(), x_tok, ctx)
    run("self_attention", A.CrossAttention(query_dim=64, heads=2, dim_head=32).eval(), x_tok)
    run("feed_forward", A.FeedForward(64, glu=True).eval(), x_tok)
    run("basic_transformer_block", A.BasicTransformerBlock(64, 2, 32, context_dim=24, checkpoint=True).eval(), x_tok, ctx)
    x_img = synth.synth_input("img", (2, 64, 8, 8))
    emb = synth.synth_input("emb", (2, 128))
    run("res_block_same", O.ResBlock(64, 128, 0.0).eval(), x_img, emb)
    run("res_block_proj", O.ResBlock(64, 128, 0.0, out_channels=96).eval(), x_img, emb)
    run("down_sample", O.DownSample(64, use_conv=True).eval(), x_img)
    run("up_sample", O.UpSample(64, use_conv=True).eval(), x_img)
    run("group_norm32", ref.utils.group_norm(64).eval(), x_img)
    run("vae_resnet_same", Mo.ResnetBlock(in_channels=64, out_channels=64, dropout=0.0, time_embedding_channels=0).eval(),
        x_img, None)
    run("vae_resnet_proj", Mo.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, time_embedding_channels=0).eval(),
        x_img, None)
    run("vae_attn_block", Mo.AttnBlock(64).eval(), x_img)
    run("vae_up_sample", Mo.UpSample(64, with_conv=True).eval(), x_img)
    # SpatialTransformer needs only the B3 alias
    A.SpatialTransformer.proj_in = property(lambda s: s.project_in)
    run("spatial_transformer_linear",
        A.SpatialTransformer(64, 2, 32, depth=1, context_dim=24, use_linear=True, use_checkpoint=True).eval(), x_img, ctx)
    run("spatial_transformer_conv",
        A.SpatialTransformer(64, 2, 32, depth=1, context_dim=24, use_linear=False, use_checkpoint=True).eval(), x_img, ctx)
    flat = {"in/tok": x_tok.numpy(), "in/ctx": ctx.numpy(), "in/img": x_img.numpy(), "in/emb": emb.numpy()}
    for name, c in cases.items():
        flat[f"{name}/out"] = c["out"]
        for k, v in c["sd"].items():
            flat[f"{name}/sd/{k}"] = v
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "leaf_modules.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), cases: {sorted(cases)}")


@torch.no_grad()
def gen_tables(ref):
    U = ref.util
    betas = U.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120)
    out = {"betas_linear": betas, "betas_cosine": np.asarray(U.make_beta_schedule("cosine", 1000)),
           "betas_sqrt_linear": U.make_beta_schedule("sqrt_linear", 1000, linear_start=0.00085, linear_end=0.0120)}
    acp = np.cumprod(1.0 - betas, axis=0)
    acp32 = torch.tensor(acp, dtype=torch.float32)
    for s in (10, 30, 50):
        ts = U.make_ddim_timesteps("uniform", s, 1000, verbose=False)
        out[f"ddim_timesteps_{s}"] = ts
        for eta in (0.0, 0.5):
            sig, a, ap = U.make_ddim_sampling_parameters(acp32, ts, eta, verbose=False)
            tag = f"{s}_eta{eta}"
            out[f"sigmas_{tag}"] = np.asarray(sig, dtype=np.float64)
            out[f"alphas_{tag}"] = np.asarray(a, dtype=np.float64)
            out[f"alphas_prev_{tag}"] = np.asarray(ap, dtype=np.float64)
    out["ddim_timesteps_quad_20"] = U.make_ddim_timesteps("quad", 20, 1000, verbose=False)

    # DDIMSampler.make_schedule on a mock model exposing the attribute names the sampler reads (B8/B9)
    mock = types.SimpleNamespace(
        num_timesteps=1000, betas=torch.tensor(betas, dtype=torch.float32), alphas_cumprod=acp32,
        alphas_cumprod_prev=torch.tensor(np.append(1.0, acp[:-1]), dtype=torch.float32), device=torch.device("cpu"))
    smp = ref.ddim.DDIMSampler(mock, device=torch.device("cpu"))
    smp.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
    out["sampler50_ddim_alphas"] = np.asarray(smp.ddim_alphas, dtype=np.float64)
    out["sampler50_ddim_alphas_prev"] = np.asarray(smp.ddim_alphas_prev, dtype=np.float64)
    out["sampler50_ddim_sigmas"] = np.asarray(smp.ddim_sigmas, dtype=np.float64)
    out["sampler50_ddim_sqrt_one_minus_alphas"] = np.asarray(smp.ddim_sqrt_one_minus_alphas, dtype=np.float64)
    out["sampler50_sqrt_recipm1_alphas_cumprod"] = smp.sqrt_recipm1_alphas_cumprod.numpy()
    save("schedule_tables", **out)


def build_ref_encoder(ref, dd):
    kw = dict(dd)
    kw["channels"] = kw.pop("ch")
    kw["out_channels"] = kw.pop("out_ch")
    orig_swish = ref.model.swish
    ref.model.swish = lambda x: ref.model.group_norm(x) if isinstance(x, int) else orig_swish(x)   # B5, should it recur
    try:
        e = ref.model.Encoder(**kw)
    finally:
        ref.model.swish = orig_swish
    return e.eval()


@torch.no_grad()
def gen_tiny_encoder(ref):
    """the VAE Encoder (model.py:145-279) and DiagonalGaussianDistribution
    (distribution.py:10-53) of the reference on the tiny config."""
    e = build_ref_encoder(ref, synth.TINY_DDCONFIG)
    seed_module(e, "encoder.")
    x = synth.synth_input("tiny_img", (2, 3, 32, 32))
    moments = e(x)
    dist_mod = ref.distribution
    params = synth.synth_input("tiny_moments", (2, 8, 4, 4), 12.0)      # wide range: exercises the log-var clamp
    out = dict(x=x, moments=moments, keys=np.asarray(sorted(e.state_dict().keys())), params=params)
    if dist_mod is not None:
        g = dist_mod.DiagonalGaussianDistribution(params)
        torch.manual_seed(123)
        smp = g.sample()
        torch.manual_seed(123)
        noise = torch.randn_like(g.mean)
        out.update(mean=g.mean, log_var=g.log_var, std=g.std, var=g.var, sample=smp, noise=noise)   # kl() is a stub (returns None)
    save("tiny_encoder", **out)


@torch.no_grad()
def gen_tiny_composites(ref):
    cfg = dict(synth.TINY_UNET, use_fp16=False)     # fp32 reference path
    u = build_ref_unet(ref, cfg)
    seed_module(u, "unet.")
    x = synth.synth_input("tiny_x", (2, 4, 16, 16))
    t = torch.tensor([981, 21], dtype=torch.long)
    ctx = synth.synth_input("tiny_ctx", (2, 7, 96))
    y = u(x, t, ctx)
    temb = repaired_time_step_embedding(t, cfg["model_channels"])
    save("tiny_unet", x=x, t=t, ctx=ctx, out=y, time_embedding=temb,
         n_params=np.asarray(sum(p.numel() for p in u.parameters())),
         keys=np.asarray(sorted(u.state_dict().keys())))

    d = build_ref_decoder(ref, synth.TINY_DDCONFIG)
    seed_module(d, "decoder.")
    z = synth.synth_input("tiny_z", (2, 4, 16, 16))
    img = d(z)
    save("tiny_decoder", z=z, out=img, keys=np.asarray(sorted(d.state_dict().keys())))


@torch.no_grad()
def gen_full(ref):
    """SD2-size (v2-inference.yaml) UNet at h=32 and Decoder at 256^2: checksums / first values only."""
    u = build_ref_unet(ref, dict(synth.SD2_UNET, use_fp16=False))
    seed_module(u, "unet.")
    x = synth.synth_input("sd2_x", (1, 4, 32, 32))
    t = torch.tensor([501], dtype=torch.long)
    ctx = synth.synth_input("sd2_ctx", (1, 77, 1024))
    y = u(x, t, ctx)
    n_params = sum(p.numel() for p in u.parameters())
    n_keys = len(u.state_dict())
    del u
    d = build_ref_decoder(ref, synth.SD2_DDCONFIG)
    seed_module(d, "decoder.")
    z = synth.synth_input("sd2_z", (1, 4, 32, 32))
    img = d(z)
    save("sd2_full_h32",
         unet_first64=y.flatten()[:64], unet_mean=y.mean(), unet_std=y.std(), unet_abs_sum=y.abs().sum(),
         unet_sub=y[0, :, ::4, ::4], unet_params=np.asarray(n_params), unet_keys=np.asarray(n_keys),
         dec_first64=img.flatten()[:64], dec_mean=img.mean(), dec_std=img.std(), dec_sub=img[0, :, ::16, ::16],
         dec_params=np.asarray(sum(p.numel() for p in d.parameters())))


@torch.no_grad()
def gen_full_init8d(ref):
    """SD2-size UNet at h=32 on the SURVEY.md §8(d) weight recipe (default nn init under manual_seed(0); zero_module'd
    tensors Generator(1) * 0.02 - tests/synth.py:init8d_state_dict), run through the reference's own layer code."""
    from stablediffusion_amd.unet import UNetModel
    cfg = dict(synth.SD2_UNET, use_fp16=False)
    sd = synth.init8d_state_dict(UNetModel, cfg)
    u = build_ref_unet(ref, cfg)
    missing = u.load_state_dict(sd, strict=True)
    x = synth.synth_input("sd2_x", (1, 4, 32, 32))
    t = torch.tensor([501], dtype=torch.long)
    ctx = synth.synth_input("sd2_ctx", (1, 77, 1024))
    y = u(x, t, ctx)
    # fingerprints of the weights, so that a test can tell "the recipe produced other weights here" from a parity failure
    fp = {k: sd[k].flatten()[:4].clone() for k in ("input_blocks.0.0.weight", "out.2.weight", "middle_block.1.proj_out.weight",
                                                   "output_blocks.11.0.out_layers.3.weight", "time_embedding.2.bias")}
    save("sd2_init8d_h32", unet_sub=y[0, :, ::4, ::4], unet_first64=y.flatten()[:64], unet_std=y.std(), unet_absmax=y.abs().max(),
         **{"fp/" + k: v for k, v in fp.items()})


@torch.no_grad()
def gen_st_d64(ref):
    """SpatialTransformer leaves with head dim 64 (the width the HIP attention kernel is built for; the leaf goldens of
    gen_leaves use head dim 32): the conv-1x1 project_in / proj_out variant (use_linear=False, attention.py:213-226) and the
    linear one, the reference's own module code (B3 alias only)."""
    A = ref.attention
    A.SpatialTransformer.proj_in = property(lambda s: s.project_in)
    x_img = synth.synth_input("img128", (2, 128, 8, 8))
    ctx = synth.synth_input("ctx", (2, 7, 24))
    flat = {"in/img": x_img.numpy(), "in/ctx": ctx.numpy()}
    for name, lin in (("spatial_transformer_conv_d64", False), ("spatial_transformer_linear_d64", True)):
        m = A.SpatialTransformer(128, 2, 64, depth=1, context_dim=24, use_linear=lin, use_checkpoint=True).eval()
        sd = seed_module(m, name + ".")
        flat[f"{name}/out"] = m(x_img, ctx).numpy()
        for k, v in sd.items():
            flat[f"{name}/sd/{k}"] = v.numpy()
    path = os.path.join(OUT, "st_d64.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only-init8d", action="store_true", help="write tests/golden/sd2_init8d_h32.npz only")
    ap.add_argument("--only-st-d64", action="store_true", help="write tests/golden/st_d64.npz only")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    reference = load_reference()
    if args.only_init8d:
        gen_full_init8d(reference)
        raise SystemExit(0)
    if args.only_st_d64:
        gen_st_d64(reference)
        raise SystemExit(0)
    gen_leaves(reference)
    gen_tables(reference)
    gen_tiny_composites(reference)
    gen_tiny_encoder(reference)
    gen_st_d64(reference)
    if args.full:
        gen_full(reference)
        gen_full_init8d(reference)
