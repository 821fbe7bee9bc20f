"""End-to-end GPU parity of the HIP path (through the C ABI) against the CPU oracle and the committed
reference goldens, on identical seeded weights / latents / text embeddings.

Gates (BASELINE.json north_star):  relative error = max|a-b| / max|b| per tensor
  * per-step eps/v prediction  <= 1e-2  against the reference's fp32 path  (NORTH_STAR)
  * decoded image PSNR          >= 35 dB

The benchmarked (headline) dtype is **fp16** - the operand type the reference's own v2-inference.yaml:24 selects
(use_fp16: True) - and every fp16 case below is asserted at NORTH_STAR = 1e-2 (measured 1.2-2.4e-3).
bf16 operands (use_bf16) run at the same MFMA rate and are kept as the secondary policy.  Since round 3 the bf16 policy
keeps its RESIDUAL STREAM in f16 (ops.stream_dtype: same bytes, 11 significant bits; only the MFMA operands - weights,
norm outputs, q/k/v, the GEGLU product - carry bf16 rounding, and the few layers that read the raw stream run as f16
layers): every per-step bf16 case of this file now measures <= 1e-2 against the fp32 oracle / the reference's goldens
(4.8e-3 ... 9.8e-3; rounds 1-2 with a bf16 stream: 0.9-1.8e-2), i.e. bf16 meets the north-star figure as well.  Its gate
is BF16_DOCUMENTED = 1.2e-2 = 1.2 x the worst measured case (teacher-forced v-prediction DDIM step, 9.83e-3) rather than
1e-2 itself, because that worst case sits 2 % under the figure and 8-bit operand rounding is input dependent: rounding
ONLY the two operands of every contraction to bf16 (oracle.model.operand_precision - an ideal bf16 MFMA with fp32
accumulators and an fp32 stream) costs 0.8-1.4e-2 on these networks (tools/precision_study.py).
Every measured number is printed as a [parity] line.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NORTH_STAR = 1e-2
BF16_DOCUMENTED = 1.2e-2      # secondary dtype only - see the module docstring; never applied to the headline dtype.
                              # = 1.2 x the worst bf16 per-step figure measured over all cases of this file (9.83e-3)
HEADLINE = "fp16"             # the dtype bench.py reports
BF16 = torch.bfloat16


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def psnr(a, b, peak=2.0):
    """images in [-1, 1] -> peak-to-peak 2."""
    mse = ((a.detach().float().cpu() - b.detach().float().cpu()) ** 2).mean().item()
    return 10 * math.log10(peak * peak / max(mse, 1e-20))


def build(cls, cfg, prefix, dev, **extra):
    """Instantiate on the meta device, then materialise the synthetic weights (pure function of the key)."""
    with torch.device("meta"):
        m = cls(**cfg, **extra)
    sd = {k: synth.synth_tensor(prefix + k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)
    return m.to(dev).eval(), sd


def policy(cfg, dt):
    """The reference's dtype flags (openai_model.py:602-603) select the MFMA operand type."""
    return dict(cfg, use_fp16=(dt == "fp16"), use_bf16=(dt == "bf16"))


TOL = {"bf16": BF16_DOCUMENTED, "fp16": NORTH_STAR}
assert TOL[HEADLINE] == NORTH_STAR
TORCH_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_tiny_unet_vs_reference_golden(dev, dt):
    from stablediffusion_amd.unet import UNetModel
    z = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    m, _ = build(UNetModel, policy(synth.TINY_UNET, dt), "unet.", dev)
    x, t, ctx = (torch.from_numpy(z[k]).to(dev) for k in ("x", "t", "ctx"))
    y = m(x, t, ctx)
    assert y.shape == x.shape and y.dtype == x.dtype
    e = rel_err(y, torch.from_numpy(z["out"]))
    print(f"[parity] tiny unet ({dt}) vs reference golden: rel_err {e:.3e}")
    assert e < TOL[dt]
    # upstream keyword spellings (timesteps=) and a second call (cached context K/V) give the same answer
    y2 = m(x, timesteps=t, context=ctx)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("mode,dt", [("reference", "bf16"), ("standard", "bf16"), ("reference", "fp16")])
def test_tiny_decoder_vs_oracle(dev, mode, dt):
    from oracle import model as OM
    from stablediffusion_amd.vae import Decoder
    m, sd = build(Decoder, synth.TINY_DDCONFIG, "decoder.", dev, vae_attn=mode, compute_dtype=TORCH_DT[dt])
    zz = synth.synth_input("tiny_z", (2, 4, 16, 16))
    img = m(zz.to(dev))
    ref = OM.decoder_forward({"decoder." + k: v for k, v in sd.items()}, synth.TINY_DDCONFIG, zz, p="decoder.",
                             vae_attn=mode)
    assert img.shape == ref.shape == (2, 3, 32, 32)
    print(f"[parity] tiny decoder ({mode}, {dt}) vs oracle: rel_err {rel_err(img, ref):.3e}")
    assert rel_err(img, ref) < TOL[dt]
    if mode == "reference":
        gold = torch.from_numpy(np.load(os.path.join(GOLD, "tiny_decoder.npz"))["out"])
        assert rel_err(img, gold) < TOL[dt]


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_cfg_pairs_prefix_equals_full_batch(dev, dt):
    """UNetModel.forward(cfg_pairs=True): for a guidance batch cat([z, z]) / cat([t, t]) / cat([uc, c]) the layers in
    front of the first cross-attention run once per pair.  Same result as pushing every row through every layer, up to
    the fp32 summation order of GEMM plans that depend on the row count (the half batch runs the tiled kernel +
    sdmi_layernorm where the full batch runs the panel kernel with the LayerNorm fused): the two 16-bit runs differ by
    about their own rounding noise - and sit at the same distance from the fp32 oracle."""
    from oracle import model as OM
    from stablediffusion_amd.unet import UNetModel
    m, sd = build(UNetModel, policy(synth.SD2_UNET, dt), "unet.", dev)
    z = synth.synth_input("pairs_z", (2, 4, 64, 64))
    ctx = synth.synth_input("pairs_ctx", (4, 77, 1024))
    t = torch.tensor([761, 12])
    x_in, t_in = torch.cat([z, z]).to(dev), torch.cat([t, t]).to(dev)
    full = m(x_in, t_in, ctx.to(dev))
    pairs = m(x_in, t_in, ctx.to(dev), cfg_pairs=True)
    e = rel_err(pairs, full)
    print(f"[parity] sd2 unet h64 cfg_pairs vs full batch ({dt}): rel_err {e:.3e}")
    assert e < (4e-3 if dt == "fp16" else 1.2e-2)
    with torch.no_grad():
        ref = OM.unet_forward(sd, synth.SD2_UNET, z[1:], t[1:], ctx[3:])       # row 3 = (z[1], t[1], ctx[3])
    e_full, e_pairs = rel_err(full[3:], ref), rel_err(pairs[3:], ref)
    print(f"[parity] ... row 3 vs fp32 oracle: full {e_full:.3e}, cfg_pairs {e_pairs:.3e}")
    assert e_pairs < TOL[dt] and e_full < TOL[dt]


@pytest.mark.parametrize("mode", ["reference", "standard"])
def test_tiny_decoder_fp32_mode(dev, mode):
    """fp32 mode of the VAE decoder (compute_dtype = torch.float32: fp32 activations, contractions on exact three-way
    bf16 splits, sdmi_f32_attention_vae) against the fp32 oracle and the reference's golden at the fp32-mode gate."""
    from oracle import model as OM
    from stablediffusion_amd.vae import Decoder
    m, sd = build(Decoder, synth.TINY_DDCONFIG, "decoder.", dev, vae_attn=mode, compute_dtype=torch.float32)
    zz = synth.synth_input("tiny_z", (2, 4, 16, 16))
    img = m(zz.to(dev))
    ref = OM.decoder_forward({"decoder." + k: v for k, v in sd.items()}, synth.TINY_DDCONFIG, zz, p="decoder.",
                             vae_attn=mode)
    print(f"[parity] tiny decoder fp32 mode ({mode}) vs oracle: rel_err {rel_err(img, ref):.3e}")
    assert img.dtype == torch.float32 and img.shape == ref.shape
    assert rel_err(img, ref) < 1e-5
    if mode == "reference":
        gold = torch.from_numpy(np.load(os.path.join(GOLD, "tiny_decoder.npz"))["out"])
        assert rel_err(img, gold) < 1e-5


def test_sd2_decode_fp32_mode(dev):
    """AutoencoderKL.decode (post_quant_conv + the SD2 decoder, 512-channel mid attention) in fp32 mode on a 16x24
    latent, the 1/scale_factor fused into the layout kernel: against the fp32 oracle at the fp32-mode gate."""
    from oracle import model as OM
    from stablediffusion_amd.vae import AutoencoderKL
    vae, vsd = build(AutoencoderKL, dict(ddconfig=synth.SD2_DDCONFIG, embed_dim=4), "vae.", dev)
    vae.set_compute_dtype(torch.float32)
    z = synth.synth_input("f32_z", (1, 4, 16, 24))
    img = vae.decode(z.to(dev), 1 / 0.18215)
    with torch.no_grad():
        ref = OM.vae_decode(vsd, synth.SD2_DDCONFIG, z / 0.18215)
    e = rel_err(img, ref)
    print(f"[parity] sd2 decoder 16x24 latent, fp32 mode: rel_err vs fp32 oracle {e:.3e}")
    assert img.shape == (1, 3, 128, 192) and e < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_sd2_unet_h32_vs_reference_golden(dev, dt):
    """Full v2-inference.yaml UNet (866 M parameters), 32x32 latent: against values the reference's own
    layer code produced (tests/golden/sd2_full_h32.npz)."""
    from stablediffusion_amd.unet import UNetModel
    z = np.load(os.path.join(GOLD, "sd2_full_h32.npz"))
    m, _ = build(UNetModel, policy(synth.SD2_UNET, dt), "unet.", dev)
    x = synth.synth_input("sd2_x", (1, 4, 32, 32)).to(dev)
    ctx = synth.synth_input("sd2_ctx", (1, 77, 1024)).to(dev)
    y = m(x, torch.tensor([501], device=dev), ctx)
    sub = torch.from_numpy(z["unet_sub"])
    scale = float(z["unet_std"]) * 4       # ~max|y|
    print(f"[parity] sd2 unet h32 ({dt}) vs reference golden: sub-grid rel_err "
          f"{(y[0, :, ::4, ::4].cpu() - sub).abs().max().item() / sub.abs().max().item():.3e}")
    assert (y[0, :, ::4, ::4].cpu() - sub).abs().max().item() / sub.abs().max().item() < TOL[dt]
    assert (y.flatten()[:64].cpu() - torch.from_numpy(z["unet_first64"])).abs().max().item() < TOL[dt] * scale
    assert abs(y.std().item() - float(z["unet_std"])) < 1e-2 * float(z["unet_std"])


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_sd2_unet_h32_init8d_recipe_vs_reference_golden(dev, dt):
    """The weight recipe SURVEY.md §8(d) prescribes - default nn init under manual_seed(0), the zero_module'd tensors
    re-randomised from Generator(1) * 0.02 (tests/synth.py:init8d_state_dict) - on the full SD2 UNet, 32x32 latent,
    against what the reference's own layer code produced on those weights (tests/golden/sd2_init8d_h32.npz)."""
    from stablediffusion_amd.unet import UNetModel
    z = np.load(os.path.join(GOLD, "sd2_init8d_h32.npz"))
    sd = synth.init8d_state_dict(UNetModel, dict(synth.SD2_UNET, use_fp16=False))
    for k in z.files:
        if k.startswith("fp/"):
            np.testing.assert_array_equal(sd[k[3:]].flatten()[:4].numpy(), z[k], err_msg=f"recipe drifted: {k}")
    with torch.device("meta"):
        m = UNetModel(**policy(synth.SD2_UNET, dt))
    m.load_state_dict(sd, assign=True)
    m = m.to(dev).eval()
    x = synth.synth_input("sd2_x", (1, 4, 32, 32)).to(dev)
    ctx = synth.synth_input("sd2_ctx", (1, 77, 1024)).to(dev)
    y = m(x, torch.tensor([501], device=dev), ctx)
    sub = torch.from_numpy(z["unet_sub"])
    e = (y[0, :, ::4, ::4].cpu() - sub).abs().max().item() / sub.abs().max().item()
    print(f"[parity] sd2 unet h32, init8d recipe ({dt}) vs reference golden: sub-grid rel_err {e:.3e}")
    assert e < TOL[dt]
    assert (y.flatten()[:64].cpu() - torch.from_numpy(z["unet_first64"])).abs().max().item() < TOL[dt] * float(z["unet_absmax"])


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_sd2_decoder_h32_vs_reference_golden(dev, dt):
    from stablediffusion_amd.vae import Decoder
    z = np.load(os.path.join(GOLD, "sd2_full_h32.npz"))
    m, _ = build(Decoder, synth.SD2_DDCONFIG, "decoder.", dev, compute_dtype=TORCH_DT[dt])
    img = m(synth.synth_input("sd2_z", (1, 4, 32, 32)).to(dev))
    assert img.shape == (1, 3, 256, 256)
    sub = torch.from_numpy(z["dec_sub"])
    print(f"[parity] sd2 decoder h32 ({dt}) vs reference golden: sub-grid rel_err "
          f"{(img[0, :, ::16, ::16].cpu() - sub).abs().max().item() / sub.abs().max().item():.3e}")
    assert (img[0, :, ::16, ::16].cpu() - sub).abs().max().item() / sub.abs().max().item() < TOL[dt]


def _latent_diffusion(dev, unet_cfg, dd_cfg, parameterization="eps"):
    """Through the reference's factory surface: instantiate_from_config on a v2-inference.yaml-shaped dict
    (reference/upstream target strings are mapped onto this package)."""
    from stablediffusion_amd import instantiate_from_config
    cfg = {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.0120, "num_timesteps_cond": 1, "log_every_t": 200, "timesteps": 1000,
        "first_stage_key": "jpg", "cond_stage_key": "txt", "image_size": 64, "channels": 4,
        "cond_stage_trainable": False, "conditioning_key": "crossattn", "monitor": "val/loss_simple_ema",
        "scale_factor": 0.18215, "use_ema": False, "parameterization": parameterization,
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(unet_cfg)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL", "params": {
            "embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dict(dd_cfg),
            "lossconfig": {"target": "torch.nn.Identity"}}},
        "cond_stage_config": "__is_unconditional__", "force_null_conditioning": True}}
    with torch.device("meta"):
        model = instantiate_from_config(cfg)
    sd = {}
    for k, v in model.state_dict().items():
        sd[k] = synth.synth_tensor(k, v.shape)
    # buffers of the schedule are not persistent; parameters get the synthetic values
    model.model.diffusion_model.load_state_dict(
        {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}, assign=True)
    model.first_stage_model.load_state_dict(
        {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}, assign=True)
    model.register_schedule(None, "linear", 1000, 0.00085, 0.0120)
    return model.to(dev).eval(), sd


@pytest.mark.parametrize("param,dt", [("eps", "bf16"), ("v", "bf16"), ("eps", "fp16"), ("v", "fp16")])
def test_ddim_trajectory_256(dev, param, dt):
    """BASELINE config 0: SD2 UNet+VAE, random-init, 256^2 (32x32 latent), batch 1, 10 DDIM steps + CFG.
    Teacher-forced per-step error (the oracle's x_t fed to both), free-running final latent, decoded PSNR."""
    from oracle import sampler as OSamp
    from oracle import model as OM
    from stablediffusion_amd.ddim import DDIMSampler
    model, sd = _latent_diffusion(dev, policy(synth.SD2_UNET, dt), synth.SD2_DDCONFIG, param)
    assert model.first_stage_model.decoder.compute_dtype == TORCH_DT[dt]       # one policy for UNet and VAE
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    vae_sd = {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}
    g = torch.Generator().manual_seed(1234)
    x_T = torch.randn(1, 4, 32, 32, generator=g)
    c = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(4321))
    uc = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(99))
    scale = 9.0
    acp = model.alphas_cumprod.cpu().numpy()

    steps = []
    with torch.no_grad():
        fn = lambda x, t, cond: OSamp.apply_model(unet_sd, synth.SD2_UNET, x, t, cond)
        x_ref = OSamp.ddim_sample(fn, acp, 10, x_T, c, uc, scale, 0.0, param,
                                  record=lambda i, step, x, e_uc, e_c, xp: steps.append((step, x, e_uc, e_c, xp)))
    # teacher-forced: the HIP UNet sees exactly the oracle's x_t
    c_in = torch.cat([uc, c]).to(dev)
    worst, worst_emul = 0.0, 0.0
    for i, (step, x, e_uc, e_c, _) in enumerate(steps):
        t = torch.full((2,), step, device=dev, dtype=torch.long)
        out = model.apply_model(torch.cat([x, x]).to(dev), t, c_in)
        worst = max(worst, rel_err(out, torch.cat([e_uc, e_c])))
        if i % 3 == 0 and dt == "bf16":   # bf16-operand oracle on a subset of the steps (each costs a CPU forward)
            with torch.no_grad(), OM.operand_precision(BF16):
                emul = OSamp.apply_model(unet_sd, synth.SD2_UNET, torch.cat([x, x]), t.cpu(), torch.cat([uc, c]))
            worst_emul = max(worst_emul, rel_err(out, emul))
    print(f"[parity] ddim {param} {dt}: teacher-forced worst per-step rel_err vs fp32 oracle {worst:.3e}"
          + (f", vs bf16-operand oracle {worst_emul:.3e}" if dt == "bf16" else ""))
    assert worst < TOL[dt], f"teacher-forced per-step eps/v relative error {worst:.3e}"

    sampler = DDIMSampler(model)
    samples, _ = sampler.sample(S=10, batch_size=1, shape=[4, 32, 32], conditioning=c.to(dev), verbose=False,
                                unconditional_guidance_scale=scale, unconditional_conditioning=uc.to(dev), eta=0.0,
                                x_T=x_T.to(dev))
    free = rel_err(samples, x_ref)
    print(f"[parity] ddim {param} {dt}: free-running 10-step latent rel_err {free:.3e}")
    assert free < 5e-2, f"free-running 10-step latent relative error {free:.3e}"

    img = model.decode_first_stage(samples)
    with torch.no_grad():
        img_ref = OSamp.decode_first_stage(vae_sd, synth.SD2_DDCONFIG, x_ref, 0.18215)
        img_same_latent = OSamp.decode_first_stage(vae_sd, synth.SD2_DDCONFIG, samples.cpu(), 0.18215)
    assert img.shape == (1, 3, 256, 256)
    # decoder parity on identical latents, normalised to the image's own dynamic range
    peak = float(img_same_latent.max() - img_same_latent.min())
    print(f"[parity] ddim {param} {dt}: decode PSNR same-latent {psnr(img, img_same_latent, peak):.1f} dB, "
          f"end-to-end {psnr(img, img_ref, float(img_ref.max() - img_ref.min())):.1f} dB")
    assert psnr(img, img_same_latent, peak) >= 35.0
    assert psnr(img, img_ref, float(img_ref.max() - img_ref.min())) >= 35.0


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_plms_matches_oracle(dev, dt):
    """PLMS (absent from the reference: restated from upstream semantics, parity unpinned by the reference).
    (i) teacher-forced: the sampler's multistep arithmetic alone - every model evaluation is answered by the ORACLE's
    UNet, so the HIP side contributes only the guidance combine (sdmi_lincomb4), the Adams-Bashforth combinations and
    the x_prev updates (sdmi_ddim_step): the 10-step trajectory must then match the oracle's to fp32 rounding;
    (ii) free-running with the HIP UNet in the given dtype."""
    from oracle import sampler as OSamp
    from stablediffusion_amd.plms import PLMSSampler
    model, sd = _latent_diffusion(dev, policy(synth.TINY_UNET, dt), synth.TINY_DDCONFIG)
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    x_T = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    c = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(6))
    uc = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(7))
    acp = model.alphas_cumprod.cpu().numpy()
    with torch.no_grad():
        fn = lambda x, t, cond: OSamp.apply_model(unet_sd, synth.TINY_UNET, x, t, cond)
        ref = OSamp.plms_sample(fn, acp, 10, x_T, c, uc, 5.0)
    kw = dict(S=10, batch_size=2, shape=[4, 16, 16], conditioning=c.to(dev), verbose=False,
              unconditional_guidance_scale=5.0, unconditional_conditioning=uc.to(dev), x_T=x_T.to(dev))
    if dt == HEADLINE:
        forced = PLMSSampler(model)
        with torch.no_grad():
            forced._apply = lambda x, t, cond, **kw: fn(x.cpu(), t.cpu(), cond.cpu()).to(dev)
            out_tf, _ = forced.sample(**kw)
        e_tf = rel_err(out_tf, ref)
        print(f"[parity] plms tiny 10-step, model evaluations teacher-forced by the oracle: latent rel_err {e_tf:.3e}")
        assert e_tf < 1e-5
    out, _ = PLMSSampler(model).sample(**kw)
    print(f"[parity] plms tiny 10-step free-running ({dt}) latent rel_err {rel_err(out, ref):.3e}")
    assert rel_err(out, ref) < (5e-2 if dt == "bf16" else 1e-2)


def test_plms_hip_graph_without_cfg_matches_eager(dev):
    """Without guidance eps_at() used to hand back the UNet output tensor itself; under hip_graph=True that is the
    graph's static output buffer, which the second evaluation of step 0 and every later replay overwrite while the
    multistep history still points at it.  Graph replay and eager launches must agree bit for bit."""
    from stablediffusion_amd.plms import PLMSSampler
    model, _ = _latent_diffusion(dev, policy(synth.TINY_UNET, "fp16"), synth.TINY_DDCONFIG)
    x_T = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    c = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(6)).to(dev)
    kw = dict(S=8, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, unconditional_guidance_scale=1.0,
              x_T=x_T)
    eager, _ = PLMSSampler(model).sample(**kw)
    graphed, _ = PLMSSampler(model, hip_graph=True).sample(**kw)
    assert torch.equal(eager, graphed)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_unet_h64_batch2_vs_oracle(dev, dt):
    """BASELINE config 1 shape: one CFG step (UNet batch 2) at a 64x64 latent, against the oracle run on
    the host cores of the GPU box."""
    from oracle import model as OM
    from stablediffusion_amd.unet import UNetModel
    m, sd = build(UNetModel, policy(synth.SD2_UNET, dt), "unet.", dev)
    x = synth.synth_input("h64_x", (2, 4, 64, 64))
    ctx = synth.synth_input("h64_ctx", (2, 77, 1024))
    t = torch.tensor([981, 981])
    y = m(x.to(dev), t.to(dev), ctx.to(dev))
    with torch.no_grad():
        ref = OM.unet_forward(sd, synth.SD2_UNET, x, t, ctx)
        with OM.operand_precision(TORCH_DT[dt]):
            emul = OM.unet_forward(sd, synth.SD2_UNET, x, t, ctx)
    print(f"[parity] sd2 unet h64 batch 2 ({dt}): rel_err vs fp32 oracle {rel_err(y, ref):.3e} "
          f"(format floor = oracle with {dt}-rounded operands vs fp32 oracle: {rel_err(emul, ref):.3e})")
    assert rel_err(y, ref) < TOL[dt]
    # BASELINE config 2 at its size: UNet batch 16 (8 images per GPU x CFG) at 64x64 - the benchmarked launch shapes
    # (256-row tiles, other split-K plans than batch 2).  Samples of a batch are independent, so rows 0/1 and 14/15
    # of a batch-16 forward that carry the batch-2 inputs must reproduce the batch-2 result that was just checked
    # against the oracle, up to the fp32 summation order; the rows in between hold other latents / prompts / steps.
    fill_x = synth.synth_input("h64_fill_x", (12, 4, 64, 64))
    fill_c = synth.synth_input("h64_fill_ctx", (12, 77, 1024))
    x16 = torch.cat([x, fill_x, x]).to(dev)
    c16 = torch.cat([ctx, fill_c, ctx]).to(dev)
    t16 = torch.cat([t, torch.arange(12) * 80 + 1, t]).to(dev)
    y16 = m(x16, t16, c16)
    assert y16.shape == (16, 4, 64, 64) and bool(torch.isfinite(y16).all())
    for rows in ((0, 2), (14, 16)):
        e_or = rel_err(y16[rows[0]:rows[1]], ref)
        e_b2 = rel_err(y16[rows[0]:rows[1]], y)
        print(f"[parity] sd2 unet h64 batch 16 ({dt}) rows {rows[0]}..{rows[1] - 1}: rel_err vs fp32 oracle {e_or:.3e}, "
              f"vs the batch-2 forward {e_b2:.3e}")
        assert e_or < TOL[dt] and e_b2 < TOL[dt]


def test_unet_768_v_prediction_vs_oracle(dev):
    """BASELINE config 3 shape: SD2.1-v 768^2 = 96x96 latent, 9216-token self-attention, v-prediction head
    (same UNet; parameterization only changes how the sampler reads the output).  One forward, batch 1, in the
    headline dtype (fp16, what v2-inference-v.yaml ships) and in bf16, against the fp32 oracle on the host cores."""
    from oracle import model as OM
    from stablediffusion_amd.unet import UNetModel
    x = synth.synth_input("h96_x", (1, 4, 96, 96))
    ctx = synth.synth_input("h96_ctx", (1, 77, 1024))
    t = torch.tensor([741])
    ref = None
    for dt in ("fp16", "bf16"):
        m, sd = build(UNetModel, policy(synth.SD2_UNET, dt), "unet.", dev)
        y = m(x.to(dev), t.to(dev), ctx.to(dev))
        assert y.shape == (1, 4, 96, 96) and bool(torch.isfinite(y).all())
        if ref is None:
            with torch.no_grad():
                ref = OM.unet_forward(sd, synth.SD2_UNET, x, t, ctx)
        print(f"[parity] sd2 unet h96 (768^2, 9216 tokens) {dt}: rel_err vs fp32 oracle {rel_err(y, ref):.3e}")
        assert rel_err(y, ref) < TOL[dt]
        del m, sd


def test_non_square_latents_vs_oracle(dev):
    """--H / --W need not be equal (txt2img.py:147-158): a 40x24 latent (320x192 image) through the SD2 UNet and a
    24x40 latent through the SD2 decoder - row counts that are not multiples of the 128/256-row tiles, attention
    lengths 960 / 240 / 60 / 15 that are ragged against the 64-key tile - against the fp32 oracle, fp16 operands."""
    from oracle import model as OM
    from stablediffusion_amd.unet import UNetModel
    from stablediffusion_amd.vae import AutoencoderKL
    m, sd = build(UNetModel, policy(synth.SD2_UNET, "fp16"), "unet.", dev)
    x = synth.synth_input("ns_x", (2, 4, 40, 24))
    ctx = synth.synth_input("ns_ctx", (2, 77, 1024))
    t = torch.tensor([17, 903])
    y = m(x.to(dev), t.to(dev), ctx.to(dev))
    with torch.no_grad():
        ref = OM.unet_forward(sd, synth.SD2_UNET, x, t, ctx)
    e = rel_err(y, ref)
    print(f"[parity] sd2 unet 40x24 latent fp16: rel_err vs fp32 oracle {e:.3e}")
    assert y.shape == (2, 4, 40, 24) and e < NORTH_STAR
    vae, vsd = build(AutoencoderKL, dict(ddconfig=synth.SD2_DDCONFIG, embed_dim=4), "vae.", dev)
    vae.set_compute_dtype(torch.float16)
    z = synth.synth_input("ns_z", (1, 4, 24, 40))
    img = vae.decode(z.to(dev), 1 / 0.18215)
    with torch.no_grad():
        ref = OM.vae_decode(vsd, synth.SD2_DDCONFIG, z / 0.18215)
    e = rel_err(img, ref)
    print(f"[parity] sd2 decoder 24x40 latent fp16: rel_err vs fp32 oracle {e:.3e}")
    assert img.shape == (1, 3, 192, 320) and e < NORTH_STAR


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_decoder_768_batch16(dev, dt):
    """BASELINE config 4 at its size: decode_first_stage standalone at 768^2, batch 16.  (i) Images of a batch are
    independent, so a batch-16 decode must give what decoding a latent alone gives, up to the fp32 summation order of
    the tile / split-K plan (which depends on M); (ii) in the headline dtype, image 0 against the fp32 oracle run on
    the host cores (one 5.75 TFLOP CPU decode), compared on a strided sub-grid and as PSNR."""
    from stablediffusion_amd.vae import AutoencoderKL
    vae, vsd = build(AutoencoderKL, dict(ddconfig=synth.SD2_DDCONFIG, embed_dim=4), "vae.", dev)
    vae.set_compute_dtype(TORCH_DT[dt])
    z = synth.synth_input("z768", (16, 4, 96, 96))
    full = vae.decode(z.to(dev), 1 / 0.18215)
    assert full.shape == (16, 3, 768, 768) and bool(torch.isfinite(full).all())
    worst = 0.0
    for i in (0, 7, 15):
        one = vae.decode(z[i:i + 1].contiguous().to(dev), 1 / 0.18215)
        worst = max(worst, rel_err(full[i:i + 1], one))
    print(f"[parity] sd2 decoder 768^2 batch 16 vs per-image ({dt}): rel_err {worst:.3e}")
    assert worst < TOL[dt]      # rounding noise of the operand format, re-ordered by a different tile plan
    if dt == HEADLINE:
        from oracle import model as OM
        with torch.no_grad():
            ref = OM.vae_decode(vsd, synth.SD2_DDCONFIG, z[:1] / 0.18215)
        got = full[:1].cpu()
        e_sub = rel_err(got[:, :, ::8, ::8], ref[:, :, ::8, ::8])
        e_all = rel_err(got, ref)
        db = psnr(got, ref, float(ref.max() - ref.min()))
        print(f"[parity] sd2 decoder 768^2 batch 16, image 0 ({dt}) vs fp32 oracle: sub-grid rel_err {e_sub:.3e}, "
              f"full rel_err {e_all:.3e}, PSNR {db:.1f} dB")
        assert e_sub < NORTH_STAR and e_all < NORTH_STAR and db >= 35.0


def test_upstream_spelled_checkpoint_through_txt2img_loader(dev, tmp_path):
    """SURVEY.md §8(f) rank 2 on the GPU: a Lightning-style checkpoint ({"state_dict", "global_step"}, txt2img.py:32-37)
    whose keys use the UPSTREAM spellings (time_embed, emb_layers, op, proj_in, ff, upsample - SURVEY §0.5) goes
    through txt2img.load_model_from_config (txt2img.py:21-64; non-executing reader, strict=False) with
    vae_attn="standard" (what real SD weights need); UNet step and decode then match the oracle on those weights."""
    from oracle import model as OM
    from oracle import sampler as OSamp
    from stablediffusion_amd import txt2img
    cfg = {"model": {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.012, "timesteps": 1000, "scale_factor": 0.18215,
        "conditioning_key": "crossattn", "parameterization": "eps",
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.TINY_UNET)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                               "params": {"embed_dim": 4, "ddconfig": dict(synth.TINY_DDCONFIG), "vae_attn": "standard"}},
        "cond_stage_config": "__is_unconditional__", "force_null_conditioning": True}}}
    with torch.device("meta"):
        shapes = {k: v.shape for k, v in txt2img.instantiate_from_config(cfg["model"]).state_dict().items()}
    ups = [("time_embedding.", "time_embed."), (".embedding_layers.", ".emb_layers."), (".operation.", ".op."),
           (".project_in.", ".proj_in."), (".feed_forward.net.", ".ff.net."), (".up_sample.", ".upsample.")]

    def upstream(k):
        for a, b in ups:
            k = k.replace(a, b)
        return k
    sd_up = {upstream(k): synth.synth_tensor(k, s) for k, s in shapes.items()}
    assert any(".emb_layers." in k for k in sd_up) and any(".proj_in." in k for k in sd_up) and \
        any(".op." in k for k in sd_up) and any(".ff.net." in k for k in sd_up) and any(".upsample." in k for k in sd_up) \
        and not any(".embedding_layers." in k or ".project_in." in k or ".up_sample." in k for k in sd_up)
    ckpt = tmp_path / "upstream.ckpt"
    torch.save({"state_dict": sd_up, "global_step": 7}, ckpt)
    model = txt2img.load_model_from_config(cfg, str(ckpt), device=dev)
    assert model.first_stage_model.decoder.vae_attn == "standard"
    # nothing was left at its constructor value: every parameter equals the checkpoint's tensor
    got = model.state_dict()
    for k, s_ in shapes.items():
        assert torch.equal(got[k].cpu(), sd_up[upstream(k)]), k
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd_up.items() if k.startswith("model.diffusion_model.")}
    vae_sd = {k[len("first_stage_model."):]: v for k, v in sd_up.items() if k.startswith("first_stage_model.")}
    x = synth.synth_input("ck_x", (2, 4, 16, 16))
    ctx = synth.synth_input("ck_ctx", (2, 7, 96))
    t = torch.tensor([801, 41])
    y = model.apply_model(x.to(dev), t.to(dev), ctx.to(dev))
    img = model.decode_first_stage(x.to(dev))
    with torch.no_grad():
        ref = OSamp.apply_model(unet_sd, synth.TINY_UNET, x, t, ctx)          # the oracle reads upstream spellings too
        iref = OSamp.decode_first_stage(vae_sd, synth.TINY_DDCONFIG, x, 0.18215, vae_attn="standard")
        iref_literal = OSamp.decode_first_stage(vae_sd, synth.TINY_DDCONFIG, x, 0.18215, vae_attn="reference")
    print(f"[parity] upstream-spelled checkpoint (fp16, vae_attn=standard): unet rel_err {rel_err(y, ref):.3e}, "
          f"decode rel_err {rel_err(img, iref):.3e} (vs the literal-reference contraction {rel_err(img, iref_literal):.3e})")
    assert rel_err(y, ref) < NORTH_STAR and rel_err(img, iref) < NORTH_STAR
    assert rel_err(img, iref_literal) > 5 * rel_err(img, iref)       # the switch really selects the other contraction


def test_txt2img_main_end_to_end(dev, tmp_path):
    """The completed txt2img loop on the tiny config: config -> model -> DDIM/PLMS sampling with CFG -> decode ->
    PNGs and a grid, deterministic for a fixed seed."""
    import yaml
    from PIL import Image
    from stablediffusion_amd import txt2img
    cfg = {"model": {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.012, "timesteps": 1000, "scale_factor": 0.18215,
        "conditioning_key": "crossattn",
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.TINY_UNET)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                               "params": {"embed_dim": 4, "ddconfig": dict(synth.TINY_DDCONFIG)}},
        "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                              "params": {"width": 64, "heads": 1, "layers": 2, "vocab_size": 512}}}}}
    (tmp_path / "tiny.yaml").write_text(yaml.safe_dump(cfg))
    grids = []
    for run, extra in enumerate(([], [], ["--plms", "--seed", "12"])):
        out = tmp_path / f"out{run}"
        args = txt2img.parse_args(["--config", str(tmp_path / "tiny.yaml"), "--out_dir", str(out), "--steps", "4",
                                   "--n_iter", "1", "--n_samples", "2", "--height", "32", "--width", "32",
                                   "--factor", "2", "--scale", "5.0", "--prompt", "a test prompt", "--seed", "11"] + extra)
        grids.append(txt2img.main(args))
        files = sorted(os.listdir(out / "samples"))
        assert files == ["00000.png", "00001.png"]
        assert Image.open(out / "samples" / "00000.png").size == (32, 32)
        assert any(f.startswith("grid-") for f in os.listdir(out))
    assert grids[0].shape == (2, 32, 32, 3) and grids[0].dtype == torch.uint8
    assert torch.equal(grids[0], grids[1])          # same seed, same images (the HIP path is deterministic)
    assert not torch.equal(grids[0], grids[2])      # another seed (and the PLMS loop): other images


def test_hip_graph_replay_is_bit_identical(dev):
    """DDIMSampler(hip_graph=True) replays the captured UNet step: same latents, bit for bit, as eager launches,
    across steps (different timesteps / inputs through the static buffers) and across two sample() calls."""
    from stablediffusion_amd.ddim import DDIMSampler
    model, _ = _latent_diffusion(dev, policy(synth.TINY_UNET, "bf16"), synth.TINY_DDCONFIG)
    x_T = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    c = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(6)).to(dev)
    uc = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(7)).to(dev)
    kw = dict(S=6, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, unconditional_guidance_scale=4.0,
              unconditional_conditioning=uc, eta=0.0)
    eager, _ = DDIMSampler(model).sample(x_T=x_T, **kw)
    g = DDIMSampler(model, hip_graph=True)
    a, _ = g.sample(x_T=x_T, **kw)
    b, _ = g.sample(x_T=x_T * 0.5, **kw)
    b_eager, _ = DDIMSampler(model).sample(x_T=x_T * 0.5, **kw)
    assert torch.equal(a, eager) and torch.equal(b, b_eager) and not torch.equal(a, b)
    # a new prompt does not re-capture: the conditioning is a static input of the graph and the UNet's cross-attention
    # K / V are re-projected into the tensor the captured launches read (UNetModel.refresh_context_kv)
    graph_obj = g._apply._graph
    c2 = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(8)).to(dev)
    kw2 = dict(kw, conditioning=c2)
    d, _ = g.sample(x_T=x_T, **kw2)
    d_eager, _ = DDIMSampler(model).sample(x_T=x_T, **kw2)
    assert g._apply._graph is graph_obj
    assert torch.equal(d, d_eager) and not torch.equal(d, a)
    # ... and the first prompt again, after the eager sampler above replaced the UNet's own cache entry
    a2, _ = g.sample(x_T=x_T, **kw)
    assert g._apply._graph is graph_obj and torch.equal(a2, eager)
    # a prompt edited in place (same tensor, new version) is seen as well
    c2.mul_(0.5)
    e, _ = g.sample(x_T=x_T, **kw2)
    e_eager, _ = DDIMSampler(model).sample(x_T=x_T, **kw2)
    assert torch.equal(e, e_eager) and not torch.equal(e, d)


FP32_GATE = 1e-5       # BASELINE north_star: per-step eps/v max relative error <= 1e-5 in fp32 mode


def _fp32_policy(cfg):
    return dict(cfg, use_fp16=False, use_bf16=False)


def test_fp32_mode_tiny_unet_vs_reference_golden(dev):
    """fp32 mode (use_fp16 = use_bf16 = False: fp32 activations, contractions on the bf16 MFMA kernels through the
    exact 3-piece split) against the golden the reference's own fp32 layer code produced."""
    from stablediffusion_amd.unet import UNetModel
    z = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    m, _ = build(UNetModel, _fp32_policy(synth.TINY_UNET), "unet.", dev)
    assert m.compute_dtype == torch.float32
    x, t, ctx = (torch.from_numpy(z[k]).to(dev) for k in ("x", "t", "ctx"))
    y = m(x, t, ctx)
    e = rel_err(y, torch.from_numpy(z["out"]))
    print(f"[parity] tiny unet (fp32 mode) vs reference golden: rel_err {e:.3e}")
    assert e < FP32_GATE


def test_fp32_mode_sd2_unet_h32_vs_reference_golden(dev):
    """The full SD2 UNet at a 32x32 latent in fp32 mode against the reference golden (sub-sampled output)."""
    from stablediffusion_amd.unet import UNetModel
    z = np.load(os.path.join(GOLD, "sd2_full_h32.npz"))
    m, _ = build(UNetModel, _fp32_policy(synth.SD2_UNET), "unet.", dev)
    x = synth.synth_input("sd2_x", (1, 4, 32, 32)).to(dev)
    ctx = synth.synth_input("sd2_ctx", (1, 77, 1024)).to(dev)
    t = torch.tensor([501], device=dev)
    y = m(x, t, ctx)
    sub = torch.from_numpy(z["unet_sub"])
    e = (y[0, :, ::4, ::4].cpu() - sub).abs().max().item() / sub.abs().max().item()
    print(f"[parity] sd2 unet h32 (fp32 mode) vs reference golden: rel_err {e:.3e}")
    assert e < FP32_GATE


def test_fp32_mode_ddim_steps_vs_oracle(dev):
    """Teacher-forced DDIM steps with CFG in fp32 mode (tiny model, eps and v heads share the UNet): per-step
    eps/v prediction against the fp32 oracle."""
    from oracle import sampler as OSamp
    model, sd = _latent_diffusion(dev, _fp32_policy(synth.TINY_UNET), synth.TINY_DDCONFIG)
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    x_T = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    c = torch.randn(1, 7, 96, generator=torch.Generator().manual_seed(6))
    uc = torch.randn(1, 7, 96, generator=torch.Generator().manual_seed(7))
    acp = model.alphas_cumprod.cpu().numpy()
    steps = []
    with torch.no_grad():
        fn = lambda x, t, cond: OSamp.apply_model(unet_sd, synth.TINY_UNET, x, t, cond)
        OSamp.ddim_sample(fn, acp, 6, x_T, c, uc, 5.0, 0.0, "eps",
                          record=lambda i, step, x, e_uc, e_c, xp: steps.append((step, x, e_uc, e_c)))
    c_in = torch.cat([uc, c]).to(dev)
    worst = 0.0
    for step, x, e_uc, e_c in steps:
        t = torch.full((2,), step, device=dev, dtype=torch.long)
        out = model.apply_model(torch.cat([x, x]).to(dev), t, c_in)
        worst = max(worst, rel_err(out, torch.cat([e_uc, e_c])))
    print(f"[parity] ddim fp32 mode: teacher-forced worst per-step rel_err vs fp32 oracle {worst:.3e}")
    assert worst < FP32_GATE


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_tiny_encoder_vs_reference_golden(dev, dt):
    """VAE Encoder (asymmetric-pad stride-2 DownSample, mid attention) and
    DiagonalGaussianDistribution against what the reference's own code produced."""
    from stablediffusion_amd.vae import Encoder, DiagonalGaussianDistribution
    z = np.load(os.path.join(GOLD, "tiny_encoder.npz"))
    enc, _ = build(Encoder, dict(synth.TINY_DDCONFIG, compute_dtype=TORCH_DT[dt]), "encoder.", dev)
    m = enc(torch.from_numpy(z["x"]).to(dev))                     # NHWC 16-bit
    assert m.shape == (2, 16, 16, 8)
    e = rel_err(m.permute(0, 3, 1, 2), torch.from_numpy(z["moments"]))
    print(f"[parity] tiny encoder ({dt}) vs reference golden: rel_err {e:.3e}")
    assert e < TOL[dt]
    g = DiagonalGaussianDistribution(torch.from_numpy(z["params"]).to(dev))
    for name, t in (("mean", g.mean), ("log_var", g.log_var), ("std", g.std), ("var", g.var),
                    ("sample", g.sample(torch.from_numpy(z["noise"]).to(dev)))):
        np.testing.assert_allclose(t.cpu().numpy(), z[name], rtol=2e-6, atol=1e-7, err_msg=name)
    assert torch.equal(g.mode(), g.mean)
    det = DiagonalGaussianDistribution(torch.from_numpy(z["params"]).to(dev), deterministic=True)
    assert float(det.std.abs().max()) == 0 and torch.equal(det.sample(), det.mean)


def test_autoencoder_encode_decode_roundtrip_runs(dev):
    """AutoencoderKL.encode -> LatentDiffusion.get_first_stage_encoding -> decode_first_stage: moments against the
    oracle's vae_encode, shapes and scale factor through the img2img entry points."""
    from oracle import model as OM
    model, sd = _latent_diffusion(dev, policy(synth.TINY_UNET, "fp16"), synth.TINY_DDCONFIG)
    vae_sd = {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}
    x = synth.synth_input("rt_img", (2, 3, 32, 32)).clamp(-1, 1)
    post = model.encode_first_stage(x.to(dev))
    with torch.no_grad():
        ref = OM.vae_encode(vae_sd, synth.TINY_DDCONFIG, x)
    e = rel_err(post.parameters, ref)
    print(f"[parity] tiny AutoencoderKL.encode moments (fp16) vs oracle: rel_err {e:.3e}")
    assert post.parameters.shape == (2, 8, 16, 16) and e < NORTH_STAR
    zlat = model.get_first_stage_encoding(post, noise=torch.zeros(2, 4, 16, 16, device=dev))
    assert torch.allclose(zlat, post.mean * float(model.scale_factor), rtol=1e-6, atol=1e-7)
    img = model.decode_first_stage(zlat)
    assert img.shape == (2, 3, 32, 32) and bool(torch.isfinite(img).all())


@pytest.mark.parametrize("variant", ["conv", "linear"])
def test_spatial_transformer_d64_vs_reference_golden(dev, variant):
    """SpatialTransformer with conv-1x1 project_in / proj_out (use_linear=False, attention.py:213-226) and with nn.Linear
    ones, head dim 64, against outputs of the REFERENCE's own module (tests/golden/st_d64.npz, tools/gen_golden.py
    --only-st-d64; B3 alias only).  On NHWC activations both variants are the same [C, C] GEMM - this pins that."""
    from stablediffusion_amd import unet as U, ops
    z = np.load(os.path.join(GOLD, "st_d64.npz"))
    name = f"spatial_transformer_{variant}_d64"
    st = U.SpatialTransformer(128, 2, 64, depth=1, context_dim=24, use_linear=(variant == "linear"))
    pre = name + "/sd/"
    st.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    st = st.to(dev)
    cd = torch.float16
    e, kv_w, _ = U.pack_spatial_transformer(st, cd)
    x, ctx = torch.from_numpy(z["in/img"]).to(dev), torch.from_numpy(z["in/ctx"]).to(dev)
    kv = ops.linear(ops.cast_f32(ctx.reshape(-1, ctx.shape[-1]).contiguous(), cd), ops.pack_linear_cat(kv_w, dtype=cd))
    out = U.run_spatial_transformer(e, ops.nchw_to_nhwc(x, dtype=cd), kv, ctx.shape[1])
    err = rel_err(out.permute(0, 3, 1, 2), torch.from_numpy(z[name + "/out"]))
    print(f"[parity] SpatialTransformer {variant} d64 fp16 vs reference golden: rel_err {err:.3e}")
    assert err < NORTH_STAR


def test_time_embedding_table_replaces_time_path(dev):
    """The per-schedule hoist of the timestep path (openai_model.py:956-964, 280): a row of
    UNetModel.time_embedding_table() handed to forward() as time_emb gives what forward() computes from time_steps itself
    (fp32 table; the only difference is the summation order of a GEMM plan that depends on the row count), samplers with and
    without the hoist agree, and the hipGraph replay takes the row as a static input."""
    from stablediffusion_amd.ddim import DDIMSampler
    model, _ = _latent_diffusion(dev, policy(synth.TINY_UNET, "fp16"), synth.TINY_DDCONFIG)
    unet = model.model.diffusion_model
    steps = torch.tensor([981, 501, 21, 1])
    table = unet.time_embedding_table(steps)
    assert table.dtype == torch.float32 and table.shape[0] == 4
    x = synth.synth_input("tiny_x4", (4, 4, 16, 16)).to(dev)
    ctx = synth.synth_input("tiny_ctx4", (4, 7, 96)).to(dev)
    for i, s_ in enumerate(steps.tolist()):
        t = torch.full((4,), s_, device=dev, dtype=torch.long)
        own = unet(x, t, ctx)
        hoisted = unet(x, t, ctx, time_emb=table[i])
        assert rel_err(hoisted, own) < 2e-3, (s_, rel_err(hoisted, own))
        rows = unet(x, t, ctx, time_emb=table[i].unsqueeze(0).expand(4, -1).contiguous())
        assert torch.equal(rows, hoisted)
    with pytest.raises(Exception):
        unet(x, torch.full((4,), 1, device=dev, dtype=torch.long), ctx, time_emb=table[0][:-8])
    x_T = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    c = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(6)).to(dev)
    uc = torch.randn(2, 7, 96, generator=torch.Generator().manual_seed(7)).to(dev)
    kw = dict(S=10, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, unconditional_guidance_scale=5.0,
              unconditional_conditioning=uc, x_T=x_T)
    hoist = DDIMSampler(model)
    a, _ = hoist.sample(**kw)
    assert hoist._time_table is not None and hoist._time_table[0].shape[0] == 10
    b, _ = DDIMSampler(model, hoist_time=False).sample(**kw)
    g, _ = DDIMSampler(model, hip_graph=True).sample(**kw)
    print(f"[parity] ddim tiny 10-step, time path hoisted vs per step: latent rel_err {rel_err(a, b):.3e}")
    assert rel_err(a, b) < 5e-3
    assert torch.equal(a, g)


@pytest.mark.slow
def test_ddim_50_steps_512_free_running(dev):
    """BASELINE.json configs[1] IN FULL: SD2 UNet + VAE, random-init, 512^2 (64x64 latent), batch 1, 50 DDIM steps with
    classifier-free guidance 9.0, eta 0, fp16 - the HIP sampler (cfg_pairs prefix, hoisted time path: what bench.py
    times) free-running against the CPU oracle on the same x_T / embeddings / weights.  SURVEY.md section 7 "hard part (d)":
    error accumulation over 50 recurrent steps.  Semantics of the loop: ddim.py:105-249 as restated.
    Prints the latent drift every 10 steps; gates: final latent relative error and decoded-image PSNR >= 35 dB."""
    from oracle import sampler as OSamp
    from stablediffusion_amd.ddim import DDIMSampler
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)))
    model, sd = _latent_diffusion(dev, policy(synth.SD2_UNET, "fp16"), synth.SD2_DDCONFIG, "eps")
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    vae_sd = {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}
    x_T = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1234))
    c = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(4321))
    uc = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(99))
    scale, S = 9.0, 50
    acp = model.alphas_cumprod.cpu().numpy()
    ref_x = {}
    with torch.no_grad():
        fn = lambda x, t, cond: OSamp.apply_model(unet_sd, synth.SD2_UNET, x, t, cond)
        x_ref = OSamp.ddim_sample(fn, acp, S, x_T, c, uc, scale, 0.0, "eps",
                                  record=lambda i, step, x, e_uc, e_c, xp: ref_x.__setitem__(i, xp.clone()))
    sampler = DDIMSampler(model)
    samples, inter = sampler.sample(S=S, batch_size=1, shape=[4, 64, 64], conditioning=c.to(dev), verbose=False,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc.to(dev), eta=0.0,
                                    x_T=x_T.to(dev), log_every_t=10)
    # x_inter = [x_T] + the latent after every step whose table index is 49 (the first) or a multiple of 10, i.e. after
    # loop iterations i = 0, 9, 19, 29, 39, 49 - the oracle's record i holds its x_prev of the same iteration
    logged = [i for i in range(S) if (S - i - 1) % 10 == 0 or i == 0]
    assert len(inter["x_inter"]) == 1 + len(logged)
    for i, xi in zip(logged, inter["x_inter"][1:]):
        print(f"[parity] ddim-50 512^2 fp16 free-running: after step {i + 1:2d}/50 latent rel_err {rel_err(xi, ref_x[i]):.3e}")
    free = rel_err(samples, x_ref)
    img = model.decode_first_stage(samples)
    with torch.no_grad():
        img_ref = OSamp.decode_first_stage(vae_sd, synth.SD2_DDCONFIG, x_ref, 0.18215)
    peak = float(img_ref.max() - img_ref.min())
    p = psnr(img, img_ref, peak)
    print(f"[parity] ddim-50 512^2 fp16 free-running: final latent rel_err {free:.3e}, decoded image PSNR {p:.1f} dB "
          f"(peak-to-peak {peak:.3f})")
    assert img.shape == (1, 3, 512, 512)
    assert free < NORTH_STAR, f"free-running 50-step latent relative error {free:.3e}"       # measured 2.6e-3
    assert p >= 35.0, f"decoded image PSNR {p:.1f} dB"
