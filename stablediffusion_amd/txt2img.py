"""txt2img surface of the reference (scripts/txt2img.py) on the HIP path.

Mirrors ``load_model_from_config`` (txt2img.py:21-64) and the command line (txt2img.py:67-243: same option names and
defaults), and completes what the reference's ``main`` stops short of (it ends after the TorchScript/IPEX set-up,
txt2img.py:262-407): the sampling loop  conditioning -> sampler.sample -> decode_first_stage -> uint8 -> PNG.

Differences, all forced by the scope of the path:
  * checkpoints are read with loaders that execute nothing from the file (``torch.load(weights_only=True)`` or
    safetensors); upstream key spellings are renamed on load (unet.rename_upstream_keys);
  * the text encoder is out of scope (absent from the reference, weights not available offline): prompts are turned
    into conditioning by ``model.get_learned_conditioning`` when the config provides a cond stage, otherwise by
    ``--cond_file`` (an .npz / .pt holding ``c`` [N, 77, context_dim] and optionally ``uc``, or BPE token ids
    ``tokens`` [N, 77] + ``uc_tokens`` [1, 77] for the text tower of ``encoders.py``) or, failing that, by a
    deterministic synthetic embedding seeded from the prompt string (announced on stdout - for smoke runs only);
  * ``--device`` must be cuda (the HIP path has no CPU fallback); ``--torchscript`` / ``--ipex`` are Intel-CPU
    paths of the reference and raise; ``--dpm`` raises (DPM-Solver is absent from the reference's samplers);
  * ``--bf16`` selects bf16 MFMA operands (use_bf16) instead of the config's fp16.
"""
import argparse
import os
import zlib
from itertools import islice

import numpy as np
import torch

from .ddim import DDIMSampler
from .plms import PLMSSampler
from .util import instantiate_from_config
from . import ops

# The reference's v2-inference.yaml (configs/stable-diffusion/v2-inference.yaml) as a dict: used when --config is
# not given, so the script runs without a YAML parser round trip.
DEFAULT_CONFIG = {"model": {
    "target": "ldm.models.diffusion.ddpm.LatentDiffusion",
    "params": {
        "linear_start": 0.00085, "linear_end": 0.0120, "num_timesteps_cond": 1, "log_every_t": 200, "timesteps": 1000,
        "first_stage_key": "jpg", "cond_stage_key": "txt", "image_size": 64, "channels": 4,
        "cond_stage_trainable": False, "conditioning_key": "crossattn", "monitor": "val/loss_simple_ema",
        "scale_factor": 0.18215, "use_ema": False,
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": {
            "use_checkpoint": True, "use_fp16": True, "image_size": 32, "in_channels": 4, "out_channels": 4,
            "model_channels": 320, "attention_resolutions": [4, 2, 1], "num_res_blocks": 2,
            "channel_mult": [1, 2, 4, 4], "num_head_channels": 64, "use_spatial_transformer": True,
            "use_linear_in_transformer": True, "transformer_depth": 1, "context_dim": 1024, "legacy": False}},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL", "params": {
            "embed_dim": 4, "monitor": "val/rec_loss",
            "ddconfig": {"double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": 128,
                         "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0},
            "lossconfig": {"target": "torch.nn.Identity"}}},
        "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                              "params": {"freeze": True, "layer": "penultimate"}},
    }}}


def load_config(path):
    """YAML -> plain dict (the reference uses OmegaConf.load, txt2img.py:262; only ``config.model`` is read)."""
    if path is None:
        return DEFAULT_CONFIG
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh)


def _read_checkpoint(ckpt):
    """state_dict from a .ckpt/.pt (``weights_only=True``: nothing in the file is executed) or a .safetensors."""
    if str(ckpt).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(str(ckpt), device="cpu"), None
    pl_sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    step = pl_sd.get("global_step") if isinstance(pl_sd, dict) else None
    return (pl_sd["state_dict"] if "state_dict" in pl_sd else pl_sd), step


def load_model_from_config(config, ckpt, device=torch.device("cuda"), verbose=False):
    """txt2img.py:21-64: instantiate ``config.model`` (a dict or an attribute-style object), load the checkpoint
    non-strictly, move to the device, eval.  ``ckpt=None`` keeps the random initialisation (benchmarks, smoke)."""
    model_cfg = config["model"] if isinstance(config, dict) else config.model
    model = instantiate_from_config(model_cfg)
    if ckpt is not None:
        print(f"Loading model from {ckpt}")
        sd, step = _read_checkpoint(ckpt)
        if step is not None:
            print(f"Global Step: {step}")
        m, u = model.load_state_dict(sd, strict=False)
        if len(m) > 0 and verbose:
            print(f"missing keys:\n{m}")
        if len(u) > 0 and verbose:
            print(f"unexpected keys:\n{u}")
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError(f"Incorrect device name. Received: {device} (the HIP path runs on the GPU only)")
    model.to(device)
    model.eval()
    return model


def parse_args(argv=None):
    """Option names and defaults of txt2img.py:67-243."""
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, nargs="?", default="a professional photograph of an astronaut riding a triceratops")
    p.add_argument("--out_dir", "--outdir", type=str, nargs="?", default="outputs/txt2img-samples")
    p.add_argument("--steps", type=int, default=50, help="number of ddim sampling steps")
    p.add_argument("--plms", action="store_true", help="use plms sampling")
    p.add_argument("--dpm", action="store_true", help="use DPM (2) sampler")
    p.add_argument("--fixed_code", action="store_true", help="the same starting code across all samples")
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=3, help="sample this often")
    p.add_argument("--height", "--H", type=int, default=512)
    p.add_argument("--width", "--W", type=int, default=512)
    p.add_argument("--channel", "--C", type=int, default=4, help="latent channels")
    p.add_argument("--factor", "--f", type=int, default=8, help="downsampling factor")
    p.add_argument("--n_samples", type=int, default=3, help="batch size")
    p.add_argument("--n_rows", type=int, default=0, help="rows in the grid (default: n_samples)")
    p.add_argument("--scale", type=float, default=9.0, help="unconditional guidance scale")
    p.add_argument("--from_file", type=str, help="read prompts from this file, one per line")
    p.add_argument("--config", type=str, default=None, help="model YAML (default: the built-in v2-inference config)")
    p.add_argument("--ckpt", type=str, default=None, help="checkpoint (.ckpt/.pt/.safetensors); omit for random weights")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--precision", type=str, choices=["full", "autocast"], default="autocast")
    p.add_argument("--repeat", type=int, default=1, help="repeat each prompt of the file this often")
    p.add_argument("--device", type=str, choices=["cpu", "cuda"], default="cuda")
    p.add_argument("--torchscript", action="store_true")
    p.add_argument("--ipex", action="store_true")
    p.add_argument("--bf16", action="store_true", help="bf16 MFMA operands instead of the config's fp16")
    p.add_argument("--cond_file", type=str, default=None,
                   help=".npz/.pt with text embeddings `c` [N, 77, context_dim] (and optionally `uc`) replacing the text encoder")
    return p.parse_args(argv)


def chunk(it, size):
    it = iter(it)
    return iter(lambda: tuple(islice(it, size)), ())


def _synthetic_embedding(prompt, n_tokens, dim):
    g = torch.Generator().manual_seed(zlib.crc32(prompt.encode()) & 0x7FFFFFFF)
    return torch.randn(n_tokens, dim, generator=g)


def _conditioning(model, prompts, cond_table, offset, context_dim, device):
    """(c, uc) for a batch of prompts: cond stage if the config gave one, else --cond_file rows, else synthetic."""
    if model.cond_stage_model is not None and getattr(model.cond_stage_model, "encodes_text", True):
        return model.get_learned_conditioning(list(prompts)), model.get_learned_conditioning(len(prompts) * [""])
    if cond_table is not None and "tokens" in cond_table:
        # token ids instead of embeddings: run them through the text tower (needs its weights from the checkpoint)
        tok = cond_table["tokens"][offset:offset + len(prompts)].to(device)
        if tok.shape[0] != len(prompts):
            raise ValueError(f"--cond_file holds {cond_table['tokens'].shape[0]} token rows, need {offset + len(prompts)}")
        utok = cond_table["uc_tokens"][:1].expand(len(prompts), -1).contiguous().to(device)
        return model.get_learned_conditioning(tok), model.get_learned_conditioning(utok)
    if cond_table is not None:
        c = cond_table["c"][offset:offset + len(prompts)]
        if c.shape[0] != len(prompts):
            raise ValueError(f"--cond_file holds {cond_table['c'].shape[0]} embeddings, need {offset + len(prompts)}")
        uc = cond_table.get("uc")
        uc = uc[:1].expand(len(prompts), -1, -1) if uc is not None else torch.zeros_like(c)
        return c.to(device).float().contiguous(), uc.to(device).float().contiguous()
    c = torch.stack([_synthetic_embedding(p, 77, context_dim) for p in prompts])
    uc = _synthetic_embedding("", 77, context_dim)[None].expand(len(prompts), -1, -1)
    return c.to(device).contiguous(), uc.to(device).contiguous()


def _load_cond_file(path):
    if path is None:
        return None
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return {k: torch.from_numpy(z[k]) for k in z.files}
    return torch.load(path, map_location="cpu", weights_only=True)


def main(args):
    if args.torchscript or args.ipex:
        raise NotImplementedError("--torchscript/--ipex are the reference's Intel-CPU paths (txt2img.py:306-407); "
                                  "this build runs the HIP path on MI355X")
    if args.dpm:
        raise NotImplementedError("DPM-Solver is not part of the reference's samplers (ldm/models/diffusion: ddim, plms)")
    if args.device != "cuda":
        raise ValueError("the HIP path runs on the GPU only (--device cuda)")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    config = load_config(args.config)
    if args.bf16:
        up = config["model"]["params"]["unet_config"]["params"]
        up["use_fp16"], up["use_bf16"] = False, True
    device = torch.device("cuda")
    model = load_model_from_config(config, args.ckpt, device)
    sampler = PLMSSampler(model) if args.plms else DDIMSampler(model, device=device)

    os.makedirs(args.out_dir, exist_ok=True)
    sample_dir = os.path.join(args.out_dir, "samples")
    os.makedirs(sample_dir, exist_ok=True)
    batch_size = args.n_samples
    n_rows = args.n_rows if args.n_rows > 0 else batch_size
    if not args.from_file:
        data = [batch_size * [args.prompt]]
    else:
        print(f"reading prompts from {args.from_file}")
        with open(args.from_file, "r") as fh:
            lines = [p for p in fh.read().splitlines() for _ in range(args.repeat)]
        data = list(chunk(lines, batch_size))
    base_count = len(os.listdir(sample_dir))
    grid_count = len(os.listdir(args.out_dir)) - 1
    cond_table = _load_cond_file(args.cond_file)
    context_dim = config["model"]["params"]["unet_config"]["params"].get("context_dim", 1024)
    if cond_table is None and not getattr(model.cond_stage_model, "encodes_text", model.cond_stage_model is not None):
        print("[txt2img] no text encoder that takes strings and no --cond_file: using synthetic embeddings seeded "
              "from the prompt strings")
    shape = [args.channel, args.height // args.factor, args.width // args.factor]
    start_code = torch.randn([batch_size] + shape, device=device) if args.fixed_code else None

    from PIL import Image
    all_u8 = []
    for _ in range(args.n_iter):
        offset = 0      # --cond_file holds one row per prompt of `data`; every n_iter pass walks it from the start
        for prompts in data:
            prompts = list(prompts)
            c, uc = _conditioning(model, prompts, cond_table, offset, context_dim, device)
            offset += len(prompts) if cond_table is not None and args.from_file else 0
            x_T = start_code[:len(prompts)] if start_code is not None else None
            samples, _ = sampler.sample(S=args.steps, conditioning=c, batch_size=len(prompts), shape=shape, verbose=False,
                                        unconditional_guidance_scale=args.scale,
                                        unconditional_conditioning=uc if args.scale != 1.0 else None,
                                        eta=args.ddim_eta, x_T=x_T)
            images = model.decode_first_stage(samples)                 # fp32 NCHW in [-1, 1]
            u8 = ops.image_to_u8(images.float().contiguous())          # clamp((x+1)/2)*255 -> uint8 NHWC
            for img in u8.cpu().numpy():
                Image.fromarray(img).save(os.path.join(sample_dir, f"{base_count:05}.png"))
                base_count += 1
            all_u8.append(u8.cpu())
    grid = torch.cat(all_u8)                                           # [N, H, W, 3]
    n = grid.shape[0]
    cols = n_rows
    rows = (n + cols - 1) // cols
    canvas = np.zeros((rows * args.height, cols * args.width, 3), dtype=np.uint8)
    for i in range(n):
        r, cidx = divmod(i, cols)
        canvas[r * args.height:(r + 1) * args.height, cidx * args.width:(cidx + 1) * args.width] = grid[i].numpy()
    Image.fromarray(canvas).save(os.path.join(args.out_dir, f"grid-{grid_count:04}.png"))
    print(f"Your samples are ready and waiting for you here: \n{args.out_dir} \n \nEnjoy.")
    return grid


if __name__ == "__main__":
    main(parse_args())
