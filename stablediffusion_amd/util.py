"""Config factory: the drop-in boundary of the reference (ldm/utils.py:27-107).

``instantiate_from_config({"target": "pkg.mod.Class", "params": {...}})`` -> ``Class(**params)``.
Same error behaviour as the reference: ``KeyError("Expected key `target` to instantiate.")`` when
``target`` is missing, ``None`` for the two sentinel strings.
"""
import importlib
from inspect import isfunction

# YAML written for the reference or for upstream names classes by module paths that do not exist
# here; map them onto this package so such a config runs unchanged on the HIP path.
TARGET_ALIASES = {
    "ldm.models.diffusion.ddpm.LatentDiffusion": "stablediffusion_amd.ddpm.LatentDiffusion",
    "ldm.modules.diffusionmodules.openaimodel.UNetModel": "stablediffusion_amd.unet.UNetModel",
    "ldm.modules.diffusion.openai_model.UNetModel": "stablediffusion_amd.unet.UNetModel",
    "ldm.models.autoencoder.AutoencoderKL": "stablediffusion_amd.vae.AutoencoderKL",
    "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder": "stablediffusion_amd.encoders.FrozenOpenCLIPEmbedder",
}


def default(val, d):
    if val is not None:
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string, reload=False):
    string = TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        if config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total
