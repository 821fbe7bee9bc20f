"""stablediffusion_amd - MI355X-native (gfx950) latent-diffusion denoising hot path.

Drop-in for the reference's UNet / VAE-decoder / DDIM path behind its own factory surface:
``instantiate_from_config`` (YAML ``target:`` strings), ``LatentDiffusion.apply_model`` /
``decode_first_stage`` and ``DDIMSampler.sample``.  Arithmetic runs in hand-written HIP kernels
(``csrc/``, C ABI in ``include/sdmi.h``); PyTorch only provides device memory, streams and
``torch.distributed``.
"""
from .util import instantiate_from_config, get_obj_from_str  # noqa: F401

__all__ = ["instantiate_from_config", "get_obj_from_str"]
