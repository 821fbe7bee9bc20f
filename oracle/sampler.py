"""Oracle: DDIM / PLMS sampling loops and the LatentDiffusion glue (fp32, CPU).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference stops before the loop body
(ldm/models/diffusion/ddim.py:184-186 and :242-249), has no ``p_sample_ddim``, no
``apply_model`` / ``decode_first_stage`` (ddpm.py:658-669 is cut off) and no PLMS sampler.
These are therefore restated from the upstream (Stability-AI/stablediffusion) semantics
the reference's stubs and SURVEY.md §3.2/§3.4 describe: "parity unpinned" by the
reference, pinned only by the schedule tables it does compute.
"""
import numpy as np
import torch

from . import model as M
from . import schedule as S


def apply_model(unet_sd, unet_cfg, x, t, cond):
    """LatentDiffusion.apply_model -> DiffusionWrapper.forward 'cross_attn' branch
    (ddpm.py:442-451): context = cat(c_crossattn, dim=1); unet(x, t, context=ctx)."""
    if isinstance(cond, dict):
        cond = torch.cat(cond["c_crossattn"], dim=1)
    elif isinstance(cond, (list, tuple)):
        cond = torch.cat(list(cond), dim=1)
    return M.unet_forward(unet_sd, unet_cfg, x, t, cond)


def decode_first_stage(vae_sd, ddcfg, z, scale_factor=0.18215, vae_attn="reference"):
    """z / scale_factor -> AutoencoderKL.decode (yaml:17)."""
    return M.vae_decode(vae_sd, ddcfg, z / scale_factor, vae_attn)


def ddim_step(x, e_uc, e_c, scale, a_t, a_prev, sigma_t, sqrt_1m_a_t, parameterization="eps",
              noise=None, temperature=1.0):
    """One p_sample_ddim update given the two model outputs."""
    out = e_uc + scale * (e_c - e_uc) if e_uc is not None else e_c
    sqrt_a = a_t ** 0.5
    if parameterization == "v":
        e_t = sqrt_a * out + sqrt_1m_a_t * x
        pred_x0 = sqrt_a * x - sqrt_1m_a_t * out
    else:
        e_t = out
        pred_x0 = (x - sqrt_1m_a_t * e_t) / sqrt_a
    dir_xt = (1.0 - a_prev - sigma_t ** 2) ** 0.5 * e_t
    n = 0.0 if noise is None else sigma_t * noise * temperature
    return (a_prev ** 0.5) * pred_x0 + dir_xt + n, pred_x0, e_t


def ddim_sample(model_fn, alphas_cumprod, n_steps, x_T, c, uc=None, scale=1.0, eta=0.0,
                parameterization="eps", record=None):
    """Full DDIM loop.  ``model_fn(x, t, cond) -> eps/v``.  CFG doubles the batch
    (x_in = cat([x, x]); c_in = cat([uc, c])) exactly as upstream's p_sample_ddim."""
    tab = S.ddim_tables(alphas_cumprod, n_steps, eta)
    ts = tab["ddim_timesteps"]
    total = ts.shape[0]
    x = x_T
    b = x.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if uc is None or scale == 1.0:
            e_uc, e_c = None, model_fn(x, t, c)
        else:
            out = model_fn(torch.cat([x, x]), torch.cat([t, t]), torch.cat([uc, c]))
            e_uc, e_c = out.chunk(2)
        noise = torch.randn_like(x) if eta > 0 else None
        x_prev, pred_x0, e_t = ddim_step(
            x, e_uc, e_c, scale, float(tab["ddim_alphas"][index]), float(tab["ddim_alphas_prev"][index]),
            float(tab["ddim_sigmas"][index]), float(tab["ddim_sqrt_one_minus_alphas"][index]),
            parameterization, noise)
        if record is not None:
            record(i, int(step), x, e_uc, e_c, x_prev)
        x = x_prev
    return x


def plms_sample(model_fn, alphas_cumprod, n_steps, x_T, c, uc=None, scale=1.0, record=None):
    """PLMS (pseudo linear multistep, 4th-order Adams-Bashforth with a 2-evaluation first
    step), eta = 0, eps-parameterisation only - upstream's PLMSSampler semantics
    (txt2img.py:96-100 only parses the flag)."""
    tab = S.ddim_tables(alphas_cumprod, n_steps, 0.0)
    ts = tab["ddim_timesteps"]
    total = ts.shape[0]
    time_range = np.flip(ts)
    x = x_T
    b = x.shape[0]

    def eps_at(xx, step):
        t = torch.full((b,), int(step), dtype=torch.long)
        if uc is None or scale == 1.0:
            return model_fn(xx, t, c)
        e_uc, e_c = model_fn(torch.cat([xx, xx]), torch.cat([t, t]), torch.cat([uc, c])).chunk(2)
        return e_uc + scale * (e_c - e_uc)

    def x_prev_of(xx, e, index):
        a_t, a_prev = float(tab["ddim_alphas"][index]), float(tab["ddim_alphas_prev"][index])
        s1m = float(tab["ddim_sqrt_one_minus_alphas"][index])
        pred_x0 = (xx - s1m * e) / a_t ** 0.5
        return a_prev ** 0.5 * pred_x0 + (1.0 - a_prev) ** 0.5 * e

    old = []
    for i, step in enumerate(time_range):
        index = total - i - 1
        step_next = time_range[min(i + 1, total - 1)]
        e_t = eps_at(x, step)
        if len(old) == 0:
            x_p = x_prev_of(x, e_t, index)
            e_next = eps_at(x_p, step_next)
            e_prime = (e_t + e_next) / 2
        elif len(old) == 1:
            e_prime = (3 * e_t - old[-1]) / 2
        elif len(old) == 2:
            e_prime = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        x_prev = x_prev_of(x, e_prime, index)
        if record is not None:
            record(i, int(step), x, e_t, x_prev)
        old = (old + [e_t])[-3:]
        x = x_prev
    return x
