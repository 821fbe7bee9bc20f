"""Oracle: host-side schedule tables (test infrastructure, see oracle/__init__.py).

Restates, in numpy/float64 exactly as the reference does:
  * make_beta_schedule            reference ldm/modules/diffusion/util.py:104-141
  * make_ddim_timesteps           reference ldm/modules/diffusion/util.py:11-61
  * make_ddim_sampling_parameters reference ldm/modules/diffusion/util.py:64-83
  * DDPM.register_schedule tables reference ldm/models/diffusion/ddpm.py:285-389
  * DDIMSampler.make_schedule     reference ldm/models/diffusion/ddim.py:45-103
"""
import math

import numpy as np


def beta_schedule(kind, n, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    # util.py:114-115 -- "linear" is linear in sqrt(beta), float64.
    if kind == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=np.float64) ** 2
    if kind == "cosine":  # util.py:117-131
        ts = np.arange(n + 1, dtype=np.float64) / n + cosine_s
        a = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    if kind == "squared_cos_cap_v2":  # util.py:86-101,132-136
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)])
    if kind == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n, dtype=np.float64)
    if kind == "sqrt":
        return np.linspace(linear_start, linear_end, n, dtype=np.float64) ** 0.5
    raise ValueError(f"schedule '{kind}' unknown.")


def ddpm_tables(kind="linear", n=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """betas / alphas_cumprod / alphas_cumprod_prev as float32 (ddpm.py:299-321)."""
    betas = beta_schedule(kind, n, linear_start, linear_end, cosine_s)
    acp = np.cumprod(1.0 - betas, axis=0)
    acp_prev = np.append(1.0, acp[:-1])
    return {
        "betas": betas.astype(np.float32),
        "alphas_cumprod": acp.astype(np.float32),
        "alphas_cumprod_prev": acp_prev.astype(np.float32),
    }


def ddim_timesteps(method, n_ddim, n_ddpm):
    if method == "uniform":  # util.py:24-29
        ts = np.asarray(list(range(0, n_ddpm, n_ddpm // n_ddim)))
    elif method == "quad":  # util.py:31-33
        ts = (np.linspace(0, np.sqrt(n_ddpm * 0.8), n_ddim) ** 2).astype(int)
    else:
        raise NotImplementedError(method)
    return ts + 1  # util.py:55


def ddim_tables(alphas_cumprod, n_ddim, eta=0.0, method="uniform"):
    """What DDIMSampler.make_schedule leaves on the sampler (ddim.py:55-103).

    alphas_cumprod: float32 array of length T (the model's buffer)."""
    acp = np.asarray(alphas_cumprod)
    ts = ddim_timesteps(method, n_ddim, acp.shape[0])
    a = acp[ts]                                             # util.py:73
    a_prev = np.asarray([acp[0]] + acp[ts[:-1]].tolist())   # util.py:74
    sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))  # util.py:77
    return {
        "ddim_timesteps": ts,
        "ddim_alphas": a,
        "ddim_alphas_prev": a_prev,
        "ddim_sigmas": sig,
        "ddim_sqrt_one_minus_alphas": np.sqrt(1.0 - a),     # ddim.py:95
    }
