"""Host-side schedule tables (tiny, numpy): the product's own statement of the reference's
ldm/modules/diffusion/util.py:11-141 helpers, with the same names and argument meaning."""
import math

import numpy as np


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "cosine":
        t = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        a = np.cos(t / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], a_min=0, a_max=0.999)
    if schedule == "squared_cos_cap_v2":
        bar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - bar((i + 1) / n_timestep) / bar(i / n_timestep), 0.999) for i in range(n_timestep)])
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discrete_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discrete_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discrete_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discrete_method}"')
    steps_out = steps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alpha_cumprod, ddim_timesteps, eta, verbose=True):
    acp = np.asarray(alpha_cumprod)
    alphas = acp[ddim_timesteps]
    alphas_prev = np.asarray([acp[0]] + acp[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule: {sigmas}")
    return sigmas, alphas, alphas_prev
