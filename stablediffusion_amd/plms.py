"""PLMSSampler on the HIP path.  The reference only parses ``--plms``
(scripts/txt2img.py:96-100) and has no plms.py; this is upstream's pseudo linear multistep sampler
restated: 4th-order Adams-Bashforth on eps with a 2-evaluation first step, eta = 0,
eps-parameterisation, around the same ``apply_model`` call as DDIM.  The guidance combine and the
multistep eps combination run in sdmi_lincomb4, the x_prev update reuses sdmi_ddim_step.
"""
import numpy as np
import torch

from . import ops
from .ddim import DDIMSampler


class PLMSSampler(DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        super().make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose)

    @torch.no_grad()
    def ddim_sampling(self, condition, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      temperature=1., unconditional_guidance_scale=1., unconditional_conditioning=None, **kw):
        if getattr(self.model, "parameterization", "eps") != "eps":
            raise NotImplementedError("PLMS is defined for eps-parameterisation only (as upstream)")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        steps = self.ddim_timesteps
        time_range = np.flip(steps)
        total = steps.shape[0]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        cfg = unconditional_conditioning is not None and unconditional_guidance_scale != 1.
        c_in = torch.cat([unconditional_conditioning, condition]).contiguous() if cfg else condition
        scale = float(unconditional_guidance_scale)

        def eps_at(x, step):
            n_in = 2 * b if cfg else b
            ts = torch.full((n_in,), int(step), device=device, dtype=torch.long)
            kw = self._time_emb_kw(step)
            if cfg and getattr(self, "cfg_pairs", False):
                kw["cfg_pairs"] = True
            out = self._apply(torch.cat([x, x]) if cfg else x, ts, c_in, **kw).float()
            if cfg:
                e_uc, e_c = out[:b].contiguous(), out[b:].contiguous()
                return ops.lincomb([(1.0 - scale, e_uc), (scale, e_c)])
            # a fresh tensor: with hip_graph=True `out` is the graph's static output buffer, which the next replay
            # overwrites while this eps is still needed (step 0's second evaluation, the multistep history)
            return out.clone(memory_format=torch.contiguous_format)

        def x_prev_of(x, e, index):
            return ops.ddim_step(x, e.contiguous(), float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index]),
                                 0.0, float(self.ddim_sqrt_one_minus_alphas[index]), want_pred_x0=True)

        old_eps = []
        for i, step in enumerate(time_range):
            index = total - i - 1
            step_next = time_range[min(i + 1, total - 1)]
            e_t = eps_at(img, step)
            if len(old_eps) == 0:
                x_p, _ = x_prev_of(img, e_t, index)
                e_prime = ops.lincomb([(0.5, e_t), (0.5, eps_at(x_p, step_next))])
            elif len(old_eps) == 1:
                e_prime = ops.lincomb([(1.5, e_t), (-0.5, old_eps[-1])])
            elif len(old_eps) == 2:
                e_prime = ops.lincomb([(23 / 12, e_t), (-16 / 12, old_eps[-1]), (5 / 12, old_eps[-2])])
            else:
                e_prime = ops.lincomb([(55 / 24, e_t), (-59 / 24, old_eps[-1]), (37 / 24, old_eps[-2]), (-9 / 24, old_eps[-3])])
            img, pred_x0 = x_prev_of(img, e_prime, index)
            old_eps = (old_eps + [e_t])[-3:]
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates
