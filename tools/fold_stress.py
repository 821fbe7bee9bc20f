#!/usr/bin/env python3
"""Accuracy of the folded LayerNorm (pack_*(ln=...)) against LayerNorm -> f16 -> GEMM as the rows' mean / std ratio grows and with
a "massive activation" channel (development aid; the numbers are in profiles/r03_lnfold_precision.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from stablediffusion_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
g = torch.Generator(device="cpu").manual_seed(3)
m, k, n = 2048, 640, 1920
w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
gamma, beta = (1 + 0.2 * torch.randn(k, generator=g)).to(dev), (0.3 * torch.randn(k, generator=g)).to(dev)
ln = (gamma, beta, 1e-5)
folded, plain = ops.pack_linear(w, None, dtype=F16, ln=ln), ops.pack_linear(w, None, dtype=F16)
for ratio in (0, 1, 3, 10, 30, 100, 300):
    for spike in (False, True):
        x = torch.randn(m, k, generator=g)
        if spike:
            x[:, 7] *= 60.0          # a "massive activation" channel
        x = (x + ratio * x.std()).to(dev).to(F16)
        ref = F.linear(F.layer_norm(x.float(), (k,), gamma, beta, 1e-5), w)
        e_f = ((ops.linear(x, folded, ln=ln).float() - ref).abs().max() / ref.abs().max()).item()
        e_u = ((ops.linear(x, plain, ln=ln).float() - ref).abs().max() / ref.abs().max()).item()
        print(f"mean/std {ratio:4d} spike {spike}: folded {e_f:.2e}   LayerNorm->f16->GEMM {e_u:.2e}", flush=True)
