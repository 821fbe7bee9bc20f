#!/usr/bin/env python3
"""Development aid: GroupNorm(32)+SiLU on the UNet's tensor shapes, each timed as a hipGraph replay (eager timing is
bound by the host), plus a checksum of the output so that two builds (SDMI_LIB=<path>) can be compared.
usage: python tools/bench_gn.py [nb=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
from small_tile_check import timed

dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [(64, 1280), (64, 2560), (256, 640), (256, 1280), (256, 1920), (256, 2560), (1024, 320), (1024, 640), (1024, 960),
          (1024, 1280), (1024, 1920), (4096, 320), (4096, 640), (4096, 960)]
tot = 0.0
for hw, c in SHAPES:
    torch.manual_seed(hw + c)
    x = (torch.randn(nb, hw, c, device=dev) * 1.5 + 0.3).to(torch.float16)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    y = ops.groupnorm(x, g, b, 1e-5, True)
    xf = x.float().view(nb, hw, 32, c // 32)
    mu, var = xf.mean((1, 3), keepdim=True), xf.var((1, 3), unbiased=False, keepdim=True)
    ref = torch.nn.functional.silu(((xf - mu) * torch.rsqrt(var + 1e-5)).view(nb, hw, c) * g + b)
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    t = timed(lambda: ops.groupnorm(x, g, b, 1e-5, True))
    tot += t
    print(f"nb {nb:3d} hw {hw:5d} C {c:5d}  {t:7.2f} us  {2 * x.numel() * 2 / t / 1e6:6.2f} TB/s  rel_err {err:.2e}  sum {y.float().sum().item():.6e}",
          flush=True)
print(f"total {tot:.1f} us", flush=True)
