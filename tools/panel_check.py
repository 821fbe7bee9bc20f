#!/usr/bin/env python3
"""Development aid: the A-panel-stationary kernels (K = 320; sdmi_gemm algo 2 / a fused LayerNorm) on their UNet shapes - time
as a hipGraph replay and a dump of the outputs, so that two builds (SDMI_LIB=<path>) can be compared bit for bit.
usage: panel_check.py <out.pt>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
from small_tile_check import timed

dev = torch.device("cuda:0")
outs = {}
for dt in (torch.float16, torch.bfloat16):
    for kind, m, n in (("geglu", 65536, 1280), ("geglu", 16384, 640), ("lin", 65536, 960), ("lin", 65536, 320), ("lin", 32768, 320),
                       ("lin", 16400, 960), ("geglu", 16500, 256)):
        for with_ln in (True, False):
            torch.manual_seed(m + n)
            x = (torch.randn(m, 320, device=dev) * 1.3 + 0.2).to(dt)
            g, b = torch.randn(320, device=dev), torch.randn(320, device=dev)
            if kind == "geglu":
                pw = ops.pack_geglu(torch.randn(2 * n, 320, device=dev) * 0.05, torch.randn(2 * n, device=dev), dtype=dt)
                res = None
            else:
                pw = ops.pack_linear(torch.randn(n, 320, device=dev) * 0.05, torch.randn(n, device=dev), dtype=dt)
                res = torch.randn(m, n, device=dev).to(dt) if n == 320 else None
            ops.FORCE_ALGO = 0 if with_ln else 2
            fn = lambda: ops.linear(x, pw, res=res, ln=(g, b, 1e-5) if with_ln else None)
            y = fn()
            t = timed(fn)
            key = f"{str(dt)[6:]}_{kind}_{m}_{n}_ln{int(with_ln)}"
            outs[key] = y.cpu()
            fl = 2 * m * n * 320 * (2 if kind == "geglu" else 1)
            print(f"{key:34s} {t:7.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  sum {y.float().sum().item():.6e} nan {int(torch.isnan(y.float()).any())}", flush=True)
ops.FORCE_ALGO = 0
torch.save(outs, sys.argv[1])
