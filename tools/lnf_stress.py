#!/usr/bin/env python3
"""Stress of the LayerNorm-folded projections (development aid): every (tile, shape) launched many times, each output compared
with the first launch of the 128 x 160 tile - the check that found the lost products of DESIGN.md section 3.
usage: lnf_stress.py [repeats=60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
one = lambda m_, n_, k_, t=128: 1
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
total_bad = 0
for (m, k, n) in ((2100, 640, 1920), (4000, 1280, 1280), (8192, 320, 320), (8192, 320, 960), (16384, 640, 640), (512, 1280, 1280),
                  (2048, 640, 640), (65536, 320, 320), (4096, 1280, 3840)):
    x = (rnd(m, k) * (0.5 + 2.0 * torch.rand(m, 1, device=dev)) + 1.5 * rnd(m, 1)).to(F16)
    gamma, beta = 1.0 + 0.2 * rnd(k), 0.3 * rnd(k)
    ln = (gamma, beta, 1e-5)
    pw = ops.pack_linear(rnd(n, k, scale=0.06), rnd(n), dtype=F16, ln=ln)
    st = ops.row_stats_of(x)
    ops._auto_split_k = one
    ops.FORCE_TILE_N, ops.FORCE_TILE_M = 0, 128
    want = ops.linear(x, pw, ln=ln, stats=st).float()
    for name, tn, tm in (("128x160", 0, 128), ("256x160", 0, 256), ("128x64", 64, 128), ("64x64", 64, 64), ("planned", 0, 0)):
        ops.FORCE_TILE_N, ops.FORCE_TILE_M = tn, tm
        ops._auto_split_k = one if name != "planned" else ops._AUTO_SPLIT_K
        bad = 0
        try:
            for _ in range(reps):
                got = ops.linear(x, pw, ln=ln, stats=st).float()
                bad += int(((got - want).abs() > 2e-2).sum())
        except ops.hip.SdmiError as e:
            print(f"{(m, k, n)} {name}: n/a", flush=True)
            continue
        total_bad += bad
        print(f"{(m, k, n)} {name}: {bad} outputs off in {reps} launches", flush=True)
    ops._auto_split_k = ops._AUTO_SPLIT_K
    ops.FORCE_TILE_N = ops.FORCE_TILE_M = 0
print("total", total_bad)
sys.exit(1 if total_bad else 0)
