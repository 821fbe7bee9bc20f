#!/usr/bin/env python3
"""Same-process A/B of the LayerNorm fold (ops.FOLD_LN) on the UNet step: two models with the same weights, one packed with
the fold and one without, timed in alternation (best of R rounds each) - box-to-box and first-loop effects cancel.
usage: ab_fold.py [B=8] [iters=10] [rounds=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stablediffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
z = torch.randn(B, 4, 64, 64, device=dev)
x = torch.cat([z, z])
c = torch.randn(2 * B, 77, 1024, device=dev)
ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
models = {}
for name, flag in (("fold", True), ("nofold", False)):
    ops.FOLD_LN = flag
    m = bench.build_model(dev, "eps", "fp16")
    for _ in range(2):
        m.apply_model(x, ts, c, cfg_pairs=True)      # packs the weights under this flag
    models[name] = m
best = {k: 1e9 for k in models}
for r in range(rounds):
    for name, m in models.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m.apply_model(x, ts, c, cfg_pairs=True)
        e0.record()
        for _ in range(iters):
            m.apply_model(x, ts, c, cfg_pairs=True)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        best[name] = min(best[name], t)
        print(f"round {r} {name}: {t:.3f} ms", flush=True)
print(f"batch {2 * B}: fold {best['fold']:.3f} ms   nofold {best['nofold']:.3f} ms   diff {best['fold'] - best['nofold']:+.3f}", flush=True)
