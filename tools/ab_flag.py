#!/usr/bin/env python3
"""Same-process A/B of a boolean switch of stablediffusion_amd.ops (FOLD_LN: the LayerNorm fold, read when the weights are
packed; FUSE_SPLIT: split-K finished inside the kernel, read per launch; ...) on the UNet step: two models with the same
weights, one built and run with the switch on and one with it off, timed in alternation (best of R rounds each) - box-to-box
and first-loop effects cancel.
usage: ab_flag.py FLAG [B=8] [iters=10] [rounds=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stablediffusion_amd import ops

FLAG = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
assert isinstance(getattr(ops, FLAG), bool)
dev = torch.device("cuda:0")
z = torch.randn(B, 4, 64, 64, device=dev)
x = torch.cat([z, z])
c = torch.randn(2 * B, 77, 1024, device=dev)
ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
models = {}
for name, flag in (("on", True), ("off", False)):
    setattr(ops, FLAG, flag)
    m = bench.build_model(dev, "eps", "fp16")
    for _ in range(2):
        m.apply_model(x, ts, c, cfg_pairs=True)      # packs the weights under this flag
    models[name] = m
best = {k: 1e9 for k in models}
for r in range(rounds):
    for name, m in models.items():
        setattr(ops, FLAG, name == "on")
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
print(f"{FLAG} batch {2 * B}: on {best['on']:.3f} ms   off {best['off']:.3f} ms   diff {best['on'] - best['off']:+.3f}", flush=True)
