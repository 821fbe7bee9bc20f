#!/usr/bin/env python3
"""UNet step time of the benchmark batch (development aid for same-box A/B runs with SDMI_LIB=<library>).
usage: step_time.py [B=8] [iters=10]   -> ms per apply_model call, without and with the samplers' cfg_pairs hint
env NO_WIDE=1: the GEMM plan without the 256 x 256 x 32 tile; GRAPH=1: hipGraph replay; DEC=1: also the decode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
from stablediffusion_amd import ops
if os.environ.get("NO_WIDE"):           # A/B: never the 256 x 256 x 32 tile
    ops.WIDE_TILE = False
model = bench.build_model(dev, "eps", "fp16")
z = torch.randn(B, 4, 64, 64, device=dev)
x = torch.cat([z, z])
c = torch.randn(2 * B, 77, 1024, device=dev)
ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
apply = model.apply_model
if os.environ.get("GRAPH"):          # replay the step from a captured hipGraph (graph.py)
    from stablediffusion_amd.graph import GraphedApplyModel
    apply = GraphedApplyModel(model)
for kw in ({}, {"cfg_pairs": True}):
    for _ in range(2):
        apply(x, ts, c, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        apply(x, ts, c, **kw)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('SDMI_LIB', 'default')} batch {2 * B} {kw}: {e0.elapsed_time(e1) / iters:.3f} ms", flush=True)
if os.environ.get("DEC"):            # the first-stage decode of the same batch
    for _ in range(2):
        model.decode_first_stage(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        model.decode_first_stage(z)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('SDMI_LIB', 'default')} decode batch {B}: {e0.elapsed_time(e1) / iters:.3f} ms", flush=True)
