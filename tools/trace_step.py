#!/usr/bin/env python3
"""One UNet step population for `rocprofv3 --kernel-trace` (development aid): usage trace_step.py [B=1] [iters=3] [latent=64].
Runs apply_model at UNet batch 2B with the samplers' cfg_pairs hint; tools/summarize_trace.py aggregates the
dispatch CSV by (kernel, grid) so that the launches of one shape stand alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
model = bench.build_model(dev, "eps", "fp16")
z = torch.randn(B, 4, h, h, device=dev)
x = torch.cat([z, z])
c = torch.randn(2 * B, 77, 1024, device=dev)
ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
for _ in range(2):
    model.apply_model(x, ts, c, cfg_pairs=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    model.apply_model(x, ts, c, cfg_pairs=True)
e1.record()
torch.cuda.synchronize()
print(f"batch {2 * B} latent {h}: {e0.elapsed_time(e1) / iters:.3f} ms per step (under the profiler when traced)", flush=True)
if os.environ.get("DEC"):
    model.decode_first_stage(z)
    torch.cuda.synchronize()
