#!/usr/bin/env python3
"""Micro-benchmark of GroupNorm(+SiLU) and LayerNorm on the SD2 UNet / decoder shapes (development aid).
Prints us and the achieved algorithmic bandwidth (one read + one write of the tensor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for b, hw, c1, c2 in [(16, 4096, 320, 0), (16, 4096, 320, 320), (16, 4096, 640, 320), (16, 1024, 640, 0), (16, 1024, 1280, 640),
                      (16, 256, 1280, 0), (16, 64, 1280, 1280), (8, 262144, 128, 0), (8, 65536, 256, 0), (8, 16384, 512, 0)]:
    s = int(hw ** 0.5)
    x1 = torch.randn(b, s, s, c1, device=dev).to(torch.bfloat16)
    x2 = torch.randn(b, s, s, c2, device=dev).to(torch.bfloat16) if c2 else None
    g, be = torch.ones(c1 + c2, device=dev), torch.zeros(c1 + c2, device=dev)
    us = timeit(lambda: ops.groupnorm(x1, g, be, 1e-5, True, x2=x2))
    nbytes = 2 * b * hw * (c1 + c2) * 2
    print(f"groupnorm b={b} hw={hw} C={c1}+{c2}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
for rows, c in [(65536, 320), (16384, 640), (4096, 1280)]:
    x = torch.randn(rows, c, device=dev).to(torch.bfloat16)
    g, be = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    us = timeit(lambda: ops.layernorm(x, g, be, 1e-5))
    print(f"layernorm rows={rows} C={c}: {us:8.1f} us  {2 * rows * c * 2 / us / 1e6:6.2f} TB/s", flush=True)
