#!/usr/bin/env python3
"""Micro-benchmark of sdmi_attention_d64 on the SD2 UNet shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LN2 = 0.6931471805599453
for scale, (b, heads, nq, nk) in [(sc, shp) for sc in (0.125,) for shp in [(16, 5, 4096, 4096), (16, 10, 1024, 1024), (16, 20, 256, 256), (16, 5, 4096, 77), (16, 10, 1024, 77), (16, 20, 256, 77), (2, 5, 9216, 9216)]]:
    c = heads * 64
    qkv = torch.randn(b * nq, 3 * c, device=dev)
    if scale == LN2:
        qkv[:, :c] *= 0.125 * 1.4426950408889634       # the fold the product path applies in the q projection
    qkv = qkv.to(torch.bfloat16)
    kv = torch.randn(b * nk, 2 * c, device=dev).to(torch.bfloat16)
    if nq == nk:
        fn = lambda: ops.attention_d64(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], heads, b, nq, nk, scale)
    else:
        fn = lambda: ops.attention_d64(qkv[:, :c], kv[:, :c], kv[:, c:], heads, b, nq, nk, scale)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 4 * b * heads * nq * nk * 64
    print(f"attn scale={scale:.3f} b={b} heads={heads} nq={nq} nk={nk}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
