#!/usr/bin/env python3
"""Run one GEMM/conv shape a few times (profiling target).  usage: one_gemm.py conv|lin a b c d e [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
dev = torch.device("cuda:0")
kind = sys.argv[1]
a = [int(x) for x in sys.argv[2:7] if x != "-"]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
BF16 = torch.bfloat16
if kind == "conv":
    b, h, w, cin, cout = a
    x = torch.randn(b, h, w, cin, device=dev).to(BF16)
    pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, torch.randn(cout, device=dev))
    fn = lambda: ops.conv3x3(x, pw)
else:
    m, n, k = a[:3]
    x = torch.randn(m, k, device=dev).to(BF16)
    pw = ops.pack_linear(torch.randn(n, k, device=dev) * 0.02, torch.randn(n, device=dev))
    fn = lambda: ops.linear(x, pw)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
