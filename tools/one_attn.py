#!/usr/bin/env python3
"""Run one attention shape a few times (profiling target).  usage: one_attn.py b heads nq nk [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
dev = torch.device("cuda:0")
b, heads, nq, nk = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
c = heads * 64
q = torch.randn(b * nq, c, device=dev).to(torch.bfloat16)
kv = torch.randn(b * nk, 2 * c, device=dev).to(torch.bfloat16)
for _ in range(iters):
    ops.attention_d64(q, kv[:, :c], kv[:, c:], heads, b, nq, nk, 0.125)
torch.cuda.synchronize()
