#!/usr/bin/env python3
"""Cycles per phase of one workgroup of sdmi_xattn_block (s_memtime stamps of wave 0 of the first / last workgroup).
Needs a library built with the stamps:  tools/ab_build.sh -DSDMI_XATTN_TIMING   and   SDMI_LIB=.../libsdmi_hip_b.so
(that build writes the stamps over the first bytes of `out`: a measurement build, not a product one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
c, heads, n, nk = 320, 5, 4096, 77
for nb in (2, 8, 16):
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = r(nb * n, c).to(F16); gamma, beta = 1 + 0.1 * r(c), 0.1 * r(c)
    wq, wo, bo = r(c, c, sc=0.06), r(c, c, sc=0.06), r(c)
    kv = r(nb * nk, 2 * c).to(F16); k, v = kv[:, :c], kv[:, c:]
    pq, po = ops.pack_linear(wq, dtype=F16), ops.pack_linear(wo, bo, dtype=F16)
    for _ in range(3):
        out = ops.xattn_block(x, (gamma, beta, 1e-5), pq, po, k, v, nk, heads, 0.125, n)
    torch.cuda.synchronize()
    t = out.view(torch.int64).reshape(-1)[:16].cpu().tolist()
    for b in (0, 1):
        ts = t[8 * b: 8 * b + 7]
        print(nb, "block", "first" if b == 0 else "last", "cycles: panel+LN %d gemm1 %d qwrite %d attn %d gemm2 %d epi %d total %d" % (
            ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[6] - ts[0]), flush=True)
