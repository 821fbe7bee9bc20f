#!/usr/bin/env python3
"""sdmi_xattn_block against the three launches it replaces (LayerNorm-folded to_q, attention over the prompt's keys,
to_out + residual) on the level-0 token stream.  usage: bench_xattn.py [images ...]   (4096 tokens per image, 77 keys)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16
c, heads, n, nk = 320, 5, 4096, 77


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for nb in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32]:
    g = torch.Generator(device="cpu").manual_seed(nb)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = r(nb * n, c).to(F16)
    gamma, beta = 1 + 0.1 * r(c), 0.1 * r(c)
    wq, wo, bo = r(c, c, sc=0.06), r(c, c, sc=0.06), r(c)
    kv = r(nb * nk, 2 * c).to(F16)
    k, v = kv[:, :c], kv[:, c:]
    ln = (gamma, beta, 1e-5)
    pq, po = ops.pack_linear(wq, dtype=F16), ops.pack_linear(wo, bo, dtype=F16)
    pqf = ops.pack_linear(wq, dtype=F16, ln=ln)
    st, slots = ops.row_stats_of(x)
    scale = 0.125

    def chain():
        q = ops.linear(x, pqf, ln=ln, stats=(st, slots))
        a = ops.attention_d64(q, k, v, heads, nb, n, nk, scale)
        return ops.linear(a, po, res=x, row_stats=True)

    def parts():
        q = ops.linear(x, pqf, ln=ln, stats=(st, slots))
        a = ops.attention_d64(q, k, v, heads, nb, n, nk, scale)
        return (timed(lambda: ops.linear(x, pqf, ln=ln, stats=(st, slots))),
                timed(lambda: ops.attention_d64(q, k, v, heads, nb, n, nk, scale)),
                timed(lambda: ops.linear(a, po, res=x, row_stats=True)))

    fused = lambda: ops.xattn_block(x, ln, pq, po, k, v, nk, heads, scale, n, row_stats=True)
    tc, tf, tp = timed(chain), timed(fused), parts()
    print(f"images {nb:3d} rows {nb * n:6d}: chain {tc:7.1f} us (q {tp[0]:.1f} attn {tp[1]:.1f} o {tp[2]:.1f})   one launch {tf:7.1f} us", flush=True)
