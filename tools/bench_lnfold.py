#!/usr/bin/env python3
"""The LayerNorm-consuming projections of the UNet, three ways (development aid):
folded (pack_*(ln=...): the kernel gathers the row sums), unfolded (sdmi_layernorm or the panel kernel + GEMM), and the bare
GEMM without any LayerNorm (what the row sums cost).  usage: bench_lnfold.py [iters=20] [B1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F16 = torch.float16
rows = (8192, 2048, 512, 128) if len(sys.argv) > 2 else (65536, 16384, 4096, 1024)
SHAPES = []
for m, c in zip(rows, (320, 640, 1280, 1280)):
    SHAPES += [("lin", m, 3 * c, c), ("lin", m, c, c), ("geglu", m, 4 * c, c)]
ROT = 6


def timed(fn):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for kind, m, n, k in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    xs = [(torch.randn(m, k, generator=g) * 1.5 + 0.3).to(dev).to(F16) for _ in range(ROT)]
    gamma, beta = (1 + 0.1 * torch.randn(k, generator=g)).to(dev), (0.1 * torch.randn(k, generator=g)).to(dev)
    ln = (gamma, beta, 1e-5)
    if kind == "geglu":
        w, b = (torch.randn(2 * n, k, generator=g) * 0.05).to(dev), torch.randn(2 * n, generator=g).to(dev)
        folded, plain = ops.pack_geglu(w, b, dtype=F16, ln=ln), ops.pack_geglu(w, b, dtype=F16)
    else:
        w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
        folded, plain = ops.pack_linear(w, None, dtype=F16, ln=ln), ops.pack_linear(w, None, dtype=F16)
    # the statistics as the model has them: left by the launch that produced the input (slot count of a K -> K projection)
    prod = ops.pack_linear((torch.randn(k, k, generator=g) * k ** -0.5).to(dev), None, dtype=F16)
    pairs = [ops.linear(x, prod, row_stats=True) for x in xs]
    xs, sts = [p_[0] for p_ in pairs], [p_[1] for p_ in pairs]
    t_f = t_u = t_b = 1e9
    for _ in range(3):          # round-robin, best of three: the first timed loop after an idle phase reads slow
        t_b = min(t_b, timed(lambda i: ops.linear(xs[i % ROT], plain)))
        t_f = min(t_f, timed(lambda i: ops.linear(xs[i % ROT], folded, ln=ln, stats=sts[i % ROT])))
        t_u = min(t_u, timed(lambda i: ops.linear(xs[i % ROT], plain, ln=ln)))
    print(f"{kind:6s} M {m:6d} N {n:5d} K {k:5d} slots {sts[0][1]:2d}: folded {t_f:7.1f} us   unfolded (LN + GEMM) {t_u:7.1f}   bare GEMM {t_b:7.1f}", flush=True)

# the producers of the stream: to_out / proj_in (K = C) and ff2 (K = 4 C) with the residual, with and without row_stats
for m, c in zip(rows, (320, 640, 1280, 1280)):
    for k in (c, 4 * c):
        g = torch.Generator(device="cpu").manual_seed(2)
        xs = [torch.randn(m, k, generator=g).to(dev).to(F16) for _ in range(ROT)]
        rs = [torch.randn(m, c, generator=g).to(dev).to(F16) for _ in range(ROT)]
        pw = ops.pack_linear((torch.randn(c, k, generator=g) * 0.05).to(dev), torch.randn(c, generator=g).to(dev), dtype=F16)
        t_p = t_s = 1e9
        for _ in range(3):
            t_p = min(t_p, timed(lambda i: ops.linear(xs[i % ROT], pw, res=rs[i % ROT])))
            t_s = min(t_s, timed(lambda i: ops.linear(xs[i % ROT], pw, res=rs[i % ROT], row_stats=True)))
        print(f"producer M {m:6d} N {c:5d} K {k:5d}: plain {t_p:7.1f} us   with row_stats {t_s:7.1f}", flush=True)
