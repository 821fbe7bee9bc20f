#!/usr/bin/env python3
"""Experiment: one UNet batch of 2B on one stream against two half batches on two streams (development aid).
usage: two_stream.py [B=8] [iters=10]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
model = bench.build_model(dev, "eps", "fp16")
h = 64
x = torch.randn(2 * B, 4, h, h, device=dev)
c = torch.randn(2 * B, 77, 1024, device=dev)
ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print(f"one stream, batch {2 * B}: {timed(lambda: model.apply_model(x, ts, c)):.2f} ms", flush=True)
for parts in (2, 4):
    n = 2 * B // parts
    xs, cs, tss = x.split(n), [t.contiguous() for t in c.split(n)], ts.split(n)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    print(f"one stream, batch {n} alone: {timed(lambda: model.apply_model(xs[0], tss[0], cs[0])):.2f} ms", flush=True)

    def multi():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                model.apply_model(xs[i], tss[i], cs[i])
        for s in streams:
            cur.wait_stream(s)
    print(f"{parts} streams x batch {n}: {timed(multi):.2f} ms", flush=True)
