#!/usr/bin/env python3
"""Development aid: the 64-column tile (tile_n = 64, 128 | 64 rows) against the planned launch on the batch-1 linear shapes:
max |difference| of the outputs (and of the row statistics) and the time of each.  usage: python tools/small_tile_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16


def timed(fn, iters=20):
    """us per call, replayed from a hipGraph of `iters` calls (eager timing is bound by the host at ~12 us a call)"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def main():
    ops.SMALL_TILE = False
    for dt in ((F16,) if os.environ.get('F16_ONLY') else (F16, torch.bfloat16)):
        for (m, n, k) in ((512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (128, 1280, 1280), (512, 3840, 1280),
                          (2048, 1920, 640), (8192, 960, 320), (512, 1280, 5120), (2048, 640, 2560), (8192, 320, 1280),
                          (200, 320, 320), (1000, 1288, 640)):
            torch.manual_seed(m + n)
            x = torch.randn(m, k, device=dev).to(dt)
            res = torch.randn(m, n, device=dev).to(dt)
            pw = ops.pack_linear(torch.randn(n, k, device=dev) * 0.03, torch.randn(n, device=dev), dtype=dt)
            stats = dt == F16 and n % 8 == 0
            ops.FORCE_TILE_N = ops.FORCE_TILE_M = 0
            base = ops.linear(x, pw, res=res, row_stats=stats)
            t0 = timed(lambda: ops.linear(x, pw, res=res, row_stats=stats))
            line = f"{str(dt)[6:]:9s} {m:6d} {n:5d} {k:5d}  plan {t0:6.1f} us"
            for tm in (128, 64):
                for sk in (1, 2, 4):
                    if sk > 1 and k // 64 // sk < 2:
                        continue
                    ops.FORCE_TILE_N, ops.FORCE_TILE_M = 64, tm
                    ops._auto_split_k = lambda m_, n_, k_, t=128, s_=sk: s_
                    try:
                        out = ops.linear(x, pw, res=res, row_stats=stats)
                        t1 = timed(lambda: ops.linear(x, pw, res=res, row_stats=stats))
                    except ops.hip.SdmiError as e:
                        line += f" | {tm}x64/{sk} n/a"
                        continue
                    finally:
                        ops._auto_split_k = ops._AUTO_SPLIT_K
                    if stats:
                        (o, st), (b, bst) = out, base
                        d = (o.float() - b.float()).abs().max().item()
                        ds = "-" if st is None or bst is None else \
                            f"{((st[0].sum(1) - bst[0].sum(1)).abs().max() / bst[0].sum(1).abs().max()).item():.1e}"
                    else:
                        d, ds = (out.float() - base.float()).abs().max().item(), "-"
                    line += f" | {tm}x64/{sk} {t1:5.1f} us d={d:.1e} s={ds}"
            print(line, flush=True)



if __name__ == "__main__":
    main()
