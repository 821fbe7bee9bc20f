#!/usr/bin/env python3
"""Micro-benchmark of sdmi_gemm on the shapes of the SD2 path (development aid).
usage: python tools/bench_gemm.py [iters] [conv|lin]   -> one line per shape: us, TFLOP/s
env TILE_N=128 forces the 128-column tile; SPLITK=<n> forces the split-K factor; SWEEP=1 tries several of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops

dev = torch.device("cuda:0")
ops.FORCE_WIDE = int(os.environ.get("WIDE", "0"))      # 1 = the 256 x 256 x 32 tile where it applies
ops.FORCE_TILE_N = 64 if os.environ.get("TILE_N") == "64" else 0      # the small tile (with TILE_M=64|128 or CONFIGS=64:1,128:1)
ops.FORCE_ALGO = int(os.environ.get("ALGO", "0"))      # 1 = tiled kernel only, 2 = panel kernel (fails where it does not apply)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
BF16 = torch.bfloat16
SHAPES = [  # kind, args
    ("conv", (16, 64, 64, 320, 320)), ("conv", (16, 32, 32, 640, 640)), ("conv", (16, 16, 16, 1280, 1280)),
    ("conv", (16, 8, 8, 1280, 1280)), ("conv", (16, 64, 64, 640, 320)), ("conv", (8, 256, 256, 256, 256)),
    ("conv", (8, 128, 128, 512, 512)), ("conv", (8, 512, 512, 128, 128)),
    ("lin", (65536, 320, 320)), ("lin", (65536, 2560, 320)), ("lin", (65536, 320, 1280)), ("lin", (16384, 640, 640)),
    ("lin", (16384, 5120, 640)), ("lin", (4096, 1280, 1280)), ("lin", (4096, 10240, 1280)), ("lin", (8192, 8192, 8192)),
]
SHAPES += [("conv", (16, 16, 16, 2560, 1280)), ("conv", (16, 8, 8, 2560, 1280)), ("conv", (16, 32, 32, 1280, 640)),
           ("conv", (16, 64, 64, 960, 320))]
SHAPES += [("linnr", (65536, 960, 320)), ("linnr", (65536, 320, 320)), ("linnr", (16384, 1920, 640)), ("linnr", (16384, 640, 640)),
           ("lin", (65536, 960, 320)), ("lin", (8192, 320, 320)), ("lin", (147456, 320, 320)),
           ("geglu", (65536, 1280, 320)), ("geglu", (16384, 2560, 640)), ("geglu", (4096, 5120, 1280))]
auto_split = ops._auto_split_k
configs = [(os.environ.get("TILE_N"), os.environ.get("SPLITK"), os.environ.get("TILE_M", "0"))]
SHAPES += [("linnr", (65536, 320, 640)), ("linnr", (65536, 320, 960)), ("lin", (16384, 640, 2560)), ("linnr", (16384, 640, 1280)),
           ("linnr", (16384, 640, 1920)), ("linnr", (4096, 3840, 1280)), ("lin", (4096, 1280, 5120)), ("linnr", (4096, 1280, 2560)),
           ("linnr", (4096, 1280, 1920)), ("linnr", (1024, 1280, 2560)), ("linnr", (1024, 3840, 1280)), ("lin", (1024, 1280, 5120))]
SHAPES += [("convout", (8, 512, 512, 128, 3)), ("convout", (16, 64, 64, 320, 4)), ("conv", (16, 16, 16, 640, 640)), ("lin", (1024, 1280, 1280)), ("conv", (16, 16, 16, 1920, 1280))]
if os.environ.get("B1"):       # the UNet-batch-2 (BASELINE configs[1]) shapes: M = 8192 / 2048 / 512 / 128 tokens
    SHAPES = [("lin", (8192, 320, 320)), ("linnr", (8192, 960, 320)), ("geglu", (8192, 1280, 320)), ("lin", (8192, 320, 1280)),
              ("lin", (2048, 640, 640)), ("linnr", (2048, 1920, 640)), ("geglu", (2048, 2560, 640)), ("lin", (2048, 640, 2560)),
              ("lin", (512, 1280, 1280)), ("linnr", (512, 3840, 1280)), ("geglu", (512, 5120, 1280)), ("lin", (512, 1280, 5120)),
              ("lin", (128, 1280, 1280)), ("linnr", (128, 3840, 1280)), ("geglu", (128, 5120, 1280)), ("lin", (128, 1280, 5120)),
              ("conv", (2, 64, 64, 320, 320)), ("conv", (2, 32, 32, 640, 640)), ("conv", (2, 16, 16, 1280, 1280)),
              ("conv", (2, 8, 8, 1280, 1280)), ("conv", (2, 8, 8, 2560, 1280)), ("conv", (2, 16, 16, 2560, 1280)),
              ("conv", (2, 32, 32, 1280, 640)), ("conv", (2, 64, 64, 640, 320))]
if os.environ.get("CONFIGS"):      # CONFIGS=256:8,128:4  -> (tile_m, split) pairs; SHAPE=<substring of the shape tuple> filters
    configs = [(None, c.split(":")[1], c.split(":")[0]) for c in os.environ["CONFIGS"].split(",")]
if os.environ.get("SHAPE"):
    SHAPES = [sh for sh in SHAPES if os.environ["SHAPE"] in str(sh[1])]
if os.environ.get("SWEEP"):
    configs = [(None, s, tm) for tm in ("128", "129", "256") for s in (None, "1", "2", "4")]
for (kind, a), (tile, split, tile_m) in [(sh, c) for sh in SHAPES for c in configs]:
    ops.FORCE_TILE_M = int(tile_m)
    if only and only != kind and not (only == "lin" and kind == "linnr"):
        continue
    tn = 128 if tile == "128" else None
    ops._auto_split_k = (lambda m, n, k, t=128, s_=split: min(int(s_), k // 128)) if split else auto_split
    if kind in ("conv", "convout"):
        b, h, w, cin, cout = a
        x = torch.randn(b, h, w, cin, device=dev).to(BF16)
        pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, torch.randn(cout, device=dev), tile_n=tn)
        fn = (lambda: ops.conv3x3(x, pw, out_nchw_f32=True)) if kind == "convout" else (lambda: ops.conv3x3(x, pw))
        flops = 2 * b * h * w * cout * 9 * cin
    elif kind == "geglu":
        m, inner, k = a
        x = torch.randn(m, k, device=dev).to(BF16)
        pw = ops.pack_geglu(torch.randn(2 * inner, k, device=dev) * 0.02, torch.randn(2 * inner, device=dev))
        fn = lambda: ops.linear(x, pw)
        flops = 2 * m * 2 * inner * k
    else:
        m, n, k = a
        x = torch.randn(m, k, device=dev).to(BF16)
        pw = ops.pack_linear(torch.randn(n, k, device=dev) * 0.02, torch.randn(n, device=dev), tile_n=tn)
        res = torch.randn(m, n, device=dev).to(BF16) if kind == "lin" else None
        fn = lambda: ops.linear(x, pw, res=res)
        if os.environ.get("ROTATE"):      # ROTATE=n: cycle through n input / residual sets (no cache-warm operands)
            nrot = int(os.environ["ROTATE"])
            xs = [torch.randn(m, k, device=dev).to(BF16) for _ in range(nrot)]
            rs = [torch.randn(m, n, device=dev).to(BF16) if kind == "lin" else None for _ in range(nrot)]
            cnt = [0]
            def fn():
                i = cnt[0] % nrot
                cnt[0] += 1
                return ops.linear(xs[i], pw, res=rs[i])
        flops = 2 * m * n * k
    try:
        for _ in range(3):
            fn()
    except ops.hip.SdmiError as e:
        print(f"{kind:5s} {str(a):32s} n/a ({e})", flush=True)
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{kind:5s} {str(a):32s} tile_n={pw.tile_n} tile_m={tile_m} split={split or 'auto'} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s",
          flush=True)
