#!/usr/bin/env python3
"""Development aid: conv3x3 on the 128 x 64 block (tile_n = 64) against the planned launch on the batch-1 conv shapes:
hipGraph-replayed time of each (split, tile) pair and the max difference of the outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from small_tile_check import timed

dev = torch.device("cuda:0")
F16 = torch.float16
shapes = [(2, 8, 8, 1280, 1280), (2, 8, 8, 2560, 1280), (2, 16, 16, 1280, 1280), (2, 16, 16, 2560, 1280), (2, 32, 32, 640, 640),
          (2, 32, 32, 1280, 640), (2, 64, 64, 320, 320), (2, 64, 64, 640, 320), (16, 8, 8, 1280, 1280), (16, 8, 8, 2560, 1280),
          (4, 8, 8, 1280, 1280), (4, 8, 8, 2560, 1280), (8, 8, 8, 1280, 1280), (1, 64, 64, 320, 320), (2, 32, 32, 320, 640),
          (2, 16, 16, 640, 1280), (4, 64, 64, 320, 320), (2, 64, 64, 8, 320), (1, 64, 64, 8, 320)]
if len(sys.argv) > 1:
    shapes = [sh for sh in shapes if sys.argv[1] in str(sh)]
for (b, h, w, cin, cout) in shapes:
    x = torch.randn(b, h, w, cin, device=dev).to(F16)
    pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, torch.randn(cout, device=dev), dtype=F16)
    ops.FORCE_TILE_N = ops.FORCE_TILE_M = 0
    ops.SMALL_TILE = False
    base = ops.conv3x3(x, pw)
    t0 = timed(lambda: ops.conv3x3(x, pw))
    plan = ops._plan_gemm(b * h * w, cout, pw.k, pw.tile_n)
    new = ops._plan_gemm(b * h * w, cout, pw.k, pw.tile_n, False, False, True, True)
    line = f"conv {str((b, h, w, cin, cout)):28s} plan {plan[0]}x{plan[3]}/{plan[1]} {t0:6.1f} us (new plan {new[0]}x{new[3]}/{new[1]})"
    nk = pw.k // 64
    for sk in (1, 2, 4, 6, 8, 12, 16, 24):
        if sk > 1 and nk // sk < 2:
            continue
        ops.FORCE_TILE_N, ops.FORCE_TILE_M = 64, 128
        ops._auto_split_k = lambda m_, n_, k_, t=128, s_=sk: s_
        try:
            out = ops.conv3x3(x, pw)
            t1 = timed(lambda: ops.conv3x3(x, pw))
            line += f" | /{sk} {t1:5.1f} d={(out.float() - base.float()).abs().max().item():.0e}"
        except ops.hip.SdmiError:
            line += f" | /{sk} n/a"
        finally:
            ops._auto_split_k = ops._AUTO_SPLIT_K
    print(line, flush=True)
