#!/usr/bin/env python3
"""Profiling target for per-shape HBM traffic of the conv3x3 kernel: launches each of the step's time-dominant conv
shapes LAUNCHES times in a fixed order (fp16) and writes the order + algorithmic bytes of each shape to
gpurun_out/pmc_conv_shapes.json.  Run once under `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE`;
tools/summarize_pmc.py maps the conv dispatches back to shapes by their order."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from stablediffusion_amd import ops

LAUNCHES = 5
SHAPES = [  # b, h, w, cin, cout  - where it runs
    (8, 512, 512, 128, 128), (8, 128, 128, 512, 512), (8, 256, 256, 256, 256), (8, 64, 64, 512, 512),   # VAE decoder
    (16, 64, 64, 320, 320), (16, 16, 16, 1280, 1280), (16, 32, 32, 640, 640), (16, 8, 8, 1280, 1280),    # UNet
    (16, 64, 64, 640, 320), (8, 512, 512, 128, 3),
]
dev = torch.device("cuda:0")
F16 = torch.float16
table = []
for b, h, w, cin, cout in SHAPES:
    x = (torch.randn(b, h, w, cin, device=dev)).to(F16)
    pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, torch.randn(cout, device=dev), dtype=F16)
    for _ in range(LAUNCHES):
        y = ops.conv3x3(x, pw, out_nchw_f32=(cout == 3))      # conv_out writes fp32 NCHW
    torch.cuda.synchronize()
    m = b * h * w
    table.append({"shape": [b, h, w, cin, cout], "M": m, "N": cout, "K": 9 * cin, "launches": LAUNCHES,
                  "algorithmic_bytes": 2 * (m * cin + 9 * cin * cout) + (4 if cout == 3 else 2) * m * cout, "flops": 2 * m * cout * 9 * cin})
    del x, pw, y
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "pmc_conv_shapes.json"), "w") as fh:
    json.dump(table, fh)
