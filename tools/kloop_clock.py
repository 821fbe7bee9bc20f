#!/usr/bin/env python3
"""Development aid for a -DSDMI_DEBUG_CLOCKS=1 build (SDMI_LIB=<path>): the shader clock the part holds inside the tiled
kernel's k-loop - block 0 stamps clock64() (shader cycles) and wall_clock64() (100 MHz) around its k-loop.
usage: kloop_clock.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusion_amd import ops, hip

dev = torch.device("cuda:0")
lib = hip.lib()
buf = (C.c_longlong * 2)()
CASES = [("conv", (16, 64, 64, 320, 320)), ("conv", (16, 32, 32, 640, 640)), ("conv", (16, 16, 16, 1280, 1280)),
         ("lin", (16384, 640, 2560)), ("lin", (4096, 10240, 1280))]
for kind, a in CASES:
    if kind == "conv":
        b, h, w, cin, cout = a
        x = torch.randn(b, h, w, cin, device=dev).to(torch.bfloat16)
        pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, torch.randn(cout, device=dev))
        fn, getter = (lambda: ops.conv3x3(x, pw)), lib.sdmi_debug_clocks_m1_bf16
    else:
        m, n, k = a
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        pw = ops.pack_linear(torch.randn(n, k, device=dev) * 0.02, torch.randn(n, device=dev))
        fn, getter = (lambda: ops.linear(x, pw)), lib.sdmi_debug_clocks_m0_bf16
    for _ in range(30):          # back to back: the clock settles to what the kernel sustains
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    assert getter(buf) == 0
    cyc, ticks = buf[0], buf[1]
    print(f"{kind} {a}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us/launch; block 0 k-loop {cyc} shader cycles in {ticks / 100:.1f} us "
          f"= {cyc / max(ticks, 1) * 100:.0f} MHz", flush=True)
