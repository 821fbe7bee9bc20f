#!/usr/bin/env python3
"""CPU study: which bf16 roundings of the HIP pipeline cost how much end-to-end error.
Monkeypatches the oracle's conv/linear/norm helpers with bf16 rounding at chosen places and reports
max|a-b|/max|b| against the plain fp32 oracle.  (Development aid, not part of the product or tests.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import model as OM
from tests import synth

q = lambda x: x.bfloat16().float()

def run(cfgname, rw, rin, rout, rnorm, rstream, rx=True, remb=True):
    cfg = synth.TINY_UNET if cfgname == "tiny" else synth.SD2_UNET
    from stablediffusion_amd.unet import UNetModel
    with torch.device("meta"):
        shapes = {k: v.shape for k, v in UNetModel(**cfg).state_dict().items()}
    sd = {k: synth.synth_tensor("unet." + k, s) for k, s in shapes.items()}
    if cfgname == "tiny":
        x = synth.synth_input("tiny_x", (2, 4, 16, 16)); t = torch.tensor([981, 21]); ctx = synth.synth_input("tiny_ctx", (2, 7, 96))
    else:
        x = synth.synth_input("sd2_x", (1, 4, 32, 32)); t = torch.tensor([501]); ctx = synth.synth_input("sd2_ctx", (1, 77, 1024))
    with torch.no_grad():
        ref = OM.unet_forward(sd, cfg, x, t, ctx)
        o_conv, o_lin, o_gn, o_ln = OM._conv, OM._lin, OM._gn, F.layer_norm
        o_gnF = F.group_norm
        def conv(sd_, p, xx, stride=1):
            w = sd_[p + "weight"]; w = q(w) if rw else w
            xi = q(xx) if rin else xx
            y = F.conv2d(xi, w, sd_.get(p + "bias"), stride=stride, padding=w.shape[-1] // 2)
            return q(y) if rout else y
        def lin(sd_, p, xx):
            w = sd_[p + "weight"]; w = q(w) if rw else w
            xi = q(xx) if rin else xx
            y = F.linear(xi, w, sd_.get(p + "bias"))
            return q(y) if rout else y
        def gn(sd_, p, xx, eps):
            y = o_gn(sd_, p, xx, eps)
            return q(y) if rnorm else y
        OM._conv, OM._lin, OM._gn = conv, lin, gn
        try:
            y = OM.unet_forward(sd, cfg, q(x) if rx else x, t, q(ctx) if rin else ctx)
        finally:
            OM._conv, OM._lin, OM._gn = o_conv, o_lin, o_gn
    return ((y - ref).abs().max() / ref.abs().max()).item(), ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    for label, kw in [
        ("weights only", dict(rw=1, rin=0, rout=0, rnorm=0, rstream=0, rx=False)),
        ("x input only", dict(rw=0, rin=0, rout=0, rnorm=0, rstream=0, rx=True)),
        ("operands (acts in) only", dict(rw=0, rin=1, rout=0, rnorm=0, rstream=0, rx=False)),
        ("weights + operands", dict(rw=1, rin=1, rout=0, rnorm=0, rstream=0, rx=True)),
        ("weights + operands + all outputs", dict(rw=1, rin=1, rout=1, rnorm=1, rstream=1, rx=True)),
    ]:
        print(f"{name:5s} {label:40s} max-rel {run(name, **kw)[0]:.3e}  rms-rel {run(name, **kw)[1]:.3e}")
