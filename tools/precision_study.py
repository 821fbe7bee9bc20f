#!/usr/bin/env python3
"""CPU study (development aid; uses the oracle, never the product path): what the 16-bit formats cost end to end.

For each weight recipe and UNet configuration, max|a-b|/max|b| (and rms-relative) of the oracle run with
  operands   : only the two operands of every contraction rounded (oracle.model.operand_precision) - the floor of an
               MFMA path with that operand type, whatever else it stores in fp32
  + storage  : additionally the residual stream / norm inputs rounded (oracle.model.storage_precision) - what the
               HIP path does when it keeps the stream in a 16-bit type
against the plain fp32 oracle.

Weight recipes:
  synth  : tests/synth.py - every tensor randn * fan_in^-0.5 (biases 0.05, norm scales 1 +- 0.1)
  init8d : SURVEY.md §8(d) - torch.manual_seed(0), default nn init, the zero_module'd tensors re-randomised with
           Generator(1), randn * 0.02
usage: precision_study.py [tiny|sd2] [synth|init8d]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import model as OM
from tests import synth


def weights(cfg, recipe):
    from stablediffusion_amd.unet import UNetModel
    if recipe == "synth":
        with torch.device("meta"):
            shapes = {k: v.shape for k, v in UNetModel(**cfg).state_dict().items()}
        return {k: synth.synth_tensor("unet." + k, s) for k, s in shapes.items()}
    return synth.init8d_state_dict(UNetModel, cfg)


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    recipe = sys.argv[2] if len(sys.argv) > 2 else "synth"
    torch.set_num_threads(8)
    cfg = synth.TINY_UNET if name == "tiny" else synth.SD2_UNET
    sd = weights(cfg, recipe)
    if name == "tiny":
        x = synth.synth_input("tiny_x", (2, 4, 16, 16)); t = torch.tensor([981, 21]); ctx = synth.synth_input("tiny_ctx", (2, 7, 96))
    else:
        x = synth.synth_input("sd2_x", (1, 4, 32, 32)); t = torch.tensor([501]); ctx = synth.synth_input("sd2_ctx", (1, 77, 1024))
    B, H = torch.bfloat16, torch.float16
    with torch.no_grad():
        ref = OM.unet_forward(sd, cfg, x, t, ctx)
        print(f"{name} / {recipe}: |ref| max {ref.abs().max():.4f} std {ref.std():.4f}")
        for label, op, st in (("bf16 operands", B, None), ("bf16 operands + bf16 storage", B, B),
                              ("bf16 operands + fp16 storage", B, H), ("fp16 operands", H, None),
                              ("fp16 operands + fp16 storage", H, H)):
            with OM.operand_precision(op), OM.storage_precision(st):
                y = OM.unet_forward(sd, cfg, x.to(op).float() if st is not None else x, t, ctx)
            m, r = rel(y, ref)
            print(f"  {label:34s} max-rel {m:.3e}  rms-rel {r:.3e}")
