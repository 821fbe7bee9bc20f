"""Oracle: fp32 CPU restatement of the text conditioning stage (OpenCLIP text tower, penultimate layer).

TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED.  The reference only *names* this stage (configs/stable-diffusion/v2-inference.yaml:63-67,
``target: ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder``, ``layer: "penultimate"``) and calls it from
``LatentDiffusion.get_learned_conditioning`` (ldm/models/diffusion/ddpm.py:658-669); ``ldm/modules/encoders`` does
not exist in the reference tree and ``open_clip`` (unpinned there: no requirements file lists it) is not installed,
so there is no reference code to run and no fixture to pin this file against.  It restates the published algorithm
of that stage: open_clip's text tower (token + positional embedding, ``layers`` pre-LN residual blocks built on
``nn.MultiheadAttention`` with the additive upper-triangular -inf mask and a 4x GELU MLP, ``ln_final``), run the way
Stability's ``FrozenOpenCLIPEmbedder.encode_with_transformer`` runs it: blocks ``0 .. layers-1-layer_idx`` with
``layer_idx = 1`` for "penultimate" (0 for "last"), then ``ln_final``; no text projection, no pooling.

State-dict driven like oracle/model.py: ``sd`` holds the text tower's tensors under open_clip's names
(``token_embedding.weight``, ``positional_embedding``, ``transformer.resblocks.N.{ln_1,attn,ln_2,mlp}...``,
``ln_final.*``).
"""
import torch
import torch.nn.functional as F


def text_tower_forward(sd, tokens, heads, layers, layer="penultimate", prefix=""):
    """tokens int64 [B, T] -> fp32 [B, T, width]."""
    g = lambda k: sd[prefix + k].float()
    x = g("token_embedding.weight")[tokens] + g("positional_embedding")           # [B, T, C]
    b, t, c = x.shape
    d = c // heads
    mask = torch.full((t, t), float("-inf")).triu_(1)                              # query i sees keys 0..i
    n_run = layers - (1 if layer == "penultimate" else 0)
    for i in range(n_run):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (c,), g(p + "ln_1.weight"), g(p + "ln_1.bias"), 1e-5)
        qkv = h @ g(p + "attn.in_proj_weight").t() + g(p + "attn.in_proj_bias")
        q, k, v = (z.view(b, t, heads, d).transpose(1, 2) for z in qkv.chunk(3, dim=-1))   # [B, H, T, d]
        w = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + mask, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(b, t, c)
        x = x + a @ g(p + "attn.out_proj.weight").t() + g(p + "attn.out_proj.bias")
        h = F.layer_norm(x, (c,), g(p + "ln_2.weight"), g(p + "ln_2.bias"), 1e-5)
        f = F.gelu(h @ g(p + "mlp.c_fc.weight").t() + g(p + "mlp.c_fc.bias"))
        x = x + f @ g(p + "mlp.c_proj.weight").t() + g(p + "mlp.c_proj.bias")
    return F.layer_norm(x, (c,), g("ln_final.weight"), g("ln_final.bias"), 1e-5)
