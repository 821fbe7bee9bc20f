"""Text conditioning stage: FrozenOpenCLIPEmbedder on the HIP path vs the oracle.

The reference names the stage (v2-inference.yaml:63-67) but does not contain it, and open_clip is not installed:
PARITY UNPINNED against the reference.  What can be checked is checked: the oracle's attention block against
PyTorch's own nn.MultiheadAttention (the module open_clip builds its blocks from) on CPU, and the HIP path against
the oracle on the GPU at the fp32-mode tolerance.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import text_encoder as OT

FP32_GATE = 1e-5        # north star: "<= 1e-5 in fp32 mode"; the stage runs in fp32 mode


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _tower(width, heads, layers, vocab, layer="penultimate", seed=1):
    from stablediffusion_amd.encoders import FrozenOpenCLIPEmbedder
    m = FrozenOpenCLIPEmbedder(layer=layer, width=width, heads=heads, layers=layers, vocab_size=vocab)
    return m.reset_parameters(seed)


def _tokens(n, vocab, seed=0):
    from stablediffusion_amd.encoders import wrap_token_ids, SOT_TOKEN, EOT_TOKEN
    g = torch.Generator().manual_seed(seed)
    lists = [torch.randint(1, vocab - 2, (int(torch.randint(0, 90, (1,), generator=g)),), generator=g).tolist()
             for _ in range(n)]
    return wrap_token_ids(lists, 77, sot=vocab - 2, eot=vocab - 1)


def test_oracle_blocks_match_torch_multihead_attention():
    """The oracle's restatement vs the torch modules open_clip's ResidualAttentionBlock is made of."""
    torch.manual_seed(0)
    c, heads, layers, vocab, t = 128, 2, 3, 300, 77
    m = _tower(c, heads, layers, vocab, layer="last")
    sd = {k[len("model."):]: v for k, v in m.state_dict().items()}
    tokens = _tokens(3, vocab)
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    mask = torch.full((t, t), float("-inf")).triu_(1)
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        mha = nn.MultiheadAttention(c, heads, batch_first=True)
        mha.in_proj_weight.data.copy_(sd[p + "attn.in_proj_weight"]); mha.in_proj_bias.data.copy_(sd[p + "attn.in_proj_bias"])
        mha.out_proj.weight.data.copy_(sd[p + "attn.out_proj.weight"]); mha.out_proj.bias.data.copy_(sd[p + "attn.out_proj.bias"])
        h = F.layer_norm(x, (c,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        x = x + mha(h, h, h, need_weights=False, attn_mask=mask)[0]
        h = F.layer_norm(x, (c,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        x = x + F.linear(F.gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])),
                         sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    want = F.layer_norm(x, (c,), sd["ln_final.weight"], sd["ln_final.bias"])
    with torch.no_grad():
        got = OT.text_tower_forward(sd, tokens, heads, layers, layer="last")
    assert rel_err(got, want.detach()) < 2e-6
    # "penultimate" = the same tower one block shorter
    with torch.no_grad():
        pen = OT.text_tower_forward(sd, tokens, heads, layers, layer="penultimate")
        short = OT.text_tower_forward(sd, tokens, heads, layers - 1, layer="last")
    assert torch.equal(pen, short)


def test_embedder_surface_cpu():
    from stablediffusion_amd.encoders import FrozenOpenCLIPEmbedder, wrap_token_ids, SOT_TOKEN, EOT_TOKEN
    from stablediffusion_amd.util import instantiate_from_config
    m = instantiate_from_config({"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                                 "params": {"freeze": True, "layer": "penultimate"}})
    assert isinstance(m, FrozenOpenCLIPEmbedder) and m.layer_idx == 1 and not any(p.requires_grad for p in m.parameters())
    keys = set(m.state_dict())
    for k in ("model.token_embedding.weight", "model.positional_embedding", "model.ln_final.bias",
              "model.transformer.resblocks.23.attn.in_proj_weight", "model.transformer.resblocks.0.attn.out_proj.bias",
              "model.transformer.resblocks.5.mlp.c_fc.weight", "model.transformer.resblocks.5.mlp.c_proj.bias",
              "model.transformer.resblocks.7.ln_2.weight"):
        assert k in keys
    assert m.state_dict()["model.transformer.resblocks.5.mlp.c_fc.weight"].shape == (4096, 1024)
    with pytest.raises(NotImplementedError):
        FrozenOpenCLIPEmbedder(layer="pooled")
    with pytest.raises(NotImplementedError):
        FrozenOpenCLIPEmbedder(width=100, heads=2, layers=1)
    t = wrap_token_ids([[7, 8], list(range(1, 200))])
    assert t.shape == (2, 77) and t[0].tolist()[:5] == [SOT_TOKEN, 7, 8, EOT_TOKEN, 0] and int(t[1, -1]) == EOT_TOKEN
    # a full open_clip state dict loads: foreign entries are ignored, the text tower's are all consumed
    small = _tower(64, 1, 2, 100)
    sd = dict(small.state_dict())
    sd["model.visual.conv1.weight"] = torch.zeros(1); sd["model.logit_scale"] = torch.zeros(()); sd["model.attn_mask"] = torch.zeros(77, 77)
    sd["model.text_projection"] = torch.zeros(64, 64)
    fresh = FrozenOpenCLIPEmbedder(width=64, heads=1, layers=2, vocab_size=100)
    r = fresh.load_state_dict(sd)
    assert not r.missing_keys and not r.unexpected_keys and fresh.weights_loaded
    with pytest.raises(RuntimeError, match="GPU"):
        fresh(torch.zeros(1, 77, dtype=torch.long))


def test_bpe_tokenizer_on_a_synthetic_vocabulary(tmp_path):
    """The real merge list ships with open_clip (absent): a two-merge vocabulary with hand-computed ids exercises
    the algorithm - byte spelling, end-of-word mark, greedy merges by rank, lower-casing, start/end/padding."""
    import gzip
    from stablediffusion_amd.encoders import SimpleTokenizer, FrozenOpenCLIPEmbedder
    path = tmp_path / "bpe.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as fh:
        fh.write("#version: synthetic\nl o\nlo w</w>\n")
    tok = SimpleTokenizer(str(path), n_merges=2)
    # vocabulary: 256 byte characters ('!' first), 256 with </w>, "lo" = 512, "low</w>" = 513, <sot> = 514, <eot> = 515
    assert (tok.sot, tok.eot) == (514, 515)
    assert tok.encode("low") == [513] and tok.encode("Low   LOW") == [513, 513]
    assert tok.encode("lower") == [512, ord("w") - 33, ord("e") - 33, 256 + ord("r") - 33]
    # a word that ends in "o": the end-of-word mark makes it a different symbol, so ("l", "o") does not apply
    assert tok.encode("lo!") == [ord("l") - 33, 256 + ord("o") - 33, 256 + 0]
    assert tok.encode("l&amp;amp;o") == tok.encode("l&o") == [256 + ord("l") - 33, 256 + ord("&") - 33, 256 + ord("o") - 33]
    t = tok(["low", "low low"])
    assert t.shape == (2, 77) and t[0].tolist()[:4] == [514, 513, 515, 0] and t[1].tolist()[:5] == [514, 513, 513, 515, 0]
    m = FrozenOpenCLIPEmbedder(width=64, heads=1, layers=1, vocab_size=516, bpe_path=str(path))
    assert m.encodes_text and m.tokenizer.encode("low") == [513]


# ----------------------------------------------------------------------------- GPU

gpu = pytest.mark.gpu


@gpu
def test_text_tower_pieces():
    from stablediffusion_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    # token + positional embedding
    table, pos = torch.randn(500, 128, generator=g), torch.randn(77, 128, generator=g)
    ids = torch.randint(0, 500, (5, 77), generator=g)
    got = ops.embed_tokens(ids.to(dev), table.to(dev), pos.to(dev)).cpu()
    assert torch.equal(got, table[ids] + pos)
    # causal fp32 attention, ragged against the 64-key staging (77 = 64 + 13)
    b, heads, t = 3, 2, 77
    qkv = torch.randn(b * t, 3 * heads * 64, generator=g)
    c = heads * 64
    q, k, v = (z.view(b, t, heads, 64).transpose(1, 2) for z in qkv.split(c, dim=-1))
    mask = torch.full((t, t), float("-inf")).triu_(1)
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + mask, -1) @ v).transpose(1, 2).reshape(b * t, c)
    qd = qkv.to(dev)
    got = ops.attention_d64(qd[:, :c], qd[:, c:2 * c], qd[:, 2 * c:], heads, b, t, t, 0.125, causal=True).cpu()
    assert rel_err(got, want) < 2e-6
    with pytest.raises(Exception):
        ops.attention_d64(qd[:, :c].bfloat16(), qd[:, c:2 * c].bfloat16(), qd[:, 2 * c:].bfloat16(), heads, b, t, t, 0.125,
                          causal=True)
    # GELU as a GEMM epilogue activation: fp32 mode (exact erf) and 16-bit (12-instruction erf)
    x, w, bias = torch.randn(231, 128, generator=g), torch.randn(320, 128, generator=g) * 0.1, torch.randn(320, generator=g)
    want = F.gelu(F.linear(x, w, bias))
    got = ops.linear(x.to(dev), ops.pack_linear(w.to(dev), bias.to(dev), dtype=torch.float32), act=2).cpu()
    assert rel_err(got, want) < 2e-6
    for dt, tol in ((torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)):
        got = ops.linear(x.to(dev).to(dt), ops.pack_linear(w.to(dev), bias.to(dev), dtype=dt), act=2).float().cpu()
        want16 = F.gelu(F.linear(x.to(dt).float(), w.to(dt).float(), bias))
        assert rel_err(got, want16) < tol


@gpu
@pytest.mark.parametrize("layer", ["penultimate", "last"])
def test_text_encoder_small_vs_oracle(layer):
    dev = torch.device("cuda:0")
    m = _tower(128, 2, 3, 1000, layer=layer)
    tokens = _tokens(5, 1000)
    sd = {k[len("model."):]: v for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = OT.text_tower_forward(sd, tokens, 2, 3, layer=layer)
    got = m.encode(tokens.to(dev)).cpu()
    assert got.shape == (5, 77, 128) and got.dtype == torch.float32
    e = rel_err(got, want)
    print(f"[parity] text tower small layer={layer}: {e:.3e}")
    assert e < FP32_GATE
    # moving the module invalidates nothing visible: same result after .to(dev)
    assert torch.equal(m.to(dev).encode(tokens.to(dev)).cpu(), got)
    with pytest.raises(ValueError):
        m.encode(torch.full((1, 77), 1000, dtype=torch.long, device=dev))      # id outside the vocabulary
    with pytest.raises(ValueError):
        m.encode(tokens[:, :50].to(dev))


@gpu
def test_text_encoder_vit_h_vs_oracle():
    """The tower v2-inference.yaml names: width 1024, 16 heads, 24 blocks (23 run), vocabulary 49408."""
    dev = torch.device("cuda:0")
    m = _tower(1024, 16, 24, 49408)
    tokens = _tokens(2, 49408, seed=5)
    sd = {k[len("model."):]: v for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = OT.text_tower_forward(sd, tokens, 16, 24, layer="penultimate")
    got = m.encode(tokens.to(dev)).cpu()
    e = rel_err(got, want)
    print(f"[parity] text tower ViT-H/14 penultimate: {e:.3e}")
    assert got.shape == (2, 77, 1024) and e < FP32_GATE


@gpu
def test_conditioning_through_latent_diffusion():
    """tokens -> get_learned_conditioning -> cross-attention context of the UNet, via a checkpoint in upstream
    spelling (cond_stage_model.model.*)."""
    from tests import synth
    from stablediffusion_amd.util import instantiate_from_config
    dev = torch.device("cuda:0")
    unet = dict(synth.TINY_UNET, context_dim=64)
    cfg = {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.012, "timesteps": 1000, "scale_factor": 0.18215,
        "conditioning_key": "crossattn",
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": unet},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                               "params": {"embed_dim": 4, "ddconfig": dict(synth.TINY_DDCONFIG)}},
        "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                              "params": {"layer": "penultimate", "width": 64, "heads": 1, "layers": 3, "vocab_size": 400}}}}
    model = instantiate_from_config(cfg)
    donor = _tower(64, 1, 3, 400, seed=9)
    ck = {"cond_stage_model." + k: v for k, v in donor.state_dict().items()}
    r = model.load_state_dict(ck, strict=False)
    assert not [k for k in r.missing_keys if k.startswith("cond_stage_model.")] and model.cond_stage_model.weights_loaded
    model = model.to(dev)
    tokens = _tokens(2, 400, seed=2)
    c = model.get_learned_conditioning(tokens.to(dev))
    sd = {k[len("model."):]: v for k, v in donor.state_dict().items()}
    with torch.no_grad():
        want = OT.text_tower_forward(sd, tokens, 1, 3, layer="penultimate")
    assert rel_err(c.cpu(), want) < FP32_GATE
    x = torch.randn(2, 4, 16, 16, device=dev)
    eps = model.apply_model(x, torch.tensor([10, 500], device=dev), c)
    assert eps.shape == x.shape and bool(torch.isfinite(eps).all())
