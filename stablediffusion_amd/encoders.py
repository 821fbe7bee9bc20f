"""Text conditioning stage: the OpenCLIP text tower behind ``FrozenOpenCLIPEmbedder``.

The reference's config names ``ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder`` as the cond stage
(configs/stable-diffusion/v2-inference.yaml:63-67) and ``LatentDiffusion.get_learned_conditioning``
(ldm/models/diffusion/ddpm.py:658-669) calls its ``encode``; the module itself (``ldm/modules/encoders``) and the
``open_clip`` package it wraps are absent from the reference tree and from this machine.
What is restated here is therefore the published upstream behaviour the YAML refers to, with the parameter names of
an ``open_clip`` text tower so an SD2 checkpoint's ``cond_stage_model.model.*`` entries load unchanged:

    tokens [B, 77] -> token_embedding + positional_embedding
                   -> resblocks[0 .. layers-1-skip]   (pre-LN: x += attn(ln_1 x, causal mask); x += mlp(ln_2 x))
                   -> ln_final -> [B, 77, width]       (no projection, no pooling)

with ``skip = 1`` for ``layer="penultimate"`` (what v2-inference.yaml asks for) and 0 for ``"last"``.

Runs on the HIP path only, in the package's fp32 mode (``ops`` fp32 entry points; every contraction on the bf16 MFMA
GEMM through the exact three-piece split): the stage runs once per prompt, so exactness is worth more than speed.
The BPE vocabulary file (``bpe_simple_vocab_16e6.txt.gz``, shipped inside the ``open_clip`` / ``clip`` packages) is
not available offline: ``encode`` takes token ids (``wrap_token_ids`` adds start/end tokens and padding the way
``open_clip.tokenize`` does), and takes strings only when ``bpe_path`` (or ``$SDMI_BPE_VOCAB``) names that file -
``SimpleTokenizer`` below is the published byte-level BPE of CLIP, minus the optional ``ftfy`` text repair.
"""
import gzip
import html
import os

import torch
import torch.nn as nn

from . import ops

# text towers by open_clip model name: width, heads, layers (vocabulary and context length are shared)
ARCHS = {"ViT-H-14": (1024, 16, 24), "ViT-L-14": (768, 12, 12), "ViT-B-32": (512, 8, 12)}
SOT_TOKEN, EOT_TOKEN = 49406, 49407


def wrap_token_ids(id_lists, context_length=77, sot=SOT_TOKEN, eot=EOT_TOKEN):
    """[[bpe ids of one prompt], ...] -> int64 [N, context_length]: <sot> ids <eot>, zero padded; a prompt that does
    not fit is cut and its last token replaced by <eot> (open_clip.tokenize)."""
    out = torch.zeros(len(id_lists), context_length, dtype=torch.long)
    for i, ids in enumerate(id_lists):
        toks = [sot] + [int(t) for t in ids] + [eot]
        if len(toks) > context_length:
            toks = toks[:context_length]
            toks[-1] = eot
        out[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
    return out


def _bytes_to_unicode():
    """The reversible byte -> printable unicode character table of GPT-2 / CLIP byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class SimpleTokenizer:
    """CLIP's tokenizer: lower-cased, whitespace-normalised text is cut into words by a fixed pattern, every word is
    spelled in byte characters with an end-of-word mark and merged greedily by the rank of the merge list in
    ``bpe_path`` (gzip or plain text: a header line, then one ``left right`` merge per line).  Vocabulary order:
    256 byte characters, the same with ``</w>``, the merges, ``<start_of_text>``, ``<end_of_text>``."""

    def __init__(self, bpe_path, n_merges=49152 - 256 - 2):
        import regex
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as fh:
            lines = fh.read().split("\n")
        merges = [tuple(m.split()) for m in lines[1:n_merges + 1] if len(m.split()) == 2]
        self.byte_encoder = _bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.pat = regex.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            best = min(pairs, key=lambda pr: self.bpe_ranks.get(pr, float("inf")))
            if best not in self.bpe_ranks:
                break
            first, second = best
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = tuple(merged)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        text = " ".join(html.unescape(html.unescape(text)).split()).strip().lower()
        ids = []
        for token in self.pat.findall(text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def __call__(self, texts, context_length=77):
        if isinstance(texts, str):
            texts = [texts]
        return wrap_token_ids([self.encode(t) for t in texts], context_length, self.sot, self.eot)


class _Affine(nn.Module):
    """weight (+ bias) holder with nn.Linear / nn.LayerNorm / nn.Embedding's parameter names.  torch.empty: building
    the stage costs nothing (no initialisation pass, no touched pages) for configs that name it but never run it."""

    def __init__(self, *shape, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        if bias:
            self.bias = nn.Parameter(torch.empty(shape[0]))


class _Attention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight = [Wq; Wk; Wv])."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * width))
        self.out_proj = _Affine(width, width)


class _Mlp(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = _Affine(hidden, width)
        self.c_proj = _Affine(width, hidden)


class _ResBlock(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.ln_1 = _Affine(width)
        self.attn = _Attention(width)
        self.ln_2 = _Affine(width)
        self.mlp = _Mlp(width, hidden)


class _Transformer(nn.Module):
    def __init__(self, width, layers, hidden):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(width, hidden) for _ in range(layers)])


class _TextTower(nn.Module):
    def __init__(self, vocab_size, context_length, width, layers, hidden):
        super().__init__()
        self.token_embedding = _Affine(vocab_size, width, bias=False)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = _Transformer(width, layers, hidden)
        self.ln_final = _Affine(width)


class FrozenOpenCLIPEmbedder(nn.Module):
    """Same constructor surface as upstream (arch, version, device, max_length, freeze, layer); ``width``, ``heads``,
    ``layers``, ``vocab_size`` override the arch table (tests use a small tower).  The modules below only hold the
    parameters: ``forward`` runs the HIP entry points and raises without the library or a GPU."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", width=None, heads=None, layers=None, vocab_size=49408, mlp_ratio=4, bpe_path=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise NotImplementedError(f"layer={layer!r}: expected one of {self.LAYERS}")
        if arch not in ARCHS and None in (width, heads, layers):
            raise NotImplementedError(f"arch={arch!r}: unknown text tower (known: {sorted(ARCHS)}); give width/heads/layers")
        w0, h0, l0 = ARCHS.get(arch, (width, heads, layers))
        self.width, self.heads, self.layers = width or w0, heads or h0, layers or l0
        if self.width != 64 * self.heads:
            raise NotImplementedError("text tower: head dim must be 64 (width == 64 * heads), as in every OpenCLIP ViT tower")
        self.arch, self.version, self.max_length, self.layer = arch, version, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        self.vocab_size = vocab_size
        # parameters are torch.empty: the values come from a checkpoint (cond_stage_model.model.*) or reset_parameters()
        self.model = _TextTower(vocab_size, max_length, self.width, self.layers, mlp_ratio * self.width)
        self.weights_loaded = False
        self._pack = None
        # strings are accepted only with the BPE vocabulary file (not available offline); txt2img falls back otherwise
        bpe_path = bpe_path or os.environ.get("SDMI_BPE_VOCAB")
        self.tokenizer = SimpleTokenizer(bpe_path, vocab_size - 256 - 256 - 2) if bpe_path else None
        self.encodes_text = self.tokenizer is not None
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def reset_parameters(self, seed=0):
        """open_clip's text-tower initialisation scales (random weights: tests and smoke runs)."""
        g = torch.Generator().manual_seed(seed)
        w, n = self.width, self.layers
        proj_std, attn_std, fc_std = (w ** -0.5) * ((2 * n) ** -0.5), w ** -0.5, (2 * w) ** -0.5
        self.model.token_embedding.weight.normal_(0, 0.02, generator=g)
        self.model.positional_embedding.normal_(0, 0.01, generator=g)
        for blk in self.model.transformer.resblocks:
            blk.attn.in_proj_weight.normal_(0, attn_std, generator=g)
            blk.attn.in_proj_bias.normal_(0, 0.02, generator=g)
            blk.attn.out_proj.weight.normal_(0, proj_std, generator=g)
            blk.attn.out_proj.bias.normal_(0, 0.02, generator=g)
            blk.mlp.c_fc.weight.normal_(0, fc_std, generator=g)
            blk.mlp.c_fc.bias.normal_(0, 0.02, generator=g)
            blk.mlp.c_proj.weight.normal_(0, proj_std, generator=g)
            blk.mlp.c_proj.bias.normal_(0, 0.02, generator=g)
            for ln in (blk.ln_1, blk.ln_2):
                ln.weight.normal_(1.0, 0.05, generator=g)
                ln.bias.normal_(0, 0.05, generator=g)
        self.model.ln_final.weight.normal_(1.0, 0.05, generator=g)
        self.model.ln_final.bias.normal_(0, 0.05, generator=g)
        self.weights_loaded = True
        self._pack = None
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts a full open_clip model's entries: the vision tower, text_projection, logit_scale and the attn_mask
        buffer are not part of this stage and are ignored."""
        keep = {k: v for k, v in state_dict.items()
                if k.startswith("model.") and not k.startswith(("model.visual.", "model.text_projection", "model.logit_scale",
                                                                "model.attn_mask"))}
        r = super().load_state_dict(keep, strict=strict, **kw)
        if not r.missing_keys:
            self.weights_loaded = True
        self._pack = None
        return r

    def _apply(self, fn, *a, **kw):
        self._pack = None
        if not self.weights_loaded:
            return self         # uninitialised storage: moving or casting it would only touch 1.4 GB for nothing
        return super()._apply(fn, *a, **kw)

    def _build_pack(self, dev):
        f32 = dict(dtype=torch.float32, device=dev)
        m = self.model
        blocks = []
        for blk in m.transformer.resblocks[:self.layers - self.layer_idx]:
            blocks.append(dict(
                ln1=(blk.ln_1.weight.to(**f32), blk.ln_1.bias.to(**f32)),
                qkv=ops.pack_linear(blk.attn.in_proj_weight.to(**f32), blk.attn.in_proj_bias.to(**f32), dtype=torch.float32),
                out=ops.pack_linear(blk.attn.out_proj.weight.to(**f32), blk.attn.out_proj.bias.to(**f32), dtype=torch.float32),
                ln2=(blk.ln_2.weight.to(**f32), blk.ln_2.bias.to(**f32)),
                fc=ops.pack_linear(blk.mlp.c_fc.weight.to(**f32), blk.mlp.c_fc.bias.to(**f32), dtype=torch.float32),
                proj=ops.pack_linear(blk.mlp.c_proj.weight.to(**f32), blk.mlp.c_proj.bias.to(**f32), dtype=torch.float32)))
        self._pack = dict(dev=dev, blocks=blocks,
                          tok=m.token_embedding.weight.to(**f32).contiguous(),
                          pos=m.positional_embedding.to(**f32).contiguous(),
                          lnf=(m.ln_final.weight.to(**f32), m.ln_final.bias.to(**f32)))

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        if not self.weights_loaded:
            raise RuntimeError("FrozenOpenCLIPEmbedder: the text tower has no weights (load a checkpoint holding "
                               "cond_stage_model.model.*, or call reset_parameters() for random ones)")
        if not (torch.is_tensor(tokens) and tokens.dtype == torch.int64 and tokens.dim() == 2
                and tokens.shape[1] == self.max_length):
            raise ValueError(f"tokens: expected int64 [B, {self.max_length}] token ids")
        if not tokens.is_cuda:
            raise RuntimeError("FrozenOpenCLIPEmbedder runs on the HIP path only: tokens must be on the GPU")
        if int(tokens.min()) < 0 or int(tokens.max()) >= self.vocab_size:
            raise ValueError(f"tokens: ids outside [0, {self.vocab_size})")
        if self._pack is None or self._pack["dev"] != tokens.device:
            self._build_pack(tokens.device)
        pk = self._pack
        b, t = tokens.shape
        c, heads = self.width, self.heads
        x = ops.embed_tokens(tokens.contiguous(), pk["tok"], pk["pos"]).view(b * t, c)
        for blk in pk["blocks"]:
            h = ops.layernorm(x, *blk["ln1"])
            qkv = ops.linear(h, blk["qkv"])
            a = ops.attention_d64(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], heads, b, t, t, 0.125, causal=True)
            x = ops.linear(a, blk["out"], res=x)
            h = ops.layernorm(x, *blk["ln2"])
            f = ops.linear(h, blk["fc"], act=2)
            x = ops.linear(f, blk["proj"], res=x)
        return ops.layernorm(x, *pk["lnf"]).view(b, t, c)

    def forward(self, text):
        if torch.is_tensor(text):
            return self.encode_with_transformer(text)
        if self.tokenizer is None:
            raise RuntimeError("FrozenOpenCLIPEmbedder: no BPE vocabulary (bpe_path / $SDMI_BPE_VOCAB; the file ships with "
                               "open_clip, which is unavailable offline); pass int64 [B, 77] token ids "
                               "(encoders.wrap_token_ids) or ready [B, 77, C] embeddings")
        return self.encode_with_transformer(self.tokenizer(text, self.max_length).to(torch.device("cuda")))

    def encode(self, text):
        return self(text)
