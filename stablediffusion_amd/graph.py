"""HIP-graph replay of the UNet step.

One UNet forward is ~290 kernel launches; at small batch (BASELINE config 1: batch 1, UNet batch 2 with CFG) the
kernels are 5-40 microseconds each and an eager step sits close to the host's launch rate.  The launch sequence
is static for fixed shapes, so it is captured once into a hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm;
the C-ABI calls launch on torch's current stream and are captured like any other launch) and replayed per step:
the only per-step host work is two small copies into the static input buffers and one graph launch.

The conditioning is a static input as well (round 3): a capture costs three forwards, and txt2img has a new prompt per
image, so keying the capture on the prompt tensor made every image pay it (26 of 291 ms per image at batch 1).  A new
prompt is copied into the captured context buffer and the UNet's cross-attention K / V - the only thing the captured
launches read of it - are re-projected into the tensor the graph already points at (UNetModel.refresh_context_kv).
"""
import torch


class GraphedApplyModel:
    """Callable with the signature of ``LatentDiffusion.apply_model(x, t, cond)`` for a fixed cond tensor."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self._graph = None
        self._x = self._t = self._out = self._cond = self._held = self._temb = None
        self._kv = self._cond_src = self._cond_ref = None

    def _capture(self, x, t, cond, cfg_pairs=False, time_emb=None):
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        static_cond = hasattr(unet, "refresh_context_kv")
        # the context is a static buffer of the capture where the UNet can refresh its K / V in place
        self._x, self._t, self._cond = x.clone(), t.clone(), (cond.clone() if static_cond else cond)
        kw = {"cfg_pairs": True} if cfg_pairs else {}
        self._temb = None
        if time_emb is not None:          # the per-step row of the samplers' time-embedding table: a static input
            self._temb = time_emb.clone()
            kw["time_emb"] = self._temb
        # warm-up on a side stream: packs weights, fills the cross-attention K/V cache, sets kernel attributes -
        # everything that must not happen inside a capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.model.apply_model(self._x, self._t, self._cond, **kw)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # the warm-up stream's scratch arena (split-K slabs ...) would otherwise stay pinned under a stream handle nobody
        # launches on again - one arena per re-capture
        from . import ops
        ops.release_scratch(stream=s.cuda_stream)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self.model.apply_model(self._x, self._t, self._cond, **kw)
        # the captured launches hold raw pointers into the packed weights and the cached cross-attention K/V of the
        # UNet: keep both alive for as long as this graph is, whatever later eager calls do to the model's caches
        self._held = (getattr(unet, "_pack", None), getattr(unet, "_ctx_cache", None))
        self._gen = getattr(unet, "_generation", 0)
        cache = getattr(unet, "_ctx_cache", None)
        self._kv = cache[1] if static_cond and cache is not None and cache[2] is self._cond else None
        # (the source tensor is kept alive: a freed prompt's address could come back with another prompt's bytes)
        self._cond_src, self._cond_ref = (cond.data_ptr(), cond._version), cond

    def __call__(self, x, t, cond, cfg_pairs=False, time_emb=None):
        if isinstance(cond, (dict, list, tuple)):
            return self.model.apply_model(x, t, cond)            # only the plain cross-attention form is captured
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        src = (cond.data_ptr(), cond._version)
        key = (tuple(x.shape), x.dtype, tuple(t.shape), tuple(cond.shape), cond.dtype,
               getattr(unet, "_generation", 0), bool(cfg_pairs),       # new weights (load_state_dict / .to()) -> re-capture
               None if time_emb is None else tuple(time_emb.shape))
        if key != self._key or (self._kv is None and src != self._cond_src):
            self._capture(x, t, cond, cfg_pairs, time_emb)
            self._key = key
        elif src != self._cond_src:
            # a new prompt: into the captured context buffer, its K / V into the tensor the captured launches read
            self._cond.copy_(cond)
            unet.refresh_context_kv(self._cond, self._kv)
            self._cond_src, self._cond_ref = src, cond
        self._x.copy_(x)
        self._t.copy_(t)
        if time_emb is not None:
            self._temb.copy_(time_emb)
        self._graph.replay()
        return self._out
