"""HIP-graph replay of the UNet step.

One UNet forward is ~700 kernel launches; at small batch (BASELINE config 1: batch 1, UNet batch 2 with CFG) the
kernels are a few microseconds each and the host launch rate, not the GPU, sets the step time.  The launch sequence
is static for fixed shapes, so it is captured once into a hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm;
the C-ABI calls launch on torch's current stream and are captured like any other launch) and replayed per step:
the only per-step host work is two small copies into the static input buffers and one graph launch.
"""
import torch


class GraphedApplyModel:
    """Callable with the signature of ``LatentDiffusion.apply_model(x, t, cond)`` for a fixed cond tensor."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self._graph = None
        self._x = self._t = self._out = self._cond = self._held = self._temb = None

    def _capture(self, x, t, cond, cfg_pairs=False, time_emb=None):
        self._x, self._t, self._cond = x.clone(), t.clone(), cond
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
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        self._held = (getattr(unet, "_pack", None), getattr(unet, "_ctx_cache", None))
        self._gen = getattr(unet, "_generation", 0)

    def __call__(self, x, t, cond, cfg_pairs=False, time_emb=None):
        if isinstance(cond, (dict, list, tuple)):
            return self.model.apply_model(x, t, cond)            # only the plain cross-attention form is captured
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        key = (tuple(x.shape), x.dtype, tuple(t.shape), cond.data_ptr(), tuple(cond.shape), cond._version,
               getattr(unet, "_generation", 0), bool(cfg_pairs),       # new weights (load_state_dict / .to()) -> re-capture
               None if time_emb is None else tuple(time_emb.shape))
        if key != self._key:
            self._capture(x, t, cond, cfg_pairs, time_emb)
            self._key = key
        self._x.copy_(x)
        self._t.copy_(t)
        if time_emb is not None:
            self._temb.copy_(time_emb)
        self._graph.replay()
        return self._out
