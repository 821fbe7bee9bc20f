#!/usr/bin/env python3
"""bench.py - images/sec of the latent-diffusion hot path on MI355X (contract: see the task statement).

A "step" is one pass of the hot path over one batch: a full 50-step DDIM sampling with classifier-free
guidance (50 UNet forwards at batch 2B) + decode_first_stage + uint8 conversion of B images of 512x512
(BASELINE.json configs[2] per-GPU slice: SD2-base, 64x64 latent, batch 8 per GPU; with --batch 1 it is
configs[1]).  Inputs (weights, x_T, text embeddings) are resident in HBM before the timed region.
Weak scaling: every rank runs its own batch; the only cross-GPU traffic is the final uint8 image gather.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
import zlib

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SD2_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, legacy=False,
                use_checkpoint=True)      # + use_fp16 / use_bf16 from --dtype (v2-inference.yaml:24 ships use_fp16: True)
SD2_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
# algorithmic FLOPs (2 per MAC), SURVEY.md §8(d) / BASELINE.md §4
UNET_GFLOP = {32: 181.1, 64: 804.3, 96: 2149.1}
DEC_GFLOP = {32: 622.2, 64: 2514.5, 96: 5754.3}
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 / fp16 (same rate), MI355X_MICROARCH.md
HBM_PEAK_TBS = 8.0            # HBM3E spec (6.3 TB/s measured achievable), MI355X_MICROARCH.md


def build_model(dev, parameterization="eps", dtype="fp16"):
    from stablediffusion_amd import instantiate_from_config
    cfg = {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.0120, "timesteps": 1000, "conditioning_key": "crossattn",
        "scale_factor": 0.18215, "use_ema": False, "parameterization": parameterization, "image_size": 64, "channels": 4,
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                        "params": dict(SD2_UNET, use_fp16=dtype == "fp16", use_bf16=dtype == "bf16")},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL", "params": {
            "embed_dim": 4, "ddconfig": dict(SD2_DDCONFIG), "lossconfig": {"target": "torch.nn.Identity"}}},
        "cond_stage_config": "__is_unconditional__", "force_null_conditioning": True}}
    with torch.device("meta"):
        model = instantiate_from_config(cfg)
    model.model.diffusion_model.to_empty(device=dev)
    model.first_stage_model.to_empty(device=dev)
    # random-init weights of the SD2 architecture, generated on the device (a function of the parameter name);
    # zero-initialised tensors get values too
    with torch.no_grad():
        for name, p in list(model.model.named_parameters()) + list(model.first_stage_model.named_parameters()):
            g = torch.Generator(device=dev).manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if name.endswith("bias"):
                p.normal_(0, 0.05, generator=g)
            elif p.dim() == 1:
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0, (p[0].numel()) ** -0.5, generator=g)
    model.register_schedule(None, "linear", 1000, 0.00085, 0.0120)
    for name, buf in list(model.named_buffers(recurse=False)):
        setattr(model, name, buf.to(dev))
    return model.eval()


def cpu_baseline(model, h, ddim_steps, scale):
    """The oracle (CPU fp32 restatement, oracle/) timed on this box's host cores, as BASELINE.md §3 plans:
    (i) BASELINE.json configs[0] run IN FULL - 256^2 (32x32 latent), batch 1, 10 DDIM steps with CFG (20 UNet-sample
    forwards) + 1 decode - reported as config0_*; (ii) one CFG UNet step (batch 2) at the benchmark latent size + one
    1-image decode at the benchmark image size, scaled to the metric's unit as 1 / (ddim_steps * t_step + t_decode)
    = `value` (the same workload per image as the GPU line's)."""
    from oracle import model as OM
    from oracle import sampler as OS
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    usd = {k: v.detach().float().cpu() for k, v in model.model.diffusion_model.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in model.first_stage_model.state_dict().items()}
    cfg = {k: v for k, v in SD2_UNET.items()}
    acp = model.alphas_cumprod.cpu().numpy()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        OM.unet_forward(usd, cfg, torch.randn(1, 4, 16, 16, generator=g), torch.tensor([501]), torch.randn(1, 77, 1024, generator=g))  # page in
        # (i) configs[0] in full
        x_T = torch.randn(1, 4, 32, 32, generator=g)
        c, uc = torch.randn(1, 77, 1024, generator=g), torch.randn(1, 77, 1024, generator=g)
        t0 = time.perf_counter()
        fn = lambda x, t, cond: OS.apply_model(usd, cfg, x, t, cond)
        lat = OS.ddim_sample(fn, acp, 10, x_T, c, uc, scale, 0.0, getattr(model, "parameterization", "eps"))
        t_loop = time.perf_counter() - t0
        t0 = time.perf_counter()
        OS.decode_first_stage(vsd, SD2_DDCONFIG, lat, 0.18215)
        t_dec0 = time.perf_counter() - t0
        # (ii) one step + one decode at the benchmark size
        x = torch.randn(2, 4, h, h, generator=g)
        ctx = torch.randn(2, 77, 1024, generator=g)
        t = torch.full((2,), 501, dtype=torch.long)
        t0 = time.perf_counter()
        OM.unet_forward(usd, cfg, x, t, ctx)
        t_step = time.perf_counter() - t0
        t0 = time.perf_counter()
        OS.decode_first_stage(vsd, SD2_DDCONFIG, x[:1], 0.18215)
        t_dec = time.perf_counter() - t0
    return {"value": 1.0 / (ddim_steps * t_step + t_dec), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"fp32 CPU oracle: BASELINE configs[0] in full (256^2, batch 1, 10 DDIM steps + CFG + decode: "
                      f"{t_loop + t_dec0:.1f} s/image) + 1 CFG UNet step (batch 2, {h}x{h} latent) + 1 VAE decode "
                      f"({8 * h}^2); value = 1/({ddim_steps}*t_step + t_decode), the GPU line's per-image workload",
            "config0_s_per_image": t_loop + t_dec0, "config0_images_per_sec": 1.0 / (t_loop + t_dec0),
            "unet_step_s": t_step, "decode_s": t_dec}


FAMILY_NAMES = {"conv3x3": "sdmi_conv_halo_kernel + sdmi_gemm_kernel<1,...> (implicit-GEMM conv3x3, MFMA 16x16x32)",
                "linear": "sdmi_gemm_kernel<0,...> (linear / conv1x1, MFMA 16x16x32)",
                "attention": "attn_fwd_d64_kernel (flash attention d=64, MFMA 32x32x16)",
                "groupnorm": "gn_stats_kernel + gn_apply_kernel (GroupNorm(32)+SiLU)", "layernorm": "layernorm_kernel"}


def measured_peaks(dev):
    """The box's own peaks (SURVEY.md section 8(d) asks for fractions against nominal AND measured peaks):
    `mfma_tflops` = a register-only v_mfma_f32_16x16x32_f16 loop on pseudo-random operands on every CU (sdmi_probe_mfma:
    no memory traffic, so it is the matrix rate at the clock the part holds on random data - the nominal 2.5 PFLOP/s
    assumes 2.4 GHz), best of 4 launches of ~10 ms after a warm-up; `hbm_tbs` = a 16-byte-per-lane copy of 1 GiB
    (sdmi_probe_copy), read + write bytes / time, best of 4."""
    import ctypes as C
    from stablediffusion_amd import hip
    lib = hip.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.zeros(512, dtype=torch.float32, device=dev)
    flop = C.c_double(0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for i in range(5):
        e0.record()
        hip.check(lib.sdmi_probe_mfma(C.c_void_p(out.data_ptr()), 40000, C.byref(flop), stream), "sdmi_probe_mfma")
        e1.record()
        torch.cuda.synchronize()
        if i:
            best = max(best, flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    bw = 0.0
    for i in range(5):
        e0.record()
        hip.check(lib.sdmi_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, stream), "sdmi_probe_copy")
        e1.record()
        torch.cuda.synchronize()
        if i:
            bw = max(bw, 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    del src, dst
    return {"mfma_tflops": best, "hbm_tbs": bw,
            "note": "mfma: register-only 16x16x32 f16 MFMA loop on pseudo-random operands, 256 workgroups x 8 waves "
                    "(sdmi_probe_mfma); hbm: 1 GiB 16-byte-per-lane copy, read + write (sdmi_probe_copy); best of 4"}


def collect_trace(fn):
    """Run fn() with a HIP-event bracket around every traced launch -> [(kind, work, seconds, split_k, shape)]."""
    from stablediffusion_amd import ops
    ops.TRACE = []
    fn()
    torch.cuda.synchronize()
    tr, ops.TRACE = ops.TRACE, None
    return [(k, f, e0.elapsed_time(e1) * 1e-3, sk, shp) for k, f, e0, e1, sk, shp in tr]


def family_table(populations, peaks=None):
    """populations: [(repeat, trace list)] - per kernel family: algorithmic work (FLOPs for conv / linear / attention,
    bytes = one read + one write for GroupNorm / LayerNorm) / HIP-event launch durations."""
    fam = {}
    tot = {}
    for kind in FAMILY_NAMES:
        w = sum(rep * sum(f for k, f, _, _, _ in tr if k == kind) for rep, tr in populations)
        tm = sum(rep * sum(t for k, _, t, _, _ in tr if k == kind) for rep, tr in populations)
        n = sum(rep * sum(1 for k, _, _, _, _ in tr if k == kind) for rep, tr in populations)
        if n:
            tot[kind] = (w, tm, n)
    t_all = sum(v[1] for v in tot.values())
    for kind, (w, tm, n) in tot.items():
        mfma = kind in ("conv3x3", "linear", "attention")
        ach = w / tm / 1e12
        peak = MFMA_PEAK_TFLOPS if mfma else HBM_PEAK_TBS
        fam[kind] = {"kernel": FAMILY_NAMES[kind], "bound": "mfma" if mfma else "hbm", "achieved": ach, "peak": peak,
                     "unit": "TFLOP/s" if mfma else "TB/s", "frac": ach / peak, "work_per_launch": w / n,
                     "avg_launch_us": tm / n * 1e6, "launches_per_step": n, "time_share": tm / t_all}
        if peaks:
            pm = peaks["mfma_tflops"] if mfma else peaks["hbm_tbs"]
            fam[kind]["peak_measured"] = pm
            fam[kind]["frac_measured"] = ach / pm if pm else None
    # the same launches on the roofline *model* (per launch the larger of its MFMA and its HBM time, nominal peaks)
    for kind, e in roofline_model(populations, MFMA_PEAK_TFLOPS * 1e12, HBM_PEAK_TBS * 1e12).items():
        fam[kind]["roofline_model"] = e
    return fam, t_all


def launch_bytes(kind, work, shp):
    """Compulsory HBM bytes of one traced launch: every operand read once, the result written once (16-bit operands;
    what the launch would move if nothing were re-read).  With the FLOPs this places the launch on the roofline:
    attainable time = max(FLOPs / MFMA peak, bytes / HBM peak)."""
    if kind in ("groupnorm", "layernorm"):
        return work                                        # (their `work` already is one read + one write)
    if kind == "attention":
        nb, heads, nq, nk = shp[:4]
        return nb * heads * (2 * nq + 2 * nk) * 64 * 2     # q, o and k, v
    m, n_w, k, stride, upsample, geglu, res, out32, batch = tuple(shp) + (0, 0, 0, 1)[max(len(shp) - 5, 0):]
    n_out = n_w // 2 if geglu else n_w
    if kind == "conv3x3":
        pixels_in = m * stride * stride // (4 if upsample else 1)
        a_bytes = pixels_in * (k // 9) * 2
    else:
        a_bytes = m * k * 2
    return batch * (a_bytes + n_w * k * 2 + m * n_out * (4 if out32 else 2) + (m * n_out * 2 if res else 0))


def roofline_model(populations, peak_flops, peak_bytes):
    """Per family: sum over launches of max(FLOPs / peak_flops, bytes / peak_bytes) against the measured durations, and
    the share of the family's time spent in launches whose byte term is the larger one (HBM-bound by the model)."""
    out = {}
    for kind in FAMILY_NAMES:
        t_roof = t_meas = t_hbm = 0.0
        for rep, tr in populations:
            for k, f, t, _, shp in tr:
                if k != kind:
                    continue
                fl = f if kind in ("conv3x3", "linear", "attention") else 0.0
                tb = launch_bytes(kind, f, shp) / peak_bytes
                tf = fl / peak_flops
                t_roof += rep * max(tf, tb)
                t_meas += rep * t
                if tb > tf:
                    t_hbm += rep * t
        if t_meas:
            out[kind] = {"frac": t_roof / t_meas, "attainable_ms": t_roof * 1e3, "measured_ms": t_meas * 1e3,
                         "hbm_bound_time_share": t_hbm / t_meas}
    return out


def executed_flops(tr):
    return sum(f for k, f, _, _, _ in tr if k in ("conv3x3", "linear", "attention"))


def roofline(model, sampler, B, h, ddim_steps, c, uc, x_T, dtype, peaks, step_ms):
    """Per kernel family of ONE benchmark step (ddim_steps UNet forwards at batch 2B + one decode of B images):
    algorithmic work / launch durations from HIP events recorded on the launch stream around every launch.
    The top-level object is the family that takes the most GPU time; `families` carries all of them."""
    dev = x_T.device
    c_in = torch.cat([uc, c]).contiguous()
    ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
    x_in = torch.cat([x_T, x_T])
    unet = model.model.diffusion_model
    temb = unet.time_embedding_table(torch.tensor([501]))[0]

    # (cfg_pairs=True + the hoisted time path: the launch population of the timed DDIM loop, whose sampler passes the
    # same hints)
    call = lambda: model.apply_model(x_in, ts, c_in, cfg_pairs=True, time_emb=temb)
    collect_trace(call)       # warm
    un = collect_trace(call)
    de = collect_trace(lambda: model.decode_first_stage(x_T))
    fam, t_all = family_table([(ddim_steps, un), (1, de)], peaks)
    if os.environ.get("SDMI_DUMP_SHAPES"):      # per-shape table of one UNet forward + one decode (development aid)
        agg = {}
        for src, lst in (("unet", un), ("dec", de)):
            for k, f, t, sk, shp in lst:
                e = agg.setdefault((src, k) + tuple(shp) + (sk,), [0, 0.0, 0.0])
                e[0] += 1; e[1] += t; e[2] += f
        with open(os.environ["SDMI_DUMP_SHAPES"], "w") as fh:
            fh.write("src kind shape... splitk calls total_us T(FLOP|B)/s\n")
            for key, (n_, t_, f_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fh.write(" ".join(str(x) for x in key) + f" {n_} {t_ * 1e6:.1f} {f_ / t_ / 1e12:.2f}\n")
    dom = max(fam, key=lambda k: fam[k]["time_share"])
    # HBM-side bytes per launch of each family: PMC figures of separate rocprofv3 passes over this same command,
    # committed under profiles/ (counters cannot be read from inside the process); null when the file is absent
    traffic, tnote = None, "no profiles/r0?_traffic.json"
    for name in ("r03g_traffic.json", "r03e_traffic.json", "r03d_traffic.json", "r03c_traffic.json", "r03b_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            for kind in fam:
                e = tj.get("per_family", {}).get(kind)
                if e:
                    fam[kind]["traffic"] = e["bytes_per_launch"]
            traffic = fam[dom].get("traffic")
            tnote = f"profiles/{name}: " + tj.get("note", "")
            break
    top = dict(fam[dom])
    top.update({"traffic": traffic, "traffic_note": tnote, "dtype": dtype, "families": fam,
                "peaks_measured": peaks,
                "bracket_sum_ms_per_step": t_all * 1e3, "step_ms": step_ms,
                "note": "time-dominant kernel family of the step; durations are HIP-event brackets around each launch "
                        "(split-K reduce launches and the event-to-event gap included, so the brackets of a step add up "
                        "to MORE than the step takes - bracket_sum_ms_per_step vs step_ms: fractions are conservative, "
                        "time_share is relative to the bracket sum); frac = against the nominal 2.5 PFLOP/s dense 16-bit "
                        "peak / the 8 TB/s HBM spec, frac_measured = against peaks_measured (this box, this run); "
                        "roofline_model.frac = sum over the family's launches of max(FLOPs / 2.5 PFLOP/s, compulsory bytes / "
                        "8 TB/s) over their measured time - a launch of low intensity (K = 320 projections: 107 FLOP/B "
                        "against a machine balance of 312) is priced at its HBM time there, not at the MFMA peak"})
    return top, executed_flops(un), executed_flops(de)


def leg_768(model, dev, args, peaks):
    """BASELINE.json configs[3]: SD2.1-v 768^2 - v-prediction, 96x96 latent (9216-token self-attention), 50 DDIM steps
    + CFG, batch 8 on one GPU; one timed sampling + decode after one warm-up, and the per-family table of its launches."""
    from stablediffusion_amd.ddim import DDIMSampler
    from stablediffusion_amd import ops
    B, h = 8, 96
    prev = model.parameterization
    model.parameterization = "v"
    try:
        x_T = torch.randn(B, 4, h, h, generator=torch.Generator().manual_seed(77)).to(dev)
        c = torch.randn(B, 77, 1024, generator=torch.Generator().manual_seed(4321)).to(dev)
        uc = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(99)).expand(B, -1, -1).contiguous().to(dev)
        sampler = DDIMSampler(model)

        def one():
            smp, _ = sampler.sample(S=args.ddim_steps, batch_size=B, shape=[4, h, h], conditioning=c, verbose=False,
                                    unconditional_guidance_scale=args.scale, unconditional_conditioning=uc, eta=0.0, x_T=x_T)
            return ops.image_to_u8(model.decode_first_stage(smp))
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = one()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert img.shape == (B, 8 * h, 8 * h, 3)
        c_in = torch.cat([uc, c]).contiguous()
        ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
        x_in = torch.cat([x_T, x_T])
        temb = model.model.diffusion_model.time_embedding_table(torch.tensor([501]))[0]
        call = lambda: model.apply_model(x_in, ts, c_in, cfg_pairs=True, time_emb=temb)
        call()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(3):
            call()
        e1.record()
        model.decode_first_stage(x_T)
        e2.record()
        torch.cuda.synchronize()
        step_ms, dec_ms = e0.elapsed_time(e1) / 3, e1.elapsed_time(e2)
        un = collect_trace(call)
        de = collect_trace(lambda: model.decode_first_stage(x_T))
        fam, _ = family_table([(args.ddim_steps, un), (1, de)], peaks)
    finally:
        model.parameterization = prev
    alg = (2 * args.ddim_steps * UNET_GFLOP[h] + DEC_GFLOP[h]) * 1e9 * B
    exe = args.ddim_steps * executed_flops(un) + executed_flops(de)
    return {"workload": f"BASELINE.json configs[3]: SD2.1-v 768^2 (96x96 latent, 9216-token self-attention), v-prediction, "
                        f"batch {B}, {args.ddim_steps} DDIM steps + CFG (UNet batch {2 * B}), decode + uint8",
            "batch": B, "s_per_batch": dt, "images_per_sec": B / dt,
            "unet_step_ms_cfg_pairs": step_ms, "decode_ms": dec_ms,
            "unet_step_tflops_algorithmic": 2 * B * UNET_GFLOP[h] / step_ms,      # GFLOP / ms = TFLOP/s
            "unet_step_mfma_frac": 2 * B * UNET_GFLOP[h] / step_ms / MFMA_PEAK_TFLOPS,
            "tflops_algorithmic_unhoisted": alg / dt / 1e12, "tflops_executed": exe / dt / 1e12,
            "mfma_frac_end_to_end_executed": exe / dt / 1e12 / MFMA_PEAK_TFLOPS,
            "mfma_frac_measured_end_to_end_executed": exe / dt / 1e12 / peaks["mfma_tflops"] if peaks else None,
            "families": fam}


def leg_decode_768(model, dev, peaks):
    """BASELINE.json configs[4]: AutoencoderKL decode_first_stage standalone at 768^2, batch 16 (92.1 TFLOP)."""
    B, h = 16, 96
    z = torch.randn(B, 4, h, h, generator=torch.Generator().manual_seed(5)).to(dev)
    model.decode_first_stage(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        img = model.decode_first_stage(z)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    assert img.shape == (B, 3, 8 * h, 8 * h)
    de = collect_trace(lambda: model.decode_first_stage(z))
    fam, _ = family_table([(1, de)], peaks)
    tf = B * DEC_GFLOP[h] / ms
    return {"workload": "BASELINE.json configs[4]: AutoencoderKL decode_first_stage standalone, 768^2, batch 16",
            "batch": B, "decode_ms": ms, "images_per_sec": B / (ms * 1e-3), "tflops_algorithmic": tf,
            "mfma_frac": tf / MFMA_PEAK_TFLOPS, "mfma_frac_measured": tf / peaks["mfma_tflops"] if peaks else None,
            "tflops_executed": executed_flops(de) / (ms * 1e-3) / 1e12, "families": fam}


def secondary_dtype(model, dt, x_in, ts, c_in, x_T, ddim_steps, B):
    """The other 16-bit operand policy on the same weights (informational line beside the headline): UNet step and
    decode times, and the images/sec they imply."""
    unet, vae = model.model.diffusion_model, model.first_stage_model
    prev = unet.compute_dtype
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    unet.dtype = unet.compute_dtype = tdt
    unet.invalidate_packed()
    vae.set_compute_dtype(tdt)
    try:
        model.apply_model(x_in, ts, c_in)
        model.decode_first_stage(x_T)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(5):
            model.apply_model(x_in, ts, c_in)
        e1.record()
        model.decode_first_stage(x_T)
        e2.record()
        torch.cuda.synchronize()
        step, dec = e0.elapsed_time(e1) / 5, e1.elapsed_time(e2)
    finally:
        unet.dtype = unet.compute_dtype = prev
        unet.invalidate_packed()
        vae.set_compute_dtype(prev)
    return {"dtype": dt, "unet_step_ms": step, "decode_ms": dec,
            "images_per_sec_from_step_and_decode": B / ((ddim_steps * step + dec) * 1e-3),
            "note": "same MFMA rate as the headline dtype; parity of this policy: tests/test_model_gpu.py docstring"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (UNet batch is 2x with CFG)")
    ap.add_argument("--latent", type=int, default=64, help="latent side: 64 = 512^2, 96 = 768^2")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--parameterization", default="eps", choices=["eps", "v"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the UNet step from a captured hipGraph (auto: when batch <= 2, where the step is launch-bound)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the BASELINE configs[1] (batch 1) latency leg")
    ap.add_argument("--dtype", choices=["fp16", "bf16"], default="fp16",
                    help="MFMA operand / storage type (the reference's use_fp16 / use_bf16 policy).  fp16 is the headline: "
                         "it is what v2-inference.yaml ships and it meets the 1e-2 parity gate; bf16 runs at the same rate")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary-dtype (bf16) UNet-step / decode timing")
    ap.add_argument("--no-legs", action="store_true", help="skip the BASELINE configs[3] (768^2 v-prediction) and configs[4] "
                                                          "(standalone 768^2 batch-16 decode) legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from stablediffusion_amd.ddim import DDIMSampler
    from stablediffusion_amd import ops
    model = build_model(dev, args.parameterization, args.dtype)
    B, h = args.batch, args.latent
    use_graph = args.hip_graph == "on" or (args.hip_graph == "auto" and B <= 2)
    sampler = DDIMSampler(model, hip_graph=use_graph)
    # synthetic inputs of SURVEY.md §8(d): per-rank latents, shared "prompt" embeddings
    from stablediffusion_amd import dp
    x_T = dp.shard_latents(world * B, rank, world, (4, h, h), base_seed=1234).to(dev)   # seeds follow the GLOBAL index
    c = torch.randn(B, 77, 1024, generator=torch.Generator().manual_seed(4321)).to(dev)
    uc = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(99)).expand(B, -1, -1).contiguous().to(dev)

    def one_step():
        samples, _ = sampler.sample(S=args.ddim_steps, batch_size=B, shape=[4, h, h], conditioning=c, verbose=False,
                                    unconditional_guidance_scale=args.scale, unconditional_conditioning=uc, eta=0.0,
                                    x_T=x_T)
        img = model.decode_first_stage(samples)
        u8 = ops.image_to_u8(img)
        # the only cross-GPU traffic: final image gather to rank 0 (RCCL); a no-op at world == 1
        allimg = dp.gather_images(u8, world * B, rank, world)
        return u8 if allimg is None else allimg[:B]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out.shape == (B, 8 * h, 8 * h, 3) and torch.isfinite(out.float()).all()

    images = world * B * args.steps
    value = images / dt
    flop_per_image = (2 * args.ddim_steps * UNET_GFLOP.get(h, 0) + DEC_GFLOP.get(h, 0)) * 1e9
    result = {
        "metric": "images/sec (node) SD2-base 512^2 50-step DDIM; UNet step ms", "value": value, "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SD2 v2-inference UNet+VAE random-init, {8 * h}^2 ({h}x{h} latent), batch {B}/GPU, "
                               f"{args.ddim_steps} DDIM steps + CFG (UNet batch {2 * B}), decode + uint8; "
                               f"BASELINE.json configs[{1 if B == 1 else 2}]",
                   "global_batch": world * B, "latent": h, "ddim_steps": args.ddim_steps, "parallelism": f"dp{world}",
                   "parameterization": args.parameterization, "hip_graph": use_graph, "cfg_pairs": True,
                   "time_path_hoisted": True},
        "unet_step_ms": None,
        # work / time of the timed region.  `algorithmic_unhoisted`: the reference's own FLOP count (SURVEY.md section 8(d):
        # 2 x ddim_steps x 804.3 + 2514.5 GFLOP per 512^2 image) over the measured time - an equivalent-work rate, NOT what
        # the matrix pipes executed, because three hoists remove work from the step loop; `effective_tflops_per_gpu` /
        # `mfma_frac_end_to_end` (filled in below from the launch trace) count only the launches that ran.
        "algorithmic_tflops_per_gpu_unhoisted": value / world * flop_per_image / 1e12 if flop_per_image else None,
        "effective_tflops_per_gpu": None, "mfma_frac_end_to_end": None,
    }
    # UNet step ms (one CFG-batched forward), HIP events on the launch stream
    c_in = torch.cat([uc, c]).contiguous()
    ts = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
    x_in = torch.cat([x_T, x_T])
    model.apply_model(x_in, ts, c_in)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        model.apply_model(x_in, ts, c_in)
    e1.record()
    torch.cuda.synchronize()
    result["unet_step_ms"] = e0.elapsed_time(e1) / 5
    # the same CFG batch with the samplers' hints: cfg_pairs (rows b and b + B are the same latent and step: the layers in
    # front of the first cross-attention run once per pair) and the hoisted time path (a row of the per-schedule
    # table instead of four launches) - what the timed DDIM loop above calls
    temb = model.model.diffusion_model.time_embedding_table(torch.tensor([501]))[0]
    model.apply_model(x_in, ts, c_in, cfg_pairs=True, time_emb=temb)
    e0.record()
    for _ in range(5):
        model.apply_model(x_in, ts, c_in, cfg_pairs=True, time_emb=temb)
    e1.record()
    torch.cuda.synchronize()
    result["unet_step_ms_cfg_pairs"] = e0.elapsed_time(e1) / 5
    e0.record()
    model.decode_first_stage(x_T)
    e1.record()
    torch.cuda.synchronize()
    result["decode_ms"] = e0.elapsed_time(e1)

    # BASELINE configs[1]: batch 1 (UNet batch 2), 50 DDIM steps + CFG, one GPU - the latency case.  The step is
    # launch-bound there, so it is replayed from a hipGraph; eager numbers are printed beside it.
    if world == 1 and B != 1 and not args.no_batch1:
        from stablediffusion_amd.graph import GraphedApplyModel
        x1, c1, uc1 = x_T[:1].contiguous(), c[:1].contiguous(), uc[:1].contiguous()
        lat = DDIMSampler(model, hip_graph=True)

        def one_image():
            smp, _ = lat.sample(S=args.ddim_steps, batch_size=1, shape=[4, h, h], conditioning=c1, verbose=False,
                                unconditional_guidance_scale=args.scale, unconditional_conditioning=uc1, eta=0.0, x_T=x1)
            return ops.image_to_u8(model.decode_first_stage(smp))
        one_image()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_image()
        torch.cuda.synchronize()
        t_img = time.perf_counter() - t1
        cin1 = torch.cat([uc1, c1]).contiguous()
        xin1, ts1 = torch.cat([x1, x1]), torch.full((2,), 501, device=dev, dtype=torch.long)
        temb1 = model.model.diffusion_model.time_embedding_table(torch.tensor([501]))[0]
        hints = dict(cfg_pairs=True, time_emb=temb1)        # what the samplers pass (guidance pairs, hoisted time path)
        steps_ms = {}
        for name, fn, kw in (("graph", GraphedApplyModel(model), hints), ("eager", model.apply_model, hints),
                             ("graph_plain", GraphedApplyModel(model), {}), ("eager_plain", model.apply_model, {})):
            fn(xin1, ts1, cin1, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                fn(xin1, ts1, cin1, **kw)
            torch.cuda.synchronize()
            steps_ms[name] = (time.perf_counter() - t1) / 10 * 1e3
        b1_un = collect_trace(lambda: model.apply_model(xin1, ts1, cin1, **hints))
        b1_de = collect_trace(lambda: model.decode_first_stage(x1))
        b1_fam, b1_sum = family_table([(args.ddim_steps, b1_un), (1, b1_de)])
        b1_dom = max(b1_fam, key=lambda k: b1_fam[k]["time_share"])
        b1_roof = dict(b1_fam[b1_dom])
        b1_roof.update({"families": b1_fam, "bracket_sum_ms_per_image": b1_sum * 1e3,
                        "launches_per_unet_step": len(b1_un),
                        "note": "as the top-level roofline object, for the batch-1 launch population (HIP-event brackets: "
                                "at this batch most launches are shorter than the bracket's own event-to-event gap, so "
                                "the fractions are lower bounds)"})
        result["batch1"] = {"workload": "BASELINE.json configs[1]: batch 1, UNet batch 2 (CFG), hipGraph replay of the UNet step",
                            "s_per_image": t_img, "images_per_sec": 1.0 / t_img,
                            "unet_step_ms_graph": steps_ms["graph"], "unet_step_ms_eager": steps_ms["eager"],
                            "unet_step_ms_graph_no_hints": steps_ms["graph_plain"],
                            "unet_step_ms_eager_no_hints": steps_ms["eager_plain"],
                            "unet_step_tflops_algorithmic": 2 * UNET_GFLOP.get(h, 0) / steps_ms["graph"],
                            "unet_step_mfma_frac": 2 * UNET_GFLOP.get(h, 0) / steps_ms["graph"] / MFMA_PEAK_TFLOPS,
                            "roofline": b1_roof,
                            "note": "unet_step_ms_graph / _eager: with the samplers' hints (cfg_pairs, hoisted time path) - what "
                                    "s_per_image runs; *_no_hints: plain apply_model(x, t, c)"}

    if rank == 0:
        peaks = measured_peaks(dev)
        result["peaks_measured"] = peaks
        if not args.no_roofline:
            result["roofline"], ex_unet, ex_dec = roofline(model, sampler, B, h, args.ddim_steps, c, uc, x_T, args.dtype, peaks,
                                                           dt / args.steps * 1e3)
            exe = args.ddim_steps * ex_unet + ex_dec            # FLOPs of the launches of ONE bench step on this rank
            alg = flop_per_image * B
            result["effective_tflops_per_gpu"] = exe / (dt / args.steps) / 1e12
            result["mfma_frac_end_to_end"] = exe / (dt / args.steps) / 1e12 / MFMA_PEAK_TFLOPS
            result["mfma_frac_measured_end_to_end"] = exe / (dt / args.steps) / 1e12 / peaks["mfma_tflops"]
            result["hoists"] = {
                "executed_gflop_per_step": exe / 1e9, "algorithmic_gflop_per_step_unhoisted": alg / 1e9,
                "removed_gflop_per_step": (alg - exe) / 1e9,
                "what": "work the step loop does not execute: (i) cross-attention K/V = to_k/to_v(context) of all 16 blocks, "
                        "computed once per prompt (attention.py:312-313); (ii) the timestep path (sinusoid, time MLP, 22 "
                        "embedding projections) computed once per schedule for all steps (openai_model.py:956-964, 280); "
                        "(iii) the guidance-pair prefix: conv_in, the first ResBlock, GroupNorm + proj_in and the first "
                        "self-attention run once per (uncond, cond) pair (UNetModel._forward_cfg_pairs)"}
        if world == 1 and h == 64 and not args.no_legs:
            result["config3_768_v"] = leg_768(model, dev, args, peaks)
            result["config4_decode_768_b16"] = leg_decode_768(model, dev, peaks)
        if world == 1 and not args.no_secondary:
            result["secondary_dtype"] = secondary_dtype(model, "bf16" if args.dtype == "fp16" else "fp16", x_in, ts, c_in, x_T,
                                                        args.ddim_steps, B)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model, h, args.ddim_steps, args.scale)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
