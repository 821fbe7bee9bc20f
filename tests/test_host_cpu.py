"""CPU-side tests (no GPU): the C ABI loads and exports every declared symbol, the ctypes mirror matches
the C struct, the host logic (factory, schedules, samplers' tables, weight packing, state-dict names,
data-parallel sharding) behaves like the reference's, and the product path fails loudly without a GPU."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def lib():
    from stablediffusion_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return hip.lib()


def test_abi_exports_every_declared_symbol(lib):
    from stablediffusion_amd import hip
    header = open(os.path.join(ROOT, "include", "sdmi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sdmi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 16
    raw = ctypes.CDLL(hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/sdmi.h but not exported"
    assert declared == set(hip.SIGNATURES), "ctypes binding and header disagree"
    assert lib.sdmi_abi_version() == hip.ABI_VERSION >= 4
    assert b"workspace" in lib.sdmi_status_str(-3) and b"no-allocation" in lib.sdmi_status_str(-5)
    assert lib.sdmi_gemm_workspace_bytes(128, 1280, 4) == 4 * 128 * 1280 * 4
    assert lib.sdmi_gemm_workspace_bytes(128, 1280, 1) == 0
    assert lib.sdmi_groupnorm_workspace_bytes(2, 4096) > 0


@pytest.mark.parametrize("cname,pyname", [("sdmi_gemm_args", "GemmArgs"), ("sdmi_xattn_args", "XattnArgs")])
def test_gemm_args_struct_matches_header(cname, pyname):
    from stablediffusion_amd import hip
    cls = getattr(hip, pyname)
    fields = [f[0] for f in cls._fields_]
    prog = f'#include "sdmi.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){{printf("%zu", sizeof({cname}));' + \
        "".join(f'printf(" %zu", offsetof({cname}, {f}));' for f in fields) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert int(out[0]) == ctypes.sizeof(cls)
    for f, off in zip(fields, out[1:]):
        assert getattr(cls, f).offset == int(off), f


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad arguments before touching the device (no compute call is made here)."""
    from stablediffusion_amd import hip
    a = hip.GemmArgs()
    assert lib.sdmi_gemm(ctypes.byref(a), None) == -1          # null pointers
    a.A, a.W, a.out = 16, 16, 16
    a.M, a.N, a.K = 8, 8, 100
    assert lib.sdmi_gemm(ctypes.byref(a), None) == -2          # K not a multiple of 64
    assert lib.sdmi_layernorm(None, None, None, None, 1, 8, 1e-5, 0, None) == -1
    assert lib.sdmi_attention_d64(16, 16, 16, 16, 1, 1, 8, 8, 60, 64, 64, 64, 0.125, 0, None) == -2
    assert lib.sdmi_attention_d64(16, 16, 16, 16, 1, 1, 8, 8, 64, 64, 64, 64, 0.125, 7, None) == -1     # unknown dtype
    assert lib.sdmi_groupnorm(16, None, 30, 0, 16, 16, 16, 1, 8, 1e-5, 1, 0, 16, 1 << 20, None) == -2
    xa = hip.XattnArgs()
    assert lib.sdmi_xattn_block(ctypes.byref(xa), None) == -1 and lib.sdmi_xattn_block_applies(ctypes.byref(xa)) == 0
    xa.x = xa.out = xa.wq = xa.wo = xa.k = xa.v = 256
    xa.ln_gamma = xa.ln_beta = 256
    xa.ldx = xa.ld_out = 320
    xa.ldk = xa.ldv = 640
    xa.M, xa.rows_per_image, xa.C, xa.heads, xa.nk, xa.dtype, xa.ln_eps = 256, 128, 320, 5, 77, hip.DTYPE_F16, 1e-5
    assert lib.sdmi_xattn_block_applies(ctypes.byref(xa)) == 1
    for field, bad in (("nk", 129), ("M", 192), ("rows_per_image", 96), ("C", 640), ("dtype", hip.DTYPE_BF16), ("ldk", 324)):
        keep = getattr(xa, field)
        setattr(xa, field, bad)
        assert lib.sdmi_xattn_block_applies(ctypes.byref(xa)) == 0, field
        assert lib.sdmi_xattn_block(ctypes.byref(xa), None) == -1, field       # refused before any launch
        setattr(xa, field, keep)


def test_factory_surface():
    from stablediffusion_amd import instantiate_from_config, get_obj_from_str
    with pytest.raises(KeyError, match="Expected key `target` to instantiate."):
        instantiate_from_config({"params": {}})
    assert instantiate_from_config("__is_first_stage__") is None
    assert instantiate_from_config("__is_unconditional__") is None
    assert isinstance(instantiate_from_config({"target": "torch.nn.Identity"}), torch.nn.Identity)
    lin = instantiate_from_config({"target": "torch.nn.Linear", "params": {"in_features": 3, "out_features": 2}})
    assert lin.weight.shape == (2, 3)
    from stablediffusion_amd.unet import UNetModel
    assert get_obj_from_str("ldm.modules.diffusionmodules.openaimodel.UNetModel") is UNetModel
    assert get_obj_from_str("ldm.modules.diffusion.openai_model.UNetModel") is UNetModel


def test_schedule_tables_match_reference_goldens():
    from stablediffusion_amd import schedule as S
    z = np.load(os.path.join(GOLD, "schedule_tables.npz"))
    betas = S.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120)
    np.testing.assert_allclose(betas, z["betas_linear"], rtol=1e-12)
    np.testing.assert_allclose(S.make_beta_schedule("cosine", 1000), z["betas_cosine"], rtol=1e-10, atol=1e-15)
    with pytest.raises(ValueError):
        S.make_beta_schedule("nope", 10)
    acp = np.cumprod(1 - betas).astype(np.float32)
    for s in (10, 30, 50):
        ts = S.make_ddim_timesteps("uniform", s, 1000, verbose=False)
        np.testing.assert_array_equal(ts, z[f"ddim_timesteps_{s}"])
        for eta in (0.0, 0.5):
            sig, a, ap = S.make_ddim_sampling_parameters(acp, ts, eta, verbose=False)
            np.testing.assert_allclose(a, z[f"alphas_{s}_eta{eta}"], rtol=1e-6)
            np.testing.assert_allclose(ap, z[f"alphas_prev_{s}_eta{eta}"], rtol=1e-6)
            np.testing.assert_allclose(sig, z[f"sigmas_{s}_eta{eta}"], rtol=1e-5, atol=1e-9)
    np.testing.assert_array_equal(S.make_ddim_timesteps("quad", 20, 1000, verbose=False), z["ddim_timesteps_quad_20"])
    with pytest.raises(NotImplementedError):
        S.make_ddim_timesteps("cubic", 10, 1000)


def _tiny_latent_diffusion():
    from tests import synth
    from stablediffusion_amd import instantiate_from_config
    cfg = {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.0120, "timesteps": 1000, "conditioning_key": "crossattn",
        "scale_factor": 0.18215, "use_ema": False,
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.TINY_UNET)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL", "params": {
            "embed_dim": 4, "ddconfig": dict(synth.TINY_DDCONFIG), "lossconfig": {"target": "torch.nn.Identity"}}},
        "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                              "params": {"freeze": True, "layer": "penultimate"}}}}
    return instantiate_from_config(cfg)        # (the ViT-H text tower is built with uninitialised weights: free)


def test_latent_diffusion_protocol_and_sampler_tables():
    """The attribute protocol the reference's sampler / txt2img read, on CPU."""
    from stablediffusion_amd.ddim import DDIMSampler
    from stablediffusion_amd.plms import PLMSSampler
    model = _tiny_latent_diffusion()
    assert model.num_timesteps == 1000 and model.parameterization == "eps" and model.scale_factor == 0.18215
    from stablediffusion_amd.encoders import FrozenOpenCLIPEmbedder
    te = model.cond_stage_model                    # yaml:63-67 names it; provided by encoders.py
    assert isinstance(te, FrozenOpenCLIPEmbedder) and (te.width, te.heads, te.layers, te.layer_idx) == (1024, 16, 24, 1)
    assert not te.weights_loaded and not te.encodes_text
    emb = torch.zeros(2, 77, 1024)
    assert model.get_learned_conditioning(emb) is emb          # ready embeddings pass through
    with pytest.raises(RuntimeError, match="no BPE vocabulary"):
        model.get_learned_conditioning(["a prompt"])
    with pytest.raises(RuntimeError, match="no weights"):
        model.get_learned_conditioning(torch.zeros(1, 77, dtype=torch.long))
    assert model.model.diffusion_model is not None and model.first_stage_model.decoder is not None
    assert model.alphas_cumprod.shape == (1000,) and torch.equal(model.alphas_bar, model.alphas_cumprod)
    assert abs(float(model.betas[0]) - 0.00085) < 1e-9 and abs(float(model.alphas_cumprod[999]) - 0.0046601) < 1e-6
    z = np.load(os.path.join(GOLD, "schedule_tables.npz"))
    smp = DDIMSampler(model, device=torch.device("cpu"))
    smp.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
    np.testing.assert_allclose(smp.ddim_alphas, z["sampler50_ddim_alphas"], rtol=1e-6)
    np.testing.assert_allclose(smp.ddim_alphas_prev, z["sampler50_ddim_alphas_prev"], rtol=1e-6)
    np.testing.assert_allclose(smp.ddim_sqrt_one_minus_alphas, z["sampler50_ddim_sqrt_one_minus_alphas"], rtol=1e-6)
    np.testing.assert_allclose(smp.sqrt_recipm1_alphas_cumprod.numpy(), z["sampler50_sqrt_recipm1_alphas_cumprod"], rtol=1e-5)
    assert np.all(smp.ddim_sigmas == 0)
    with pytest.raises(ValueError):
        PLMSSampler(model, device=torch.device("cpu")).make_schedule(10, ddim_eta=0.5, verbose=False)
    with pytest.raises(TypeError):
        smp.sample(batch_size=1, shape=[4, 8, 8])
    with pytest.raises(RuntimeError):
        model.get_learned_conditioning(["a prompt"])


def test_no_cpu_fallback():
    """A product path that cannot reach the HIP kernels must raise, never compute something else."""
    from stablediffusion_amd import hip
    from tests import synth
    from stablediffusion_amd.unet import UNetModel
    m = UNetModel(**synth.TINY_UNET).eval()
    with pytest.raises(hip.SdmiError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 16, 16), torch.tensor([1]), torch.zeros(1, 7, 96))
    with pytest.raises(NotImplementedError):
        UNetModel(**dict(synth.TINY_UNET, use_scale_shift_norm=True))
    with pytest.raises(AssertionError):
        UNetModel(**dict(synth.TINY_UNET, context_dim=None))


def test_state_dict_names_and_upstream_aliases():
    from tests import synth
    from stablediffusion_amd.unet import UNetModel, rename_upstream_keys
    from stablediffusion_amd.vae import AutoencoderKL
    z = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    m = UNetModel(**synth.TINY_UNET)
    assert sorted(m.state_dict()) == list(z["keys"])
    # zero-initialised tensors as in the reference
    assert float(m.out[2].weight.abs().max()) == 0 and float(m.input_blocks[1][0].out_layers[3].weight.abs().max()) == 0
    up = {}
    for k, v in m.state_dict().items():
        k = k.replace("time_embedding.", "time_embed.").replace(".embedding_layers.", ".emb_layers.")
        k = k.replace(".operation.", ".op.").replace(".project_in.", ".proj_in.").replace(".feed_forward.net.", ".ff.net.")
        up[k] = torch.full_like(v, 0.5)
    assert sorted(rename_upstream_keys(up)) == sorted(m.state_dict())
    r = m.load_state_dict(up, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    assert float(m.time_embedding[0].weight[0, 0]) == 0.5
    vae = AutoencoderKL(embed_dim=4, ddconfig=dict(synth.TINY_DDCONFIG), lossconfig={"target": "torch.nn.Identity"})
    zd = np.load(os.path.join(GOLD, "tiny_decoder.npz"))
    assert sorted(vae.decoder.state_dict()) == list(zd["keys"])
    vsd = {k.replace(".up_sample.", ".upsample."): v for k, v in vae.state_dict().items()}
    assert not vae.load_state_dict(vsd, strict=True).missing_keys
    # reference keyword spellings
    vae2 = AutoencoderKL(dd_config=dict(synth.TINY_DDCONFIG), loss_config={"target": "torch.nn.Identity"}, embedding_dim=4)
    assert sorted(vae2.state_dict()) == sorted(vae.state_dict())


def test_weight_packing_layouts():
    from stablediffusion_amd import ops
    w = torch.randn(40, 24, 3, 3)
    pw = ops.pack_conv3x3(w, torch.randn(40))
    assert pw.w.shape == (128, 256) and pw.cin == 24 and pw.n_out == 40 and pw.k == 256
    ref = w.permute(0, 2, 3, 1).reshape(40, 216).to(torch.bfloat16)
    assert torch.equal(pw.w[:40, :216], ref) and float(pw.w[40:].abs().max()) == 0 and float(pw.w[:, 216:].abs().max()) == 0
    w4 = torch.randn(8, 4, 3, 3)
    p4 = ops.pack_conv3x3(w4)
    assert p4.cin == 8 and torch.equal(p4.w[:8, :72].reshape(8, 9, 8)[:, :, :4], w4.permute(0, 2, 3, 1).reshape(8, 9, 4).to(torch.bfloat16))
    assert float(p4.w[:8, :72].reshape(8, 9, 8)[:, :, 4:].abs().max()) == 0
    wg, bg = torch.randn(128, 64), torch.randn(128)          # inner = 64
    pg = ops.pack_geglu(wg, bg)
    assert pg.geglu and pg.n_out == 64
    for r in (0, 15, 16, 31, 32, 47, 48, 100):
        half, col = (r % 32) // 16, (r // 32) * 16 + r % 16
        assert torch.equal(pg.w[r, :64], wg[half * 64 + col].to(torch.bfloat16)) and float(pg.bias[r]) == float(bg[half * 64 + col])
    pc = ops.pack_linear_cat([torch.ones(3, 64), 2 * torch.ones(5, 64)])
    assert pc.n_out == 8 and float(pc.w[2, 0]) == 1 and float(pc.w[3, 0]) == 2
    assert ops._auto_split_k(65536, 320, 2880) == 1 and ops._auto_split_k(128, 1280, 23040) >= 8


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_c_packers_match_the_torch_reference_bit_for_bit(dtype):
    """sdmi_pack_*_host (the C ABI's packers, host twins of the device kernels) against tests/ref_pack.py - the torch
    packers this package used before - on shapes that exercise every padding: K not a multiple of 64 / 32, N not a
    multiple of the tile, the 160-column tile, Cin = 4 (latent) / 3 (image), concatenation along N, GEGLU with and
    without bias.  fp32 = the six-plane form of the fp32 mode."""
    from stablediffusion_amd import ops
    from tests import ref_pack as R
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.7

    def same(pw, ref):
        assert pw.w.shape == ref.shape and pw.w.dtype == ref.dtype
        assert torch.equal(pw.w.view(torch.int16), ref.view(torch.int16))

    for n, k, tn in ((40, 24, None), (320, 320, None), (320, 320, 128), (1000, 72, None), (8, 1024, None), (130, 200, 160)):
        w, b = rnd(n, k), rnd(n)
        pw = ops.pack_linear(w, b, dtype=dtype, tile_n=tn)
        same(pw, R.pack_linear_cat([w], dtype, tn))
        assert torch.equal(pw.bias, b) and pw.n_out == n and pw.k == pw.w.shape[1]
        assert pw.tile_n == (tn or R.tile_n_for(n))
        if dtype == torch.float32:
            assert pw.split_cp == (k + 31) // 32 * 32 and pw.dtype == torch.bfloat16
    ws = [rnd(320, 96), rnd(320, 96), rnd(17, 96)]
    same(ops.pack_linear_cat(ws, dtype=dtype), R.pack_linear_cat(ws, dtype))
    w4 = rnd(64, 48, 1, 1)                                  # conv1x1 weights take the linear path
    same(ops.pack_linear(w4, dtype=dtype), R.pack_linear_cat([w4], dtype))
    for inner, k, with_bias in ((64, 64, True), (1280, 320, True), (48, 40, False)):
        w, b = rnd(2 * inner, k), (rnd(2 * inner) if with_bias else None)
        pg = ops.pack_geglu(w, b, dtype=dtype)
        ref_w, ref_b = R.pack_geglu(w, b, dtype)
        same(pg, ref_w)
        assert pg.geglu and pg.n_out == inner and pg.tile_n == 128
        assert (pg.bias is None) == (ref_b is None) and (ref_b is None or torch.equal(pg.bias, ref_b))
    for cout, cin, tn in ((40, 24, None), (320, 4, None), (3, 128, None), (8, 3, None), (160, 40, 128), (512, 64, None)):
        w = rnd(cout, cin, 3, 3)
        pc = ops.pack_conv3x3(w, rnd(cout), dtype=dtype, tile_n=tn)
        same(pc, R.pack_conv3x3(w, dtype, tn))
        assert pc.cin == (6 * ((cin + 31) // 32 * 32) if dtype == torch.float32 else (cin + 7) // 8 * 8)


def test_c_packer_argument_checks():
    from stablediffusion_amd import hip
    import ctypes as C
    lib = hip.lib()
    w = torch.zeros(4, 8)
    out = torch.zeros(128, 64, dtype=torch.int16)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.sdmi_pack_linear_host(p(w), 4, 8, 8, 0, 4, p(out), 0, 128) == 0
    assert lib.sdmi_pack_linear_host(None, 4, 8, 8, 0, 4, p(out), 0, 128) == -1
    assert lib.sdmi_pack_linear_host(p(w), 4, 8, 8, 0, 4, p(out), 0, 96) == -1            # tile_n must be 128 | 160
    assert lib.sdmi_pack_linear_host(p(w), 4, 8, 4, 0, 4, p(out), 0, 128) == -1           # ld_w < K
    assert lib.sdmi_pack_linear_host(p(w), 4, 8, 8, 2, 4, p(out), 0, 128) == -1           # rows past rows_total
    assert lib.sdmi_pack_geglu_host(p(w), None, 2, 8, 8, p(out), None, 0) == -1           # inner % 16
    assert lib.sdmi_pack_bytes(320, 320, 160) == 320 * 320 * 2 and lib.sdmi_pack_bytes(321, 64, 160) == 480 * 64 * 2
    assert lib.sdmi_pack_tile_n(320, 0) == 160 and lib.sdmi_pack_tile_n(320, 1) == 128 and lib.sdmi_pack_tile_n(512, 0) == 128
    assert lib.sdmi_pack_linear_k(320, 0) == 320 and lib.sdmi_pack_linear_k(72, 1) == 128 and lib.sdmi_pack_linear_k(40, 2) == 384
    assert lib.sdmi_pack_conv3x3_cin(4, 0) == 8 and lib.sdmi_pack_conv3x3_cin(4, 2) == 192
    assert lib.sdmi_pack_conv3x3_k(4, 0) == 128 and lib.sdmi_pack_conv3x3_k(320, 0) == 2880


def test_dp_shard_ranges():
    from stablediffusion_amd import dp
    for gb, world in ((64, 8), (10, 4), (3, 8), (0, 2)):
        rs = [dp.shard_range(gb, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == gb
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in rs) - min(e - b for b, e in rs) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(8, 2, 2)
    a = dp.shard_latents(10, 1, 4, (4, 8, 8))
    full = dp.shard_latents(10, 0, 1, (4, 8, 8))
    b, e = dp.shard_range(10, 1, 4)
    assert torch.equal(a, full[b:e])          # independent of the GPU count


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from stablediffusion_amd import dp
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
gb = 5                                        # ragged: rank 0 owns 3 images, rank 1 owns 2
lat = dp.shard_latents(gb, rank, world, (4, 8, 8))
# stand-in for "sample + decode": a deterministic per-image function (the real path needs a GPU)
img = ((lat.mean(dim=(1, 2, 3), keepdim=True).expand(-1, 3, 16, 16).permute(0, 2, 3, 1) * 50 + 128)
       .clamp(0, 255).to(torch.uint8).contiguous())
out = dp.gather_images(img, gb, rank, world)
if rank == 0:
    full = dp.shard_latents(gb, 0, 1, (4, 8, 8))
    ref = ((full.mean(dim=(1, 2, 3), keepdim=True).expand(-1, 3, 16, 16).permute(0, 2, 3, 1) * 50 + 128)
           .clamp(0, 255).to(torch.uint8))
    assert out.shape == (gb, 16, 16, 3) and torch.equal(out, ref), "gather order / content"
    print("DP_OK")
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


def test_dp_gather_world2_gloo():
    """N > 1 path on CPU: two ranks (gloo), ragged shards, gather to rank 0 in global-index order equals the
    single-process result."""
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(_DP_WORKER)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29517", script, ROOT],
                           capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DP_OK" in p.stdout


def test_txt2img_surface(tmp_path):
    """scripts/txt2img.py surface: option names/defaults (txt2img.py:67-243), checkpoint loader semantics
    (txt2img.py:21-64: non-strict load, upstream key spellings, nothing executed from the file), loud failure
    off the GPU."""
    import yaml
    from tests import synth
    from stablediffusion_amd import txt2img
    a = txt2img.parse_args([])
    assert (a.steps, a.n_iter, a.n_samples, a.scale, a.height, a.width, a.channel, a.factor) == (50, 3, 3, 9.0, 512, 512, 4, 8)
    assert a.ddim_eta == 0.0 and a.seed == 42 and a.precision == "autocast" and not a.plms and a.device == "cuda"
    a = txt2img.parse_args(["--plms", "--H", "768", "--W", "768", "--outdir", "x", "--from_file", "p.txt", "--repeat", "2"])
    assert a.plms and a.height == 768 and a.out_dir == "x" and a.repeat == 2
    assert list(txt2img.chunk(range(5), 2)) == [(0, 1), (2, 3), (4,)]
    up = txt2img.DEFAULT_CONFIG["model"]["params"]["unet_config"]["params"]
    assert up["model_channels"] == 320 and up["context_dim"] == 1024 and up["use_fp16"]

    cfg = {"model": {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": {
        "linear_start": 0.00085, "linear_end": 0.012, "timesteps": 1000, "scale_factor": 0.18215,
        "conditioning_key": "crossattn",
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.TINY_UNET)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                               "params": {"embed_dim": 4, "ddconfig": dict(synth.TINY_DDCONFIG)}},
        "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                              "params": {"width": 64, "heads": 1, "layers": 2, "vocab_size": 512}}}}}
    path = tmp_path / "tiny.yaml"
    path.write_text(yaml.safe_dump(cfg))
    assert txt2img.load_config(str(path)) == cfg
    # a checkpoint in upstream spelling, saved as plain tensors
    from stablediffusion_amd.util import instantiate_from_config
    ref_model = instantiate_from_config(cfg["model"])
    sd = {}
    for k, v in ref_model.state_dict().items():
        sd[k.replace(".up_sample.", ".upsample.")] = torch.full_like(v, 0.25) if v.is_floating_point() else v
    ckpt = tmp_path / "tiny.ckpt"
    torch.save({"state_dict": sd, "global_step": 7}, ckpt)
    with pytest.raises(ValueError, match="GPU only"):
        txt2img.load_model_from_config(cfg, str(ckpt), device="cpu")
    got, step = txt2img._read_checkpoint(str(ckpt))
    assert step == 7 and set(got) == set(sd)
    m = instantiate_from_config(cfg["model"])
    missing, unexpected = m.load_state_dict(got, strict=False)
    assert not [k for k in missing if "diffusion_model" in k or "first_stage" in k] and not unexpected
    assert float(m.model.diffusion_model.out[2].weight.flatten()[0]) == 0.25
    with pytest.raises(NotImplementedError):
        txt2img.main(txt2img.parse_args(["--ipex"]))
    with pytest.raises(NotImplementedError):
        txt2img.main(txt2img.parse_args(["--dpm"]))


def test_narrow_conv_is_never_split(lib):
    """the UNet `out` conv (320 -> 4) at batch 2 / batch 1 got split-K 4 / 8 from the occupancy model, and
    sdmi_gemm's narrow conv kernel has no split-K form - the launch then took the 128-column tiled kernel plus a reduce
    pass for 4 real columns.  The host plan now pins split_k = 1 wherever the narrow kernel applies."""
    import torch
    from stablediffusion_amd import hip, ops
    w = torch.randn(4, 320, 3, 3)
    pw = ops.pack_conv3x3(w, torch.zeros(4), dtype=torch.float16)
    for m_rows, side in ((8192, 64), (4096, 64), (2 * 32 * 32, 32)):
        nb = m_rows // (side * side)
        assert ops._plan_gemm(m_rows, 4, pw.k, pw.tile_n)[1] > 1          # what the tiled kernel would have been given
        a = hip.GemmArgs()
        a.mode, a.H, a.W_in, a.Cin, a.Ho, a.Wo, a.stride, a.M = 1, side, side, 320, side, side, 1, m_rows
        out = torch.empty(nb, 4, side, side)
        keep = ops._fill_common(a, pw, out, True, None, side * side, None, 0, 2, 1.0, m_rows)
        assert keep is None and a.split_k == 1
    # a conv the narrow kernel does not take keeps its split
    w2 = torch.randn(320, 320, 3, 3)
    pw2 = ops.pack_conv3x3(w2, torch.zeros(320), dtype=torch.float16)
    a = hip.GemmArgs()
    a.mode, a.H, a.W_in, a.Cin, a.Ho, a.Wo, a.stride, a.M = 1, 64, 64, 320, 64, 64, 1, 8192
    assert not ops._narrow_conv_applies(a, pw2)


def test_fold_policy_of_layernorm_consuming_projections():
    """ops.fold_applies: folded wherever there is no plain twin; with one, the GEGLU projections run unfolded from 4096 rows
    on and the K = 320 projections the panel kernel serves (3 x 320 columns, GEGLU) from 16384 rows on."""
    import types
    from stablediffusion_amd import ops
    mk = lambda **kw: types.SimpleNamespace(**{"ln_colsum": object(), "plain": object(), "geglu": False, "k": 640, "n_out": 1920, **kw})
    assert not ops.fold_applies(mk(ln_colsum=None), 128)
    assert ops.fold_applies(mk(plain=None, geglu=True), 1 << 20)
    assert ops.fold_applies(mk(geglu=True), 2048) and not ops.fold_applies(mk(geglu=True), 4096)
    assert ops.fold_applies(mk(k=320, n_out=960), 8192) and not ops.fold_applies(mk(k=320, n_out=960), 16384)
    # the level-0 GEGLU projection below the panel kernel's rows (UNet batch 2: 8192 rows) stays folded
    assert ops.fold_applies(mk(geglu=True, k=320, n_out=1280), 8192) and not ops.fold_applies(mk(geglu=True, k=320, n_out=1280), 16384)
    assert ops.fold_applies(mk(k=320, n_out=320), 65536) and ops.fold_applies(mk(), 65536)


def test_layernorm_fold_algebra_of_the_packers():
    """ops._fold_ln / _ln_colsum (host side of sdmi_gemm_args.ln_colsum): LayerNorm(x) @ W^T + b equals
    rstd * (x @ W'^T - mean * rowsum(W')) + b' with W' = W * gamma and b' = b + W @ beta - here in fp64 with the row
    statistics taken as (sum, sum of squares), which is what the kernels receive."""
    import torch
    import torch.nn.functional as F
    from stablediffusion_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(37, 320, generator=g) * 2.0 + 0.8).double()
    w, b = torch.randn(96, 320, generator=g) * 0.05, torch.randn(96, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    wf, bf = ops._fold_ln(w, b, (gamma, beta, 1e-5))
    colsum = ops._ln_colsum(wf)
    s1, s2 = x.sum(1), (x * x).sum(1)
    mean = s1 / 320
    rstd = (s2 / 320 - mean * mean + 1e-5).rsqrt()
    folded = rstd[:, None] * (x @ wf.double().T - mean[:, None] * colsum.double()[None, :]) + bf.double()
    ref = F.linear(F.layer_norm(x, (320,), gamma.double(), beta.double(), 1e-5), w.double(), b.double())
    assert (folded - ref).abs().max() < 1e-5 * ref.abs().max()
    # no bias: the folded layer still gets one (W @ beta)
    wf2, bf2 = ops._fold_ln(w, None, (gamma, beta))
    assert torch.equal(wf2, wf) and torch.allclose(bf2, bf - b, atol=1e-6)



def test_no_packed_fma_with_a_broadcast_second_source():
    """A guard on the built library's machine code (needs llvm-objdump, no GPU).  v_pk_fma_f32 with the broadcast operand on
    its SECOND source (op_sel:[0,1,0]) came out without its low product in lanes 48-63 now and then on MI355X when two 8-wave
    workgroups shared a CU (the 128 x 64 tile's LayerNorm-folded epilogue, round 3: csrc/sdmi_gemm_kernel.h lnf_apply).  The
    instruction alone is fine (tools/micro/pk_fma_opsel_probe.hip), the cause inside that compiled sequence is open; the
    source forms in use compile to the first-source broadcast, which never showed it, and this test fails if a compiler
    change brings the other form back anywhere in libsdmi_hip.so."""
    import struct
    from stablediffusion_amd import hip
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(hip.LIB_PATH) or not os.path.exists(objdump):
        pytest.skip("library or llvm-objdump not present")
    data = open(hip.LIB_PATH, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)][1:]      # (the first header is the host library itself)
    assert starts, "no gfx950 code objects found in the library"
    hits, kernels = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for n, i in enumerate(starts):
            if data[i + 4] != 2 or struct.unpack_from("<H", data, i + 0x12)[0] != 224:      # ELF64, EM_AMDGPU
                continue
            shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
            path = os.path.join(tmp, f"co{n}.elf")
            with open(path, "wb") as f:
                f.write(data[i:i + shoff + shentsize * shnum])
            out = subprocess.run([objdump, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
            name = None
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    name, kernels = m.group(1), kernels + 1
                elif "v_pk_fma_f32" in line and "op_sel:[0,1,0]" in line:
                    hits.append((name, line.strip()))
    assert kernels > 50, kernels
    assert not hits, hits[:4]


def test_plan_takes_the_small_tile_where_it_was_measured_to_win():
    """ops._plan_gemm (pure host logic): the 64-column tile for the launches of a few hundred rows - the UNet's
    projections at batch 2, the 8 x 8 level, the one-row-tile convs - and the larger tiles everywhere the sweeps
    (profiles/r03d_small_tile_sweep.txt, r03d_small_conv_sweep.txt) showed them ahead."""
    from stablediffusion_amd import ops
    lin = lambda m, n, k, small=True: ops._plan_gemm(m, n, k, 160, False, False, small, False)
    conv = lambda m, n, cin, small=True: ops._plan_gemm(m, n, 9 * cin, 160, False, False, small, True)
    assert lin(512, 1280, 1280) == (64, 1, 3, 64) and lin(128, 1280, 1280) == (64, 1, 3, 64)
    assert lin(2048, 640, 640) == (128, 1, 3, 64) and lin(8192, 320, 320) == (128, 1, 3, 64)
    assert lin(2048, 640, 2560) == (128, 1, 3, 64)              # replaces split 4 + reduce
    assert lin(512, 1280, 5120)[3] == 160 and lin(512, 1280, 5120)[1] > 1      # long K against the rows: stays split
    for big in ((65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (65536, 960, 320), (8192, 960, 320)):
        assert lin(*big)[3] == 160, big
    assert lin(512, 1280, 1280, small=False)[3] == 160
    assert conv(128, 1280, 1280)[3] == 64 and conv(128, 1280, 1280)[1] > 1    # one row tile: split on the small tile
    assert conv(8192, 320, 320) == (128, 1, 3, 64) and conv(4096, 320, 320) == (128, 1, 3, 64)
    for big in ((512, 1280, 2560), (2048, 640, 640), (8192, 320, 640), (16384, 640, 640), (65536, 320, 320), (1024, 1280, 1280)):
        assert conv(*big)[3] == 160, big
    # GEGLU and weights without whole 64-row tiles never take it (the caller's small_ok), and a split is only planned
    # on the small tile for convs of at most 256 rows
    assert all(conv(m, 1280, 1280)[1] == 1 or conv(m, 1280, 1280)[3] == 160 for m in (512, 1024, 2048))


def test_bench_roofline_model_prices_each_launch_at_its_larger_bound():
    """bench.py's `roofline_model` (host logic): compulsory bytes per traced launch and, per family, the sum of
    max(FLOPs / peak, bytes / peak) over the measured time."""
    import bench
    # K = 320 projection with a residual: 2 M (K + 2 N) + 2 N K bytes; 107 FLOP/B, under the machine balance of 312
    m, n, k = 65536, 320, 320
    shp = (m, n, k, 0, 0, 0, 1, 0, 1)
    by = bench.launch_bytes("linear", 2 * m * n * k, shp)
    assert by == 2 * (m * k + n * k + 2 * m * n)
    assert 100 < 2 * m * n * k / by < 110
    # GEGLU halves the stored columns; an fp32 output doubles its bytes; batched launches multiply
    assert bench.launch_bytes("linear", 0, (128, 640, 64, 0, 0, 1, 0, 0, 1)) == 2 * (128 * 64 + 640 * 64 + 128 * 320)
    assert bench.launch_bytes("linear", 0, (128, 640, 64, 0, 0, 0, 0, 1, 3)) == 3 * (2 * (128 * 64 + 640 * 64) + 4 * 128 * 640)
    # conv3x3: the input image once (stride 2 reads four times the output pixels, the x2 upsample a quarter)
    assert bench.launch_bytes("conv3x3", 0, (1024, 64, 9 * 32, 1, 0, 0, 0, 0, 1)) == 2 * (1024 * 32 + 64 * 288 + 1024 * 64)
    assert bench.launch_bytes("conv3x3", 0, (1024, 64, 9 * 32, 2, 0, 0, 0, 0, 1)) == 2 * (4096 * 32 + 64 * 288 + 1024 * 64)
    assert bench.launch_bytes("conv3x3", 0, (1024, 64, 9 * 32, 1, 1, 0, 0, 0, 1)) == 2 * (256 * 32 + 64 * 288 + 1024 * 64)
    assert bench.launch_bytes("attention", 0, (2, 5, 4096, 77)) == 2 * 5 * (2 * 4096 + 2 * 77) * 64 * 2
    assert bench.launch_bytes("groupnorm", 12345, (2, 64, 320)) == 12345
    # a 5-field shape (older traces) still prices: no residual, 16-bit output, one batch
    assert bench.launch_bytes("linear", 0, (128, 640, 64, 0, 0)) == 2 * (128 * 64 + 640 * 64 + 128 * 640)
    tr = [("linear", 2 * m * n * k, 40e-6, 1, shp),                                    # HBM-bound by the model
          ("linear", 2 * 4096 * 1280 * 1280, 20e-6, 1, (4096, 1280, 1280, 0, 0, 0, 0, 0, 1))]      # MFMA-bound
    rm = bench.roofline_model([(2, tr)], 2.5e15, 8e12)["linear"]
    t0, t1 = by / 8e12, 2 * 4096 * 1280 * 1280 / 2.5e15
    assert abs(rm["frac"] - (t0 + t1) / 60e-6) < 1e-9
    assert abs(rm["hbm_bound_time_share"] - 40 / 60) < 1e-9
    fam, _ = bench.family_table([(1, tr)])
    assert fam["linear"]["roofline_model"]["frac"] > fam["linear"]["frac"]
