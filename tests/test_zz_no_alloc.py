"""sdmi_groupnorm_noalloc (include/sdmi.h): the no-allocation variant of sdmi_groupnorm, whose per-slab statistics
live in a static device array inside the library.  Through the C ABI, against sdmi_groupnorm on the same inputs,
bit for bit (same kernels, same summation order), on two calls in a row - the second with a smaller input, so stale
contents of the shared array from the first call would show - plus the size limit's error.  Collected last on purpose
(the round-end run is `pytest -x`)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(fn_name, x, gamma, beta, eps, silu, with_ws):
    from stablediffusion_amd import hip
    lib = hip.lib()
    nb, hw, c = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    out = torch.empty_like(x)
    dt = hip.DTYPE_F16 if x.dtype == torch.float16 else hip.DTYPE_BF16
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    if with_ws:
        nbytes = lib.sdmi_groupnorm_workspace_bytes(nb, hw)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        st = lib.sdmi_groupnorm(p(x), None, c, 0, p(gamma), p(beta), p(out), nb, hw, eps, silu, dt, p(ws), nbytes, stream)
    else:
        st = lib.sdmi_groupnorm_noalloc(p(x), None, c, 0, p(gamma), p(beta), p(out), nb, hw, eps, silu, dt, stream)
    torch.cuda.synchronize()
    return st, out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_groupnorm_noalloc_matches_groupnorm_twice(dev, dtype):
    from stablediffusion_amd import hip
    g = torch.Generator().manual_seed(7)
    # first call: 4 images of 64x64 pixels (two-launch path: 64 slabs per image in the static array);
    # second call: smaller in every dimension (2 images of 48x48 pixels, fewer channels)
    for nb, side, c in ((4, 64, 320), (2, 48, 128)):
        x = (torch.randn(nb, side, side, c, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
        gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
        beta = (0.1 * torch.randn(c, generator=g)).to(dev)
        st_ref, ref = _call("sdmi_groupnorm", x, gamma, beta, 1e-5, 1, True)
        st, out = _call("sdmi_groupnorm_noalloc", x, gamma, beta, 1e-5, 1, False)
        assert st_ref == hip.SDMI_OK and st == hip.SDMI_OK
        assert bool(torch.isfinite(out.float()).all())
        assert torch.equal(out, ref), f"noalloc differs from sdmi_groupnorm at nb={nb}, {side}x{side}, C={c}"
        # and against an independent fp32 formula, to the 16-bit output rounding
        xf = x.float().reshape(nb, side * side, 32, c // 32)
        mu = xf.mean(dim=(1, 3), keepdim=True)
        var = xf.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((xf - mu) * torch.rsqrt(var + 1e-5)).reshape(nb, side, side, c) * gamma + beta
        y = y * torch.sigmoid(y)
        tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
        assert float((out.float() - y).abs().max()) < tol * float(y.abs().max())


def test_groupnorm_noalloc_rejects_oversize(dev):
    """17 images x 256 slabs = 4352 > 4096 slab summaries: refused with SDMI_ERR_TOO_LARGE, nothing launched."""
    from stablediffusion_amd import hip
    lib = hip.lib()
    nb, side, c = 17, 256, 32
    assert lib.sdmi_groupnorm_workspace_bytes(nb, side * side) > 4096 * 32 * 3 * 4
    x = torch.zeros(nb, side, side, c, dtype=torch.float16, device=dev)
    out = torch.full_like(x, 7.0)
    gamma = torch.ones(c, device=dev)
    beta = torch.zeros(c, device=dev)
    st, _ = -99, None
    p = lambda t: C.c_void_p(t.data_ptr())
    st = lib.sdmi_groupnorm_noalloc(p(x), None, c, 0, p(gamma), p(beta), p(out), nb, side * side, 1e-5, 0, hip.DTYPE_F16,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert st == -5 and b"no-allocation" in lib.sdmi_status_str(st)
    assert float(out.min()) == 7.0          # untouched
    # 16 images of the same size are within the limit
    st = lib.sdmi_groupnorm_noalloc(p(x[:16]), None, c, 0, p(gamma), p(beta), p(out[:16]), 16, side * side, 1e-5, 0,
                                    hip.DTYPE_F16, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert st == hip.SDMI_OK


def test_groupnorm_noalloc_limit_counts_the_slabs_the_launch_uses(dev):
    """512 images of 32 x 32 pixels run as the fused single launch (one slab entry per image: 512 <= 4096), although
    sdmi_groupnorm_workspace_bytes() - an upper bound that always counts ceil(hw / 64) = 16 slabs here - exceeds the
    array's size: the limit is the one include/sdmi.h states, not the workspace formula."""
    from stablediffusion_amd import hip
    lib = hip.lib()
    nb, side, c = 512, 32, 64
    assert lib.sdmi_groupnorm_workspace_bytes(nb, side * side) > 4096 * 32 * 3 * 4
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(nb, side, side, c, generator=g) * 1.3 - 0.2).to(torch.float16).to(dev)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
    beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    st_ref, ref = _call("sdmi_groupnorm", x, gamma, beta, 1e-6, 0, True)
    st, out = _call("sdmi_groupnorm_noalloc", x, gamma, beta, 1e-6, 0, False)
    assert st_ref == hip.SDMI_OK and st == hip.SDMI_OK
    assert torch.equal(out, ref)
