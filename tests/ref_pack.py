"""The torch weight packers the package used before the C ABI's sdmi_pack_* existed - kept here as the independent
reference the C packers (host and device) are checked against bit for bit.  Each returns the packed 16-bit matrix
[Npad, Kpad] (and, for GEGLU, the permuted bias)."""
import torch

BF16 = torch.bfloat16
_BN, _BK, _SPLIT_PAD = 128, 64, 32


def _round_up(x, m):
    return (x + m - 1) // m * m


def tile_n_for(n, geglu=False):
    """160-column tiles where they divide the width and the layer is not GEGLU, else 128."""
    return 160 if (n % 160 == 0 and not geglu) else _BN


def pad_pack(w2d, dtype=BF16, tile_n=_BN):
    n, k = w2d.shape
    out = torch.zeros(_round_up(n, tile_n), _round_up(k, _BK), dtype=dtype, device=w2d.device)
    out[:n, :k] = w2d.to(dtype)
    return out


def split_planes(w):
    """fp32 [..., K] -> bf16 [..., 6, Kp]: weight planes (w0 w1 w0 w2 w1 w0) matching the activation planes
    (x0 x0 x1 x0 x1 x2) of sdmi_f32_split3, with w = w0 + w1 + w2 exactly."""
    k = w.shape[-1]
    kp = _round_up(k, _SPLIT_PAD)
    wz = torch.zeros(*w.shape[:-1], kp, dtype=torch.float32, device=w.device)
    wz[..., :k] = w
    w0 = wz.to(BF16)
    r1 = wz - w0.float()
    w1 = r1.to(BF16)
    w2 = (r1 - w1.float()).to(BF16)
    return torch.stack([w0, w1, w0, w2, w1, w0], dim=-2), kp


def pack_linear_cat(weights, dtype=BF16, tile_n=None):
    w = torch.cat([x.detach().reshape(x.shape[0], -1).float() for x in weights], dim=0)
    tn = tile_n or tile_n_for(w.shape[0])
    if dtype == torch.float32:
        planes, kp = split_planes(w)
        return pad_pack(planes.reshape(w.shape[0], 6 * kp), BF16, tn)
    return pad_pack(w, dtype, tn)


def pack_geglu(weight, bias, dtype=BF16):
    w = weight.detach().float()
    inner = w.shape[0] // 2
    r = torch.arange(2 * inner, device=w.device)
    src = ((r % 32) // 16) * inner + (r // 32) * 16 + (r % 16)
    b = bias.detach().float()[src].contiguous() if bias is not None else None
    if dtype == torch.float32:
        planes, kp = split_planes(w[src])
        return pad_pack(planes.reshape(2 * inner, 6 * kp), BF16), b
    return pad_pack(w[src], dtype), b


def pack_conv3x3(weight, dtype=BF16, tile_n=None):
    w = weight.detach().float()
    cout, cin = w.shape[0], w.shape[1]
    tn = tile_n or tile_n_for(cout)
    if dtype == torch.float32:
        planes, cp = split_planes(w.permute(0, 2, 3, 1))              # [cout, 3, 3, 6, cp]: six planes inside every tap
        return pad_pack(planes.reshape(cout, 9 * 6 * cp), BF16, tn)
    cin_p = _round_up(cin, 8)
    wp = torch.zeros(cout, 3, 3, cin_p, dtype=torch.float32, device=w.device)
    wp[..., :cin] = w.permute(0, 2, 3, 1)
    return pad_pack(wp.reshape(cout, 9 * cin_p), dtype, tn)
