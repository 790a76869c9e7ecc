"""Dense BEV / head layers restated on torch-CPU (NCHW, fp32), following
  quant/smoothquant.py:38-99 (SQConv2d.forward — unfold, column abs-max, s = a^alpha / w^(1-alpha), fake-quant, matmul),
  pcdet/models/backbones_2d/base_bev_backbone.py:30-112 (BaseBEVBackbone),
  pcdet/models/backbones_2d/map_to_bev/height_compression.py:19-23,
  pcdet/models/dense_heads/center_head.py:12-46,75-97,385-391.
The TensorQuantizer inside is oracle.quantizer (parity unpinned); the SQConv2d arithmetic is pinned against the
reference class itself by tests/golden/sqconv2d_*.npz (made by tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import quantizer as Q


def _fq(t, amax, bits):
    return torch.from_numpy(Q.fake_quant(t.numpy(), np.asarray(amax, dtype=np.float32), bits))


def smooth_scale(act_scale, w_scale, alpha):
    """quant/smoothquant.py:75-76: act_scale**alpha / w_scale**(1-alpha), zeros → 1.

    torch.pow(x, 0.5) is computed as sqrt(x). torch's CPU sqrt is a vectorised routine whose last bit varies with the
    host's SIMD level (observed: 16 vs 260 of 1152 values off by one ulp on two machines), while CUDA/HIP sqrt and
    numpy's are IEEE correctly rounded — so the restatement pins the IEEE result."""
    a = act_scale.numpy().astype(np.float32)
    w = w_scale.numpy().astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        if alpha == 0.5:
            s = np.sqrt(a) / np.sqrt(w)
        else:
            s = np.power(a, np.float32(alpha)) / np.power(w, np.float32(1 - alpha))
    s = s.astype(np.float32)
    s[s == 0] = 1
    return torch.from_numpy(s)


def sq_conv2d(x, weight, bias, stride, padding, alpha, w_bits=8, a_bits=8, amax_x=None, amax_w=None, return_stats=False,
              quantize=True, scale_override=None):
    """x [B,C,H,W], weight [O,I,k,k]. alpha=None → no smoothing (plain per-tensor act / per-oc weight fake quant).
    amax_x / amax_w: calibrated amax (static PTQ: values beyond them clip at -2^(b-1) / 2^(b-1)-1). quantize=False is the
    calibration forward of quant/quantize.py:175-195 (quantizers disabled, calibrators enabled): the smoothed operands go
    through unquantised and the amax the two quantizers see is returned in the stats."""
    O, I, k, _ = weight.shape
    ick = I * k * k
    w = weight.reshape(O, ick)
    bs, _, h_in, w_in = x.shape
    cols = F.unfold(x, kernel_size=k, dilation=1, padding=padding, stride=stride)
    cols = cols.permute(0, 2, 1).reshape(-1, ick)
    if alpha is not None:
        w_scale = w.abs().max(dim=0)[0]
        act_scale = cols.abs().max(dim=0)[0]
        scale = smooth_scale(act_scale, w_scale, alpha)
        if return_stats:
            ieee_scale = scale
        if scale_override is not None:      # golden-vector tests: the scale bits the reference's torch build produced
            scale = torch.as_tensor(scale_override, dtype=torch.float32)
        w = w * scale
        cols = cols / scale
    aw = Q.reduce_amax(w.numpy(), axis=0) if amax_w is None else np.asarray(amax_w, np.float32).reshape(O, 1)
    ax = Q.reduce_amax(cols.numpy(), axis=None) if amax_x is None else np.float32(amax_x)
    wq = _fq(w, aw, w_bits) if quantize else w
    xq = _fq(cols, ax, a_bits) if quantize else cols
    y = xq @ wq.T
    y = y.view(bs, -1, O).permute(0, 2, 1)
    h_out = (h_in + 2 * padding - (k - 1) - 1) // stride + 1
    w_out = (w_in + 2 * padding - (k - 1) - 1) // stride + 1
    y = y.reshape(bs, O, h_out, w_out)
    if bias is not None:
        y = y + bias.view(1, O, 1, 1)
    if return_stats:
        return y, {"amax_x": np.float32(ax), "amax_w": aw.reshape(-1), "scale": None if alpha is None else ieee_scale.numpy()}
    return y


def bn2d(x, bn, eps):
    return F.batch_norm(x, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, eps)


def height_compression(features, indices, spatial_shape, batch_size):
    """features [M,C], indices [M,4] (b,z,y,x) → [B, C*D, H, W] (dense() then view)."""
    D, H, W = spatial_shape
    C = features.shape[1]
    dense = torch.zeros((batch_size, C, D, H, W), dtype=torch.float32)
    idx = torch.as_tensor(indices).long()
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.as_tensor(features)
    return dense.view(batch_size, C * D, H, W)
