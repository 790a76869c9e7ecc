"""Generates tests/golden/*.npz by running the REFERENCE's own importable classes in the build container
(/root/reference is not available on the GPU box; only these arrays travel).

    sqconv2d_*.npz    quant/smoothquant.py::SQConv2d.forward, instantiated the way quant/quantize.py:59-63 does
                      (__new__ + copied vars of an nn.Conv2d), with the restated TensorQuantizer injected as
                      _weight_quantizer / _input_quantizer (pytorch_quantization itself is not installed).
    bev_backbone.npz  pcdet/models/backbones_2d/base_bev_backbone.py::BaseBEVBackbone (fp32) on a seeded input.

Usage (build container only):  python tests/golden/make_golden.py

Kept as the record of how the committed fixtures were produced in round 2. Since round 3 the build environment refuses to execute
modules of the reference checkout, so this script is no longer run; the .npz files are the pinned data.
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import quantizer as Q  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


class RestatedQuantizer:
    """Stand-in for pytorch_quantization's TensorQuantizer (dynamic max, signed 8 bit)."""

    def __init__(self, axis, bits=8):
        self.axis, self.bits = axis, bits

    def __call__(self, x):
        amax = Q.reduce_amax(x.detach().numpy(), axis=self.axis)
        return torch.from_numpy(Q.fake_quant(x.detach().numpy(), amax, self.bits))


def make_sqconv(sm, cin, cout, H, W, stride, padding, alpha, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=padding, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / np.sqrt(9 * cin))
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    inst = sm.SQConv2d.__new__(sm.SQConv2d)
    for k, val in vars(conv).items():
        if isinstance(val, tuple):
            val = val[0]
        setattr(inst, k, val)
    inst._weight_quantizer = RestatedQuantizer(axis=0)
    inst._input_quantizer = RestatedQuantizer(axis=None)
    inst.scaling_factor = alpha
    x = torch.relu(torch.randn(1, cin, H, W, generator=g)) * (torch.rand(1, cin, 1, 1, generator=g) * 3 + 0.1)
    with torch.no_grad():
        y = inst.forward(x.clone())
        # the smoothing scale exactly as smoothquant.py:69-76 computes it (same expressions, same torch build, same machine): torch's
        # CPU pow is not correctly rounded, so this is what pins WHICH last bit the golden output was made with
        cols = torch.nn.functional.unfold(x, kernel_size=3, dilation=1, padding=padding, stride=stride).permute(0, 2, 1).reshape(-1, cin * 9)
        w_scale = torch.max(conv.weight.view(cout, cin * 9).abs().detach(), dim=0)[0]
        act_scale = torch.max(cols.abs().detach(), dim=0)[0]
        scale = act_scale ** alpha / w_scale ** (1 - alpha)
        scale[scale == 0] = 1
    return dict(x=x.numpy(), weight=conv.weight.detach().numpy(), bias=conv.bias.detach().numpy() if bias else np.zeros(0, np.float32),
                y=y.numpy(), stride=np.int64(stride), padding=np.int64(padding), alpha=np.float64(alpha), scale_torch=scale.numpy())


def main():
    sm = _load(os.path.join(REF, "quant/smoothquant.py"), "ref_smoothquant")
    np.savez_compressed(os.path.join(HERE, "sqconv2d_16x8.npz"), **make_sqconv(sm, 16, 8, 12, 10, 1, 1, 0.5, 1))
    np.savez_compressed(os.path.join(HERE, "sqconv2d_64x32_s2.npz"), **make_sqconv(sm, 64, 32, 16, 14, 2, 1, 0.5, 2, bias=False))
    np.savez_compressed(os.path.join(HERE, "sqconv2d_32x16_a03.npz"), **make_sqconv(sm, 32, 16, 9, 11, 1, 1, 0.3, 3))
    np.savez_compressed(os.path.join(HERE, "sqconv2d_128x64.npz"), **make_sqconv(sm, 128, 64, 14, 12, 1, 1, 0.5, 4))
    np.savez_compressed(os.path.join(HERE, "sqconv2d_256x128_s2.npz"), **make_sqconv(sm, 256, 128, 12, 12, 2, 1, 0.5, 5))

    bb = _load(os.path.join(REF, "pcdet/models/backbones_2d/base_bev_backbone.py"), "ref_bev")

    class Cfg(dict):
        __getattr__ = dict.get

    mc = Cfg(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[16, 32], UPSAMPLE_STRIDES=[1, 2],
             NUM_UPSAMPLE_FILTERS=[32, 32])
    net = bb.BaseBEVBackbone(mc, 32).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.05 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.9 + 0.2 * torch.rand(m.running_var.shape, generator=g))
        x = torch.randn(1, 32, 16, 12, generator=g)
        y = net({"spatial_features": x})["spatial_features_2d"]
    out = {"x": x.numpy(), "y": y.numpy()}
    for k, v in net.state_dict().items():
        out["state." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "bev_backbone.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
