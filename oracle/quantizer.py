"""TensorQuantizer (pytorch_quantization) restated in fp32 numpy — PARITY UNPINNED (library absent).

Semantics recalled from the library: amax = max|x| over every dim except `axis` (dynamic) or a
calibrated value; scale = (2^(bits-1) - 1) / amax; where amax <= 2^-24 the multiply uses scale 0 and the divide 1;
q = clamp(round_half_even(x * scale), -2^(bits-1), 2^(bits-1) - 1); output = q / scale.
Construction sites: quant/quant.py:14-32, quant/quantize.py:64-75.
"""
import numpy as np

EPS = np.float32(1.0 / (1 << 24))


def reduce_amax(x, axis=None):
    a = np.abs(np.asarray(x, dtype=np.float32))
    if axis is None:
        return a.max() if a.size else np.float32(0)
    red = tuple(i for i in range(a.ndim) if i != axis)
    return a.max(axis=red, keepdims=True)


def fake_quant(x, amax, bits=8, return_codes=False):
    x = np.asarray(x, dtype=np.float32)
    amax = np.asarray(amax, dtype=np.float32)
    qmax = np.float32(2.0 ** (bits - 1) - 1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = qmax / amax
    zero = amax <= EPS
    mul = np.where(zero, np.float32(0), scale).astype(np.float32)
    div = np.where(zero, np.float32(1), scale).astype(np.float32)
    q = np.clip(np.rint(x * mul), -qmax - 1, qmax).astype(np.float32)     # np.rint = round half to even
    out = (q / div).astype(np.float32)
    if return_codes:
        return out, q, div
    return out


class MaxCalibrator:
    """pytorch_quantization.calib.MaxCalibrator: running element-wise max of per-batch amax (quant/quantize.py:175-207)."""

    def __init__(self, axis=None):
        self.axis = axis
        self.amax = None

    def collect(self, x):
        a = reduce_amax(x, self.axis)
        self.amax = a if self.amax is None else np.maximum(self.amax, a)
