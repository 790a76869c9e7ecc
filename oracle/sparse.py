"""Sparse 3-D convolution (spconv SubMConv3d / SparseConv3d) and QConvNd restated in numpy — PARITY UNPINNED for
the third-party spconv arithmetic; follows the call sites pcdet/models/backbones_3d/spconv_backbone.py:8-67,184-295
and the quant wrapper quant/quant.py:36-58.

Conventions: indices rows (b, z, y, x); weights KRSC [Cout, kz, ky, kx, Cin];
out[o] = sum_k W[:, k, :] @ in[o*stride - pad + k]; SubM keeps the input index set; a strided-conv output site is
active iff one active input falls in its window; out_shape = (in + 2p - k)//s + 1.
"""
import itertools

import numpy as np

from . import quantizer as Q


def _key(idx, shape):
    idx = idx.astype(np.int64)
    return ((idx[:, 0] * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


class _Lookup:
    def __init__(self, indices, shape):
        k = _key(indices, shape)
        self.order = np.argsort(k, kind="stable")
        self.sorted = k[self.order]
        self.shape = shape

    def find(self, idx, valid):
        k = _key(idx, self.shape)
        pos = np.searchsorted(self.sorted, k)
        pos = np.minimum(pos, self.sorted.size - 1)
        hit = valid & (self.sorted[pos] == k)
        return np.where(hit, self.order[pos], -1)


def _offsets(ksize):
    return list(itertools.product(range(ksize[0]), range(ksize[1]), range(ksize[2])))


def out_shape(shape, ksize, stride, pad):
    return [(shape[i] + 2 * pad[i] - ksize[i]) // stride[i] + 1 for i in range(3)]


def subm_neighbors(indices, shape, ksize=(3, 3, 3)):
    """nbr i32 [M, kvol]: input row at indices[m] + (k - centre), or -1."""
    lk = _Lookup(indices, shape)
    cols = []
    for (kz, ky, kx) in _offsets(ksize):
        d = np.array([0, kz - ksize[0] // 2, ky - ksize[1] // 2, kx - ksize[2] // 2], dtype=np.int64)
        q = indices.astype(np.int64) + d[None, :]
        valid = np.all((q[:, 1:] >= 0) & (q[:, 1:] < np.asarray(shape)[None, :]), axis=1)
        cols.append(lk.find(np.where(valid[:, None], q, 0), valid))
    return np.stack(cols, axis=1).astype(np.int32)


def conv_outputs(indices, shape, ksize, stride, pad):
    """Active output set (sorted by linear key) + gather table nbr i32 [M_out, kvol] of a strided sparse conv."""
    oshape = out_shape(shape, ksize, stride, pad)
    ks, st, pd = np.asarray(ksize), np.asarray(stride), np.asarray(pad)
    cand = []
    for off in _offsets(ksize):
        num = indices[:, 1:].astype(np.int64) + pd[None, :] - np.asarray(off)[None, :]
        ok = np.all((num >= 0) & (num % st[None, :] == 0), axis=1)
        o = num // st[None, :]
        ok &= np.all(o < np.asarray(oshape)[None, :], axis=1)
        cand.append(np.concatenate([indices[ok, :1].astype(np.int64), o[ok]], axis=1))
    cand = np.concatenate(cand, axis=0)
    keys = _key(cand, oshape)
    _, first = np.unique(keys, return_index=True)
    out_idx = cand[first].astype(np.int32)                        # sorted by key
    lk = _Lookup(indices, shape)
    cols = []
    for off in _offsets(ksize):
        q = out_idx.astype(np.int64).copy()
        q[:, 1:] = q[:, 1:] * st[None, :] - pd[None, :] + np.asarray(off)[None, :]
        valid = np.all((q[:, 1:] >= 0) & (q[:, 1:] < np.asarray(shape)[None, :]), axis=1)
        cols.append(lk.find(np.where(valid[:, None], q, 0), valid))
    return out_idx, np.stack(cols, axis=1).astype(np.int32), oshape


ACCUM_DTYPE = np.float32     # tests flip this to float64 to measure the path's sensitivity to summation order


def gather_conv(features, weight_krsc, nbr, bias=None):
    """out[m] = sum_k W_k @ features[nbr[m,k]] in fp32; weight KRSC [Cout, kz,ky,kx, Cin]."""
    cout, cin = weight_krsc.shape[0], weight_krsc.shape[-1]
    w = weight_krsc.reshape(cout, -1, cin).astype(ACCUM_DTYPE)      # [Cout, kvol, Cin]
    out = np.zeros((nbr.shape[0], cout), dtype=ACCUM_DTYPE)
    for k in range(nbr.shape[1]):
        rows = np.nonzero(nbr[:, k] >= 0)[0]
        if rows.size:
            out[rows] += features[nbr[rows, k]].astype(ACCUM_DTYPE) @ w[:, k, :].T
    out = out.astype(np.float32)
    if bias is not None:
        out = out + bias.astype(np.float32)[None, :]
    return out


def qconv_weight(weight_krsc, w_bits=8):
    """quant/quant.py:37-48 — view [oc, ic*k^3], per-row (axis 0) fake-quant, restore KRSC."""
    oc = weight_krsc.shape[0]
    w2 = np.moveaxis(weight_krsc, -1, 1).reshape(oc, -1)           # permute(0, dim-1, 1..) → [oc, ic, kz,ky,kx]
    amax = Q.reduce_amax(w2, axis=0)
    wq, codes, div = Q.fake_quant(w2, amax, w_bits, return_codes=True)
    shape = (oc, weight_krsc.shape[-1]) + tuple(weight_krsc.shape[1:-1])
    back = lambda t: np.ascontiguousarray(np.moveaxis(t.reshape(shape), 1, -1))
    return back(wq), back(codes), div.reshape(oc)


def qconv_act(features, act_bits=8, cw=True, amax=None):
    """quant/quant.py:21-31,50-52 — per-column (axis=1) or per-tensor fake-quant of the feature rows."""
    if amax is None:
        amax = Q.reduce_amax(features, axis=1 if cw else None)
    return Q.fake_quant(features, amax, act_bits), amax


def batchnorm(x, bn, eps):
    """nn.BatchNorm1d / 2d in eval mode on the channel-last axis."""
    w, b, mean, var = (np.asarray(bn[k], dtype=np.float32) for k in ("weight", "bias", "running_mean", "running_var"))
    inv = (w / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return (x * inv + (b - mean * inv)).astype(np.float32)
