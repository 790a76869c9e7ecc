"""Tensor-level wrappers over the C ABI: PyTorch-ROCm tensors in, tensors out.

torch is used here only for device memory and the current HIP stream. Every function enqueues on
`torch.cuda.current_stream()` and returns without synchronising; dynamic sizes stay on the device
(`count` tensors, int32[1]).
"""
import ctypes

import torch

from . import lib as _lib
from .lib import Conv2dGeom, DecodeCfg, I3, VoxelCfg, check

i32 = torch.int32
f32 = torch.float32


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "aq3d ops take device tensors (no CPU fallback)"
    assert t.is_contiguous(), "aq3d ops take contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pow2_at_least(n):
    t = 1024
    while t < n:
        t <<= 1
    return t


def table_size_for(cap):
    return _pow2_at_least(2 * max(int(cap), 1))


# ------------------------------------------------------------------ voxelization
def make_voxel_cfg(point_cloud_range, voxel_size, grid_size, num_features, point_stride, batch_col,
                   max_points_per_voxel, max_voxels):
    cfg = VoxelCfg()
    for j in range(3):
        cfg.range_min[j] = float(point_cloud_range[j])
        cfg.voxel_size[j] = float(voxel_size[j])
        cfg.grid[j] = int(grid_size[j])
    cfg.num_features = int(num_features)
    cfg.point_stride = int(point_stride)
    cfg.batch_col = int(bool(batch_col))
    cfg.max_points_per_voxel = int(max_points_per_voxel)
    cfg.max_voxels = int(max_voxels)
    return cfg


def voxelize(points, cfg, cap, batch_index=0, append=False, out=None, workspace=None):
    """points f32 [N, stride] → (coords i32 [cap,4], feats f32 [cap,C], npts i32 [cap], count i32 [1])."""
    L = _lib.load()
    n = int(points.shape[0])
    dev = points.device
    if out is None:
        coords = torch.empty((cap, 4), dtype=i32, device=dev)
        feats = torch.empty((cap, cfg.num_features), dtype=f32, device=dev)
        npts = torch.empty((cap,), dtype=i32, device=dev)
        count = torch.zeros((1,), dtype=i32, device=dev)
    else:
        coords, feats, npts, count = out
    need = L.aq3d_voxelize_workspace_bytes(max(n, 1))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    check(L.aq3d_voxelize(_p(points), n, ctypes.byref(cfg), int(batch_index), int(bool(append)), _p(coords), _p(feats),
                          _p(npts), _p(count), int(cap), _p(workspace), workspace.numel(), _stream()),
          "aq3d_voxelize")
    return coords, feats, npts, count


def mean_vfe(voxels, num_points):
    L = _lib.load()
    m, P, C = voxels.shape
    out = torch.empty((m, C), dtype=f32, device=voxels.device)
    check(L.aq3d_mean_vfe(_p(voxels), _p(num_points.to(f32).contiguous()), m, P, C, _p(out), _stream()), "aq3d_mean_vfe")
    return out


# ------------------------------------------------------------------ sparse index generation
class HashTable:
    def __init__(self, cap, device):
        self.size = table_size_for(cap)
        self.keys = torch.empty((self.size,), dtype=torch.int64, device=device)
        self.vals = torch.empty((self.size,), dtype=i32, device=device)


def hash_build(coords, count, shape, table=None):
    L = _lib.load()
    cap = coords.shape[0]
    table = table or HashTable(cap, coords.device)
    check(L.aq3d_hash_build(_p(coords), _p(count), cap, I3(*shape), _p(table.keys), _p(table.vals), table.size,
                            _stream()), "aq3d_hash_build")
    return table


def subm_neighbors(coords, count, shape, ksize, table, out=None):
    L = _lib.load()
    cap = coords.shape[0]
    kvol = ksize[0] * ksize[1] * ksize[2]
    nbr = out if out is not None else torch.empty((cap, kvol), dtype=i32, device=coords.device)
    check(L.aq3d_subm_neighbors(_p(coords), _p(count), cap, I3(*shape), I3(*ksize), _p(table.keys), _p(table.vals),
                                table.size, _p(nbr), _stream()), "aq3d_subm_neighbors")
    return nbr


def conv_out_shape(shape, ksize, stride, pad):
    return [(shape[i] + 2 * pad[i] - ksize[i]) // stride[i] + 1 for i in range(3)]


def conv_outputs(coords_in, count_in, shape_in, ksize, stride, pad, in_table, cap_out, workspace=None):
    """→ (coords_out i32 [cap_out,4], count_out i32 [1], nbr i32 [cap_out,kvol], out_table)."""
    L = _lib.load()
    dev = coords_in.device
    cap_in = coords_in.shape[0]
    kvol = ksize[0] * ksize[1] * ksize[2]
    out_table = HashTable(cap_out, dev)
    coords_out = torch.empty((cap_out, 4), dtype=i32, device=dev)
    count_out = torch.zeros((1,), dtype=i32, device=dev)
    nbr = torch.empty((cap_out, kvol), dtype=i32, device=dev)
    need = L.aq3d_conv_outputs_workspace_bytes(cap_in, kvol, out_table.size)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    check(L.aq3d_conv_outputs(_p(coords_in), _p(count_in), cap_in, I3(*shape_in), I3(*ksize), I3(*stride), I3(*pad),
                              _p(in_table.keys), _p(in_table.vals), in_table.size, _p(coords_out), _p(count_out),
                              cap_out, _p(nbr), _p(out_table.keys), _p(out_table.vals), out_table.size,
                              _p(workspace), workspace.numel(), _stream()), "aq3d_conv_outputs")
    return coords_out, count_out, nbr, out_table


# ------------------------------------------------------------------ sparse conv + quantizer
def absmax_cols(x, count, amax=None):
    L = _lib.load()
    cap, C = x.shape
    if amax is None:
        amax = torch.zeros((C,), dtype=f32, device=x.device)
    check(L.aq3d_absmax_cols(_p(x), _p(count), cap, C, _p(amax), _stream()), "aq3d_absmax_cols")
    return amax


def quantize_rows(x, amax, count, bits=8, per_tensor=False, want_codes=True, want_fq=False):
    """→ (codes i8 [cap,C] | None, fq f32 [cap,C] | None, dequant f32 [C])."""
    L = _lib.load()
    cap, C = x.shape
    codes = torch.empty((cap, C), dtype=torch.int8, device=x.device) if want_codes else None
    fq = torch.empty((cap, C), dtype=f32, device=x.device) if want_fq else None
    dq = torch.empty((C,), dtype=f32, device=x.device)
    check(L.aq3d_quantize_rows(_p(x), _p(amax), _p(count), cap, C, int(bits), int(bool(per_tensor)), _p(codes), _p(fq),
                               _p(dq), _stream()), "aq3d_quantize_rows")
    return codes, fq, dq


ACT_F32, ACT_I8 = 0, 1
W_F32, W_I8 = 0, 1
FLAG_RELU = 1


def spconv(feats_in, weight, nbr, count_out, cout, *, act_dequant=None, w_dequant=None, bias=None, post_scale=None,
           post_shift=None, residual=None, relu=False, absmax_out=None, out=None):
    """Gather-GEMM sparse conv. feats_in f32|i8 [*, cin]; weight f32|i8 [kvol, cout, cin]; nbr i32 [cap_out, kvol]."""
    L = _lib.load()
    cap_out, kvol = nbr.shape
    cin = feats_in.shape[1]
    assert weight.shape == (kvol, cout, cin), (tuple(weight.shape), (kvol, cout, cin))
    act_mode = ACT_I8 if feats_in.dtype == torch.int8 else ACT_F32
    w_mode = W_I8 if weight.dtype == torch.int8 else W_F32
    y = out if out is not None else torch.empty((cap_out, cout), dtype=f32, device=nbr.device)
    check(L.aq3d_spconv(_p(feats_in), act_mode, _p(act_dequant), _p(weight), w_mode, _p(w_dequant), _p(bias),
                        _p(post_scale), _p(post_shift), _p(residual), _p(nbr), _p(count_out), cap_out, cin, cout, kvol,
                        FLAG_RELU if relu else 0, _p(y), _p(absmax_out), _stream()), "aq3d_spconv")
    return y
