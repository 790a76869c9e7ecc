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


def _pv(t):
    """pointer of a (possibly strided) view; the caller passes the pixel stride separately"""
    if t is None:
        return None
    assert t.is_cuda, "aq3d ops take device tensors (no CPU fallback)"
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


def mean_vfe(voxels, num_points, rows=None, out=None):
    """voxels f32 [m, P, C] → mean features; rows i32 [m] (optional) sends voxel v to row rows[v] of `out` (-1: dropped)."""
    L = _lib.load()
    m, P, C = voxels.shape
    if out is None:
        out = torch.empty((m, C), dtype=f32, device=voxels.device)
    check(L.aq3d_mean_vfe(_p(voxels), _p(num_points.to(f32).contiguous()), m, P, C, _p(rows), _p(out), _stream()), "aq3d_mean_vfe")
    return out


# ------------------------------------------------------------------ sparse index levels
class IndexLevel:
    """Device arrays of one index level (include/aq3d.h: aq3d_level) + the ctypes struct that names them."""

    def __init__(self, shape, batch, cap, device, stats=None, sorted_rows=True):
        L = _lib.load()
        self.shape, self.batch, self.cap = [int(v) for v in shape], int(batch), int(cap)
        self.words = int(L.aq3d_level_words(I3(*self.shape), self.batch))
        self.bits = torch.zeros((self.words,), dtype=torch.int64, device=device)
        self.base = torch.zeros((self.words,), dtype=i32, device=device)
        self.row_of_rank = torch.zeros((self.cap,), dtype=i32, device=device)
        self.coords = torch.zeros((self.cap, 4), dtype=i32, device=device)
        # count | total[2] share one 4-int row so that a plan can read every level's totals back with one copy
        self.stats = stats if stats is not None else torch.zeros((4,), dtype=i32, device=device)
        self.count, self.total = self.stats[0:1], self.stats[1:3]
        c = _lib.Level()
        for j in range(3):
            c.shape[j] = self.shape[j]
        c.batch, c.cap, c.flags = self.batch, self.cap, (1 if sorted_rows else 0)       # AQ3D_LEVEL_SORTED
        self.sorted_rows = bool(sorted_rows)
        c.bits, c.base, c.row_of_rank = self.bits.data_ptr(), self.base.data_ptr(), self.row_of_rank.data_ptr()
        c.coords, c.count, c.total = self.coords.data_ptr(), self.count.data_ptr(), self.total.data_ptr()
        self.c = c

    def live(self):
        """(rows, total sites, largest per-sample total) — one device→host read."""
        v = self.stats[:3].cpu().tolist()
        return int(v[0]), int(v[1]), int(v[2])


def index_workspace(max_cap, max_words, max_points, device):
    """Scratch of the index builders (zero-initialised: its first 256 bytes hold self-resetting arrival counters)."""
    need = _lib.load().aq3d_index_workspace_bytes(int(max_cap), int(max_words), int(max_points))
    return torch.zeros((need,), dtype=torch.uint8, device=device)


def _ws_for(level, points, workspace):
    need = _lib.load().aq3d_index_workspace_bytes(level.cap, level.words, int(points))
    if workspace is None or workspace.numel() < need:
        workspace = torch.zeros((need,), dtype=torch.uint8, device=level.bits.device)
    return workspace


def level_from_points(points, cfg, level, feats_out=None, npts_out=None, batch_index=0, workspace=None):
    """Hard voxelization + mean VFE straight into `level` (rows in the level's order). → (feats f32 [cap, C], npts i32 [cap])."""
    L = _lib.load()
    n = int(points.shape[0])
    if feats_out is None:
        feats_out = torch.zeros((level.cap, cfg.num_features), dtype=f32, device=points.device)
    if npts_out is None:
        npts_out = torch.zeros((level.cap,), dtype=i32, device=points.device)
    workspace = _ws_for(level, n, workspace)
    check(L.aq3d_level_from_points(_p(points), n, ctypes.byref(cfg), int(batch_index), ctypes.byref(level.c), _p(feats_out),
                                   _p(npts_out), _p(workspace), workspace.numel(), _stream()), "aq3d_level_from_points")
    return feats_out, npts_out


def level_from_coords(coords, level, workspace=None, rows_out=None):
    """coords i32 [n, 4] rows (b, z, y, x) → builds `level`, returns row_of_input i32 [n]."""
    L = _lib.load()
    n = int(coords.shape[0])
    if rows_out is None:
        rows_out = torch.empty((max(n, 1),), dtype=i32, device=coords.device)
    workspace = _ws_for(level, n, workspace)
    check(L.aq3d_level_from_coords(_p(coords) if n else None, n, ctypes.byref(level.c), _p(rows_out), _p(workspace),
                                   workspace.numel(), _stream()), "aq3d_level_from_coords")
    return rows_out[:n]


def level_downsample(level_in, ksize, stride, pad, level_out, workspace=None):
    L = _lib.load()
    workspace = _ws_for(level_out, 0, workspace)
    check(L.aq3d_level_downsample(ctypes.byref(level_in.c), I3(*ksize), I3(*stride), I3(*pad), ctypes.byref(level_out.c),
                                  _p(workspace), workspace.numel(), _stream()), "aq3d_level_downsample")
    return level_out


def nbr_jobs(jobs):
    """ctypes array of aq3d_nbr_job from (level_in, ksize, stride, pad, nbr tensor i32 [cap_out, kvol]) tuples."""
    arr = (_lib.NbrJob * len(jobs))()
    for q, (lin, ks, st, pd, nbr) in zip(arr, jobs):
        assert nbr.dtype == i32 and nbr.is_contiguous() and nbr.shape[1] == ks[0] * ks[1] * ks[2]
        q.level_in = ctypes.pointer(lin.c)
        for j in range(3):
            q.ksize[j], q.stride[j], q.pad[j] = int(ks[j]), int(st[j]), int(pd[j])
        q.nbr = nbr.data_ptr()
    return arr


def level_neighbors(level_out, jobs):
    """Gather tables of `level_out` (<= 4 per launch). jobs: nbr_jobs(...) array or a list of tuples."""
    if isinstance(jobs, (list, tuple)):
        jobs = nbr_jobs(jobs)
    L = _lib.load()
    i = 0
    while i < len(jobs):          # one launch: <= 4 tables and <= 64 table entries per row
        n, width = 0, 0
        while i + n < len(jobs) and n < 4:
            kv = jobs[i + n].ksize[0] * jobs[i + n].ksize[1] * jobs[i + n].ksize[2]
            if n > 0 and width + kv > 64:
                break
            n, width = n + 1, width + kv
        sub = ctypes.cast(ctypes.byref(jobs, i * ctypes.sizeof(_lib.NbrJob)), ctypes.POINTER(_lib.NbrJob))
        check(L.aq3d_level_neighbors(ctypes.byref(level_out.c), sub, n, _stream()), "aq3d_level_neighbors")
        i += n


def subm_neighbors(level, ksize=(3, 3, 3), out=None):
    """Gather table of a submanifold conv on `level`: i32 [cap, kvol]."""
    kvol = ksize[0] * ksize[1] * ksize[2]
    nbr = out if out is not None else torch.empty((level.cap, kvol), dtype=i32, device=level.bits.device)
    level_neighbors(level, [(level, ksize, (1, 1, 1), tuple(k // 2 for k in ksize), nbr)])
    return nbr


def conv_out_shape(shape, ksize, stride, pad):
    return [(shape[i] + 2 * pad[i] - ksize[i]) // stride[i] + 1 for i in range(3)]


def level_merge(levels, scales, level_out, workspace=None):
    """VoxelNeXt bev_out index: union of <= 3 levels projected onto (y, x) → 2-D `level_out`."""
    L = _lib.load()
    n = len(levels)
    ptrs = (ctypes.c_void_p * n)(*[ctypes.addressof(l.c) for l in levels])
    sc = (ctypes.c_int32 * n)(*[int(v) for v in scales])
    workspace = _ws_for(level_out, 0, workspace)
    check(L.aq3d_level_merge(ptrs, sc, n, ctypes.byref(level_out.c), _p(workspace), workspace.numel(), _stream()), "aq3d_level_merge")
    return level_out


def level_merge_sum(levels, feats, scales, level_out, feats_out=None):
    L = _lib.load()
    n = len(levels)
    C = feats[0].shape[1]
    ptrs = (ctypes.c_void_p * n)(*[ctypes.addressof(l.c) for l in levels])
    fp = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
    sc = (ctypes.c_int32 * n)(*[int(v) for v in scales])
    for l, f in zip(levels, feats):
        assert f.shape == (l.cap, C) and f.dtype == f32 and f.is_contiguous()
    if feats_out is None:
        feats_out = torch.zeros((level_out.cap, C), dtype=f32, device=feats[0].device)
    check(L.aq3d_level_merge_sum(ptrs, fp, sc, n, C, ctypes.byref(level_out.c), _p(feats_out), _stream()), "aq3d_level_merge_sum")
    return feats_out


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
W_F32, W_I8, W_F16S = 0, 1, 2
FLAG_RELU, FLAG_SKIP_F32, FLAG_W16 = 1, 2, 4


def spconv(feats_in, weight, nbr, count_out, cout, *, act_dequant=None, w_dequant=None, bias=None, post_scale=None,
           post_shift=None, residual=None, relu=False, absmax_out=None, out=None, next_scale=None, next_codes=None, next_bits=8):
    """Gather-GEMM sparse conv (fp32 / weight-only layers). feats_in f32|i8 [*, cin]; weight f32|i8 [kvol, cout, cin];
    nbr i32 [cap_out, kvol]. The w8a8 QConvNd layers take spconv_prep + spconv_coded."""
    L = _lib.load()
    cap_out, kvol = nbr.shape
    act_mode = ACT_I8 if feats_in.dtype == torch.int8 else ACT_F32
    w_mode = W_I8 if weight.dtype == torch.int8 else (W_F16S if weight.dtype == torch.float16 else W_F32)
    cin = feats_in.shape[1]
    assert weight.shape == ((kvol * cout, 2, cin) if w_mode == W_F16S else (kvol, cout, cin)), (tuple(weight.shape), (kvol, cout, cin))
    y = out if out is not None else torch.empty((cap_out, cout), dtype=f32, device=nbr.device)
    check(L.aq3d_spconv(_p(feats_in), act_mode, _p(act_dequant), _p(weight), w_mode, _p(w_dequant), _p(bias),
                        _p(post_scale), _p(post_shift), _p(residual), _p(nbr), _p(count_out), cap_out, int(feats_in.shape[0]),
                        cin, cout, kvol, FLAG_RELU if relu else 0, _p(y), _p(absmax_out), _p(next_scale), _p(next_codes),
                        int(next_bits), _stream()), "aq3d_spconv")
    return y


def spconv_coded_supported(cin, cout, kvol):
    return bool(_lib.load().aq3d_spconv_coded_supported(int(cin), int(cout), int(kvol)))


def spconv_prep(x, amax, count, weight_codes, w_dequant, bits=8, per_tensor=False, out=None, cap=None):
    """Quantiser + per-frame weight copy of one QConvNd layer in ONE launch (aq3d_spconv_prep).
    x f32 [cap, C]; weight_codes i8 [kvol, cout, C] → (codes i8 [cap, C], wprep i8 [kvol, cout, 3, C] (weight planes), out_mul f32 [cout]).
    x=None: weight copy only (static PTQ, once); weight_codes=None: rows only."""
    L = _lib.load()
    if out is None:
        cap_, C = x.shape
        kvol, cout, _ = weight_codes.shape
        codes = torch.zeros((cap_, C), dtype=torch.int8, device=x.device)
        wprep = torch.empty((kvol, cout, 3, C), dtype=torch.int8, device=x.device)
        out_mul = torch.empty((cout,), dtype=f32, device=x.device)
    else:
        codes, wprep, out_mul = out
    C = int(amax.shape[0])
    if weight_codes is not None:
        kvol, cout, cin = weight_codes.shape
        assert cin == C and weight_codes.dtype == torch.int8
    else:
        kvol, cout = 1, 1
    cap_ = int(x.shape[0]) if x is not None else int(cap or 1)
    check(L.aq3d_spconv_prep(_p(x), _p(amax), _p(count), cap_, C, int(bits), int(bool(per_tensor)), _p(weight_codes),
                             _p(w_dequant), kvol, cout, _p(codes) if x is not None else None, _p(wprep) if weight_codes is not None else None,
                             _p(out_mul) if weight_codes is not None else None, _stream()), "aq3d_spconv_prep")
    return codes, wprep, out_mul


def spconv_coded(codes, wprep, out_mul, nbr, count_out, *, bias=None, post_scale=None, post_shift=None,
                 residual=None, relu=False, absmax_out=None, out=None, next_scale=None, next_codes=None, next_bits=8, skip_f32=False,
                 w16=False):
    """Gather-GEMM on int8-coded activations (aq3d_spconv_coded). codes i8 [cap_in, cin]; wprep i8 [kvol, cout, 3, cin];
    w16: multiply the two upper weight planes only (16-bit fixed-point weights)."""
    L = _lib.load()
    cap_out, kvol = nbr.shape
    cap_in, cin = codes.shape
    cout = wprep.shape[1]
    assert tuple(wprep.shape) == (kvol, cout, 3, cin) and wprep.dtype == torch.int8, (tuple(wprep.shape), (kvol, cout, 3, cin))
    y = out if out is not None else torch.empty((cap_out, cout), dtype=f32, device=nbr.device)
    check(L.aq3d_spconv_coded(_p(codes), cap_in, _p(wprep), _p(out_mul), _p(bias), _p(post_scale), _p(post_shift), _p(residual),
                              _p(nbr), _p(count_out), cap_out, cin, cout, kvol,
                              (FLAG_RELU if relu else 0) | (FLAG_SKIP_F32 if skip_f32 else 0) | (FLAG_W16 if w16 else 0), _p(y),
                              _p(absmax_out),
                              _p(next_scale), _p(next_codes), int(next_bits), _stream()), "aq3d_spconv_coded")
    return y


def spconv_problems(recs):
    """ctypes array of aq3d_spconv_problem from dicts(feats_in, act_dequant, weight, w_dequant, bias, post_scale, post_shift,
    residual, nbr, count_out, feats_out, absmax_out, cout_real)."""
    arr = (_lib.SpconvProblem * len(recs))()
    for q, r in zip(arr, recs):
        for k in ("feats_in", "act_dequant", "weight", "w_dequant", "bias", "post_scale", "post_shift", "residual", "nbr",
                  "count_out", "feats_out", "absmax_out"):
            setattr(q, k, _addr(r.get(k)))
        q.cout_real = int(r.get("cout_real") or 0)
    return arr


def spconv_group(problems, act_mode, w_mode, cap_out, cap_in, cin, cout, kvol, relu):
    """One launch for up to 8 sparse convs of one shape on the f32 / int8-coded operand path."""
    check(_lib.load().aq3d_spconv_group(problems, len(problems), int(act_mode), int(w_mode), int(cap_out), int(cap_in), int(cin),
                                        int(cout), int(kvol), FLAG_RELU if relu else 0, _stream()), "aq3d_spconv_group")


# ------------------------------------------------------------------ sparse → dense
def scatter_dense(feats, coords, count, shape, batch_size, out=None):
    """→ f32 NHWC [B, H, W, C*D], channel = c*D + z (HeightCompression)."""
    L = _lib.load()
    cap, C = feats.shape
    D, H, W = shape
    if out is None:
        out = torch.empty((batch_size, H, W, C * D), dtype=f32, device=feats.device)
    check(L.aq3d_scatter_dense(_p(feats), _p(coords), _p(count), cap, C, I3(*shape), batch_size, _p(out), _stream()),
          "aq3d_scatter_dense")
    return out


# ------------------------------------------------------------------ dense convs
def conv_geom(H, W, Cin, Cout, ksize=3, stride=1, pad=1, in_ld=None, out_ld=None, out_coff=0, deconv=0, relu=False):
    g = Conv2dGeom()
    g.H, g.W, g.Cin, g.Cout, g.ksize, g.stride, g.pad = H, W, Cin, Cout, ksize, stride, pad
    if deconv:
        g.Ho, g.Wo = H, W
    else:
        g.Ho = (H + 2 * pad - ksize) // stride + 1
        g.Wo = (W + 2 * pad - ksize) // stride + 1
    g.in_ld = in_ld or Cin
    g.out_ld = out_ld or Cout
    g.out_coff = out_coff
    g.deconv = deconv
    g.relu = int(bool(relu))
    return g


def out_hw(g):
    return (g.H * g.deconv, g.W * g.deconv) if g.deconv else (g.Ho, g.Wo)


def piece_absmax(x, H, W, C, ldc=None, pieces=None):
    """x f32 NHWC [H, W, ldc] (first C channels used) → pieces f32 [16, C]."""
    L = _lib.load()
    if pieces is None:
        pieces = torch.zeros((16, C), dtype=f32, device=x.device)
    check(L.aq3d_piece_absmax(_p(x), H, W, C, ldc or C, _p(pieces), _stream()), "aq3d_piece_absmax")
    return pieces


def sq_param_bytes(Cin, Cout, ksize):
    return _lib.load().aq3d_sq_param_bytes(Cin, Cout, ksize)


def sq_prep(pieces, weight_k, w_colmax, g, alpha, params=None, w_bits=8, a_bits=8, static_amax_x=0.0,
            static_amax_w=None, amax_x_acc=None):
    """weight_k f32 [Cout, K] with K ordered (kh, kw, c). → params block (uint8 tensor).
    amax_x_acc: f32 [1] accumulator zeroed once per frame by the caller (a fresh zero tensor when omitted)."""
    L = _lib.load()
    if params is None:
        params = torch.empty((sq_param_bytes(g.Cin, g.Cout, g.ksize),), dtype=torch.uint8, device=pieces.device)
    if amax_x_acc is None:
        amax_x_acc = torch.zeros((1,), dtype=f32, device=pieces.device)
    check(L.aq3d_sq_prep(_p(pieces), _p(weight_k), _p(w_colmax), ctypes.byref(g), float(alpha if alpha is not None else -1.0),
                         int(w_bits), int(a_bits), float(static_amax_x), _p(static_amax_w), _p(amax_x_acc), _p(params),
                         _stream()), "aq3d_sq_prep")
    return params


def conv2d_i8(x, params, g, bias=None, post_scale=None, post_shift=None, out=None, pieces_out=None, pieces_mode=0):
    """pieces_out: optional f32 [16, ld] accumulator that receives the piece abs-max of the OUTPUT (fused, caller-zeroed)."""
    L = _lib.load()
    Ho, Wo = out_hw(g)
    y = out if out is not None else torch.empty((Ho, Wo, g.out_ld), dtype=f32, device=x.device)
    check(L.aq3d_conv2d_i8(_p(x), _p(params), ctypes.byref(g), _p(bias), _p(post_scale), _p(post_shift), _p(y),
                           _p(pieces_out), 0 if pieces_out is None else pieces_out.shape[1], int(pieces_mode), _stream()), "aq3d_conv2d_i8")
    return y


def _addr(t):
    return None if t is None else t.data_ptr()


def conv2d_sq16(x, params, g, w_bits, a_bits, bias=None, post_scale=None, post_shift=None, out=None, workspace=None):
    """SQConv2d with 9..16 activation bits on fake-quantised values (aq3d_conv2d_sq16); params from sq_prep(a_bits=...)."""
    L = _lib.load()
    need = L.aq3d_conv2d_sq16_workspace_bytes(g.Cin, g.Cout, g.ksize)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=x.device)
    y = out if out is not None else torch.empty((g.Ho, g.Wo, g.out_ld), dtype=f32, device=x.device)
    check(L.aq3d_conv2d_sq16(_pv(x), _p(params), ctypes.byref(g), int(w_bits), int(a_bits), _p(bias), _p(post_scale), _p(post_shift),
                             _pv(y), _p(workspace), workspace.numel(), _stream()), "aq3d_conv2d_sq16")
    return y


def conv_problems(recs):
    """ctypes array of aq3d_conv2d_problem from dicts(x, w, g, bias, post_scale, post_shift, y, pieces_out, pieces_mode)."""
    arr = (_lib.Conv2dProblem * len(recs))()
    for q, r in zip(arr, recs):
        po = r.get("pieces_out")
        q.x, q.weight_or_params, q.geom = _addr(r["x"]), _addr(r["w"]), ctypes.pointer(r["g"])
        q.bias, q.post_scale, q.post_shift = _addr(r.get("bias")), _addr(r.get("post_scale")), _addr(r.get("post_shift"))
        q.y, q.pieces_out = _addr(r["y"]), _addr(po)
        q.pieces_ld, q.pieces_mode = (0 if po is None else int(po.shape[1])), int(r.get("pieces_mode", 0))
        q.weight_split = _addr(r.get("w_split"))
    return arr


def conv2d_i8_group(problems):
    """One launch for up to 8 int8 convs of one launch class (problems: ctypes array from conv_problems)."""
    check(_lib.load().aq3d_conv2d_i8_group(problems, len(problems), _stream()), "aq3d_conv2d_i8_group")


def conv2d_f32_group(problems):
    check(_lib.load().aq3d_conv2d_f32_group(problems, len(problems), _stream()), "aq3d_conv2d_f32_group")


def sq_prep_problems(recs):
    """ctypes array of aq3d_sq_prep_problem from dicts(pieces, weight, w_colmax, g, alpha, static_x, static_w, amax_x_acc, params)."""
    arr = (_lib.SqPrepProblem * len(recs))()
    for q, r in zip(arr, recs):
        q.pieces, q.weight, q.w_colmax, q.geom = _addr(r["pieces"]), _addr(r["weight"]), _addr(r["w_colmax"]), ctypes.pointer(r["g"])
        q.alpha = float(r["alpha"] if r["alpha"] is not None else -1.0)
        q.static_amax_x = float(r.get("static_x", 0.0))
        q.static_amax_w, q.amax_x_acc, q.params = _addr(r.get("static_w")), _addr(r["amax_x_acc"]), _addr(r["params"])
    return arr


def sq_prep_group(problems, w_bits=8, a_bits=8):
    check(_lib.load().aq3d_sq_prep_group(problems, len(problems), int(w_bits), int(a_bits), _stream()), "aq3d_sq_prep_group")


def split_weight(weight_k):
    """f32 [rows, K] → f16 [rows, 2, K] (hi | lo' planes): the pre-split operand of the fp32 GEMMs (conv2d_f32(..., weight_split=))."""
    rows, K = weight_k.shape
    out = torch.empty((rows, 2, K), dtype=torch.float16, device=weight_k.device)
    check(_lib.load().aq3d_conv2d_split_weight(_p(weight_k), rows, K, _p(out), _stream()), "aq3d_conv2d_split_weight")
    return out


def conv2d_f32(x, weight_k, g, bias=None, post_scale=None, post_shift=None, out=None, pieces_out=None, weight_split=None, pieces_mode=0):
    L = _lib.load()
    Ho, Wo = out_hw(g)
    y = out if out is not None else torch.empty((Ho, Wo, g.out_ld), dtype=f32, device=x.device)
    if weight_split is not None or pieces_mode:
        conv2d_f32_group(conv_problems([dict(x=x, w=weight_k, g=g, bias=bias, post_scale=post_scale, post_shift=post_shift, y=y,
                                             pieces_out=pieces_out, pieces_mode=pieces_mode, w_split=weight_split)]))
        return y
    check(L.aq3d_conv2d_f32(_p(x), _p(weight_k), ctypes.byref(g), _p(bias), _p(post_scale), _p(post_shift), _p(y),
                            _p(pieces_out), 0 if pieces_out is None else pieces_out.shape[1], _stream()), "aq3d_conv2d_f32")
    return y


def conv_weight_to_k(weight_oihw):
    """nn.Conv2d weight [O, I, kh, kw] → [O, K] with K ordered (kh, kw, c)."""
    O = weight_oihw.shape[0]
    return weight_oihw.permute(0, 2, 3, 1).reshape(O, -1).contiguous()


def deconv_weight_to_k(weight_iohw):
    """nn.ConvTranspose2d weight [I, O, d, d] (kernel == stride) → [(i*d + j)*O + o, I]."""
    I = weight_iohw.shape[0]
    return weight_iohw.permute(2, 3, 1, 0).reshape(-1, I).contiguous()


# ------------------------------------------------------------------ decode + NMS
def make_decode_cfg(H, W, num_classes, K, has_vel, feature_map_stride, voxel_size, pc_range, post_center_range,
                    score_thresh, hm_ld, reg_ld):
    c = DecodeCfg()
    c.H, c.W, c.num_classes, c.K, c.has_vel = H, W, num_classes, K, int(bool(has_vel))
    c.feature_map_stride = float(feature_map_stride)
    for j in range(2):
        c.voxel_size[j] = float(voxel_size[j])
        c.pc_range_min[j] = float(pc_range[j])
    for j in range(6):
        c.post_center_range[j] = float(post_center_range[j])
    c.score_thresh = float(score_thresh) if score_thresh is not None else -1.0
    c.hm_ld, c.reg_ld = hm_ld, reg_ld
    return c


def center_decode(hm, center, center_z, dim, rot, vel, cfg, workspace=None, out=None):
    """→ (boxes f32 [K, 7|9], scores f32 [K], labels i32 [K], count i32 [1]); rows beyond count are undefined."""
    L = _lib.load()
    dev = hm.device
    nb = 9 if cfg.has_vel else 7
    if out is None:
        boxes = torch.zeros((cfg.K, nb), dtype=f32, device=dev)
        scores = torch.zeros((cfg.K,), dtype=f32, device=dev)
        labels = torch.zeros((cfg.K,), dtype=i32, device=dev)
        count = torch.zeros((1,), dtype=i32, device=dev)
    else:
        boxes, scores, labels, count = out
    need = L.aq3d_center_decode_workspace_bytes(ctypes.byref(cfg))
    if isinstance(workspace, str) and workspace == "min":      # the smallest workspace the call accepts: one workgroup per class
        workspace = torch.empty((cfg.num_classes * cfg.K * 8 + 256,), dtype=torch.uint8, device=dev)
    elif workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    check(L.aq3d_center_decode(_pv(hm), _pv(center), _pv(center_z), _pv(dim), _pv(rot), _pv(vel), ctypes.byref(cfg), _p(boxes),
                               _p(scores), _p(labels), _p(count), _p(workspace), workspace.numel(), _stream()),
          "aq3d_center_decode")
    return boxes, scores, labels, count


def voxel_decode(hm, center, center_z, dim, rot, vel, coords, count, cfg, workspace=None, out=None):
    """Sparse-head decode of ONE sample: rows of hm/center/... are the live sites, coords i32 [cap, 4] rows (b, z, y, x).
    → (boxes f32 [K, 7|9], scores f32 [K], labels i32 [K], count i32 [1])."""
    L = _lib.load()
    dev = hm.device
    nb = 9 if cfg.has_vel else 7
    if out is None:
        boxes = torch.zeros((cfg.K, nb), dtype=f32, device=dev)
        scores = torch.zeros((cfg.K,), dtype=f32, device=dev)
        labels = torch.zeros((cfg.K,), dtype=i32, device=dev)
        cnt = torch.zeros((1,), dtype=i32, device=dev)
    else:
        boxes, scores, labels, cnt = out
    cap = coords.shape[0]
    need = L.aq3d_voxel_decode_workspace_bytes(ctypes.byref(cfg), cap)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    assert hm.shape[0] >= cap and hm.stride(0) == cfg.hm_ld and center.stride(0) == cfg.reg_ld
    check(L.aq3d_voxel_decode(_pv(hm), _pv(center), _pv(center_z), _pv(dim), _pv(rot), _pv(vel), _p(coords), _p(count), cap,
                              ctypes.byref(cfg), _p(boxes), _p(scores), _p(labels), _p(cnt), _p(workspace), workspace.numel(),
                              _stream()), "aq3d_voxel_decode")
    return boxes, scores, labels, cnt


def nms_rotated(boxes, count, thresh, post_max, workspace=None, out=None):
    """boxes f32 [cap, ld] sorted by descending score → (keep i32 [cap], keep_count i32 [1])."""
    L = _lib.load()
    cap, ld = boxes.shape
    dev = boxes.device
    if out is None:
        keep = torch.zeros((cap,), dtype=i32, device=dev)
        kc = torch.zeros((1,), dtype=i32, device=dev)
    else:
        keep, kc = out
    need = L.aq3d_nms_workspace_bytes(cap)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    check(L.aq3d_nms_rotated(_p(boxes), ld, _p(count), cap, float(thresh), int(post_max), _p(keep), _p(kc), _p(workspace),
                             workspace.numel(), _stream()), "aq3d_nms_rotated")
    return keep, kc


def head_problems(recs):
    """ctypes array of aq3d_head_problem from dicts(hm, center, center_z, dim, rot, vel, coords, count, cfg, cand=(boxes, scores,
    labels, count), dec_ws, thresh, nms_cap, post_max, keep, keep_count, nms_ws, class_map, out=(boxes, scores, labels)). The caller
    keeps every tensor (and cfg) alive."""
    arr = (_lib.HeadProblem * len(recs))()
    for q, r in zip(arr, recs):
        q.hm, q.center, q.center_z, q.dim, q.rot = _addr(r["hm"]), _addr(r["center"]), _addr(r["center_z"]), _addr(r["dim"]), _addr(r["rot"])
        q.vel, q.coords, q.count = _addr(r.get("vel")), _addr(r.get("coords")), _addr(r.get("count"))
        q.cap = 0 if r.get("coords") is None else int(r["coords"].shape[0])
        q.sample = int(r.get("sample", -1))
        q.cfg = ctypes.pointer(r["cfg"])
        cb, cs, cl, cc = r["cand"]
        q.cand_boxes, q.cand_scores, q.cand_labels, q.cand_count = _addr(cb), _addr(cs), _addr(cl), _addr(cc)
        q.decode_workspace, q.decode_workspace_bytes = _addr(r["dec_ws"]), int(r["dec_ws"].numel())
        q.nms_thresh, q.nms_cap, q.post_max = float(r["thresh"]), int(r["nms_cap"]), int(r["post_max"])
        q.keep, q.keep_count = _addr(r["keep"]), _addr(r["keep_count"])
        q.nms_workspace, q.nms_workspace_bytes = _addr(r["nms_ws"]), int(r["nms_ws"].numel())
        q.class_map = _addr(r.get("class_map"))
        bo, so, lo = r["out"]
        q.boxes_out, q.scores_out, q.labels_out = _addr(bo), _addr(so), _addr(lo)
    return arr


def decode_nms_batch(problems):
    """top-K, decode, NMS mask, sweep and gather of up to 8 heads in five launches (aq3d_decode_nms_batch)."""
    check(_lib.load().aq3d_decode_nms_batch(problems, len(problems), _stream()), "aq3d_decode_nms_batch")


def gather_dets(boxes, scores, labels, class_map, keep, keep_count, out=None):
    """→ (boxes f32 [cap, ld], scores f32 [cap], labels i64 [cap]) rows [0, keep_count) valid; labels are 1-based."""
    L = _lib.load()
    cap, ld = boxes.shape
    dev = boxes.device
    if out is None:
        bo = torch.zeros((cap, ld), dtype=f32, device=dev)
        so = torch.zeros((cap,), dtype=f32, device=dev)
        lo = torch.zeros((cap,), dtype=torch.int64, device=dev)
    else:
        bo, so, lo = out
    check(L.aq3d_gather_dets(_p(boxes), ld, _p(scores), _p(labels), _p(class_map), _p(keep), _p(keep_count), cap, _p(bo),
                             _p(so), _p(lo), _stream()), "aq3d_gather_dets")
    return bo, so, lo


def sq_param_views(params, Cin, Cout, ksize):
    """Views into a parameter block written by sq_prep (layout of csrc/sq_layout.h)."""
    K = Cin * ksize * ksize
    al = lambda v: (v + 255) // 256 * 256
    o_r = al(K * 4)
    o_mul = o_r + al(K * 4)
    o_amaxw = o_mul + al(Cout * 4)
    o_scal = o_amaxw + al(Cout * 4)
    o_wq = o_scal + 256
    return {
        "s": params[0:K * 4].view(f32),
        "rinv": params[o_r:o_r + K * 4].view(f32),
        "out_mul": params[o_mul:o_mul + Cout * 4].view(f32),
        "amax_w": params[o_amaxw:o_amaxw + Cout * 4].view(f32),
        "scal": params[o_scal:o_scal + 32].view(f32),     # [sx, amax_x, 1/sx, ...]
        "wq": params[o_wq:o_wq + Cout * K].view(torch.int8).view(Cout, K),
    }


# ------------------------------------------------------------------ evaluation helpers
def boxes_iou3d(boxes_a, boxes_b):
    """boxes_iou3d_gpu of the reference: f32 [n, m] 3-D IoU of boxes_a [n, >=7] vs boxes_b [m, >=7]."""
    L = _lib.load()
    boxes_a, boxes_b = boxes_a.contiguous(), boxes_b.contiguous()
    n, m = boxes_a.shape[0], boxes_b.shape[0]
    out = torch.zeros((n, m), dtype=f32, device=boxes_a.device)
    check(L.aq3d_boxes_iou3d(_p(boxes_a) if n else None, n, boxes_a.shape[1], _p(boxes_b) if m else None, m, boxes_b.shape[1],
                             _p(out) if n * m else None, _stream()), "aq3d_boxes_iou3d")
    return out


def recall_counts(iou, thresholds, counts):
    """counts i32 [T] += #columns of iou whose max over rows exceeds thresholds[t]."""
    L = _lib.load()
    n, m = iou.shape
    check(L.aq3d_recall_counts(_p(iou) if n * m else None, n, m, _p(thresholds), thresholds.numel(), _p(counts), _stream()),
          "aq3d_recall_counts")
    return counts
