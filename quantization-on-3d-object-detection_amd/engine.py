"""Device plan of the CenterPoint hot path: compiles the parameter containers (modules.py) and quantisation
descriptors (quant.py) into preallocated device buffers + an ordered list of HIP kernel launches, then runs a frame.

One forward = voxelize → index levels (5 bitmap levels + gather tables) → 21 sparse convs → sparse→dense → 12 BEV convs + 2 deblocks →
shared conv + head branches → decode → rotated NMS, all enqueued on the current stream with NO device→host
synchronisation until the final read of the per-sample detection counts. Dynamic sizes stay on the device.

Memory is laid out for a 288 GB HBM3E part: every conv owns its output buffer at full capacity (no reuse
juggling), gather tables are materialised per index level ([cap, 27] i32), and all per-frame statistics (abs-max
slots, piece maxima, counters) live in ONE arena cleared by a single memset per frame.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .modules import SparseBasicBlock, SparseConvolution, SparseSequential, VoxelResBackBone8xVoxelNeXt
def _graph_replay_safe():
    import sys
    pkg = sys.modules.get(__name__.rsplit(".", 1)[0])
    return bool(getattr(pkg, "GRAPH_REPLAY_SAFE", False))


from .quant import QConv2d, QConvNd, QuantConv2d, SQConv2d, _QuantConv2dBase, quantize_weight_rows, surgery_epoch

f32, i32 = torch.float32, torch.int32


def _np(t):
    return t.detach().float().cpu().numpy()


def fold_bn(bn):
    """eval-mode BatchNorm → (scale, shift) with IEEE fp32 numpy ops: y = x*scale + shift."""
    w, b = _np(bn.weight), _np(bn.bias)
    mean, var = _np(bn.running_mean), _np(bn.running_var)
    scale = (w / np.sqrt(var + np.float32(bn.eps))).astype(np.float32)
    shift = (b - mean * scale).astype(np.float32)
    return scale, shift


class _Arena:
    """Bump allocator over one zero-initialised float32 tensor that is memset once per frame."""

    def __init__(self, device):
        self.device = device
        self.size = 0
        self.buf = None
        self.views = []

    def alloc(self, n):
        off = self.size
        self.size += (int(n) + 63) // 64 * 64
        self.views.append((off, int(n)))
        return len(self.views) - 1

    def finalize(self):
        self.buf = torch.zeros((max(self.size, 64),), dtype=f32, device=self.device)

    def view(self, handle, shape=None, dtype=f32):
        off, n = self.views[handle]
        v = self.buf[off:off + n]
        if dtype != f32:
            v = v.view(dtype)
        return v.view(shape) if shape is not None else v

    def clear(self):
        self.buf.zero_()


class SparseTensorView:
    """What the reference hands around as spconv.SparseConvTensor: features/indices of the live rows."""

    def __init__(self, features, indices, count, spatial_shape, batch_size):
        self._f, self._i, self._c = features, indices, count
        self.spatial_shape, self.batch_size = spatial_shape, batch_size

    @property
    def features(self):
        return self._f[:int(self._c.item())]

    @property
    def indices(self):
        return self._i[:int(self._c.item())]

    def dense_nhwc(self):
        D, H, W = self.spatial_shape
        return ops.scatter_dense(self._f, self._i, self._c, (D, H, W), self.batch_size)


class _Timed:
    """HIP-event bracket around a group of launches on the current stream (no host sync)."""

    def __init__(self, eng, cat):
        self.eng, self.cat = eng, cat

    def __enter__(self):
        t = self.eng.timers
        if t is not None and (self.eng.timer_filter is None or self.cat in self.eng.timer_filter):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            self.e0 = None

    def __exit__(self, *a):
        if self.e0 is not None:
            self.e1.record()
            self.eng.timers.setdefault(self.cat, []).append((self.e0, self.e1))


class LevelOverflow(ValueError):
    """An index level of a frame has more active sites than its buffers have rows (the frame was not computed)."""

    def __init__(self, level, total, cap, msg):
        super().__init__(msg)
        self.level, self.total, self.cap = int(level), int(total), int(cap)


class Engine:
    NSLOT = 2     # result slots: frame N+1 can be enqueued while frame N's detections are still being read back

    def __init__(self, det, device):
        self.device = device
        self.timers = None          # dict category -> [(event, event)] when event timing is on
        self.timer_filter = None
        self.epoch = surgery_epoch()
        self.det = det
        ds = det.dataset
        self.batch_cap = int(getattr(det, "max_batch_size", 1))
        self.num_point_features = int(ds.point_feature_encoder.num_point_features)
        self.pc_range = [float(v) for v in ds.point_cloud_range]
        self.voxel_size = [float(v) for v in ds.voxel_size]
        self.grid_size = [int(v) for v in ds.grid_size]
        proc = {c.NAME: c for c in ds.dataset_cfg.DATA_PROCESSOR}
        vcfg = proc["transform_points_to_voxels"]
        self.max_points_per_voxel = int(vcfg.MAX_POINTS_PER_VOXEL)
        self.max_voxels = int(vcfg.MAX_NUMBER_OF_VOXELS["test"])
        # Row capacities. Level 0 (voxels): det.voxel_capacity or MAX_NUMBER_OF_VOXELS x batch. A strided level can hold up to
        # prod(ceil(k / s)) = 8 times its input's sites (spconv sizes its outputs dynamically, spconv_backbone.py:230-231); its
        # buffers get det.level_capacity_factor (4) x the level-0 capacity, never more than the level has cells, or what
        # det.level_capacity = {level index: rows} says. A frame that outgrows a buffer RAISES (counts() / forward()).
        cap_opt = getattr(det, "voxel_capacity", None)
        self.cap = int(cap_opt) if cap_opt else self.max_voxels * self.batch_cap
        self.level_factor = float(getattr(det, "level_capacity_factor", None) or 4.0)
        self.level_caps = dict(getattr(det, "level_capacity", None) or {})
        self._stats_pool = torch.zeros((32, 4), dtype=i32, device=device)      # per level: count, total, per-sample max, pad
        self.arena = _Arena(device)
        self.steps = []
        self.tensors = {}
        self.sparse_w16 = bool(getattr(det, "sparse_w16", False))
        self._head_args = {}
        self._force_rows = False     # teacher-forced tests: run the rows-only quantiser even where a producer epilogue does it
        if isinstance(det.backbone_3d, VoxelResBackBone8xVoxelNeXt):
            self.fully_sparse = True
            self._build_voxelnext(det.backbone_3d, det.dense_head)
        else:
            self.fully_sparse = False
            self._build_sparse(det.backbone_3d)
            self._build_dense(det.backbone_2d, det.dense_head)
        self._decide_row_order()
        self.arena.finalize()
        # A frame is ONE linear launch plan (voxelizer, index chain, sparse convs, BEV, head, decode) = one captured hipGraph on one
        # stream. Rounds 1-2 ran CenterPoint's index chain as a phase of its own on a second stream, one frame ahead, and VoxelNeXt's
        # on a stream forked inside the frame (a parallel branch of the captured graph): four linear plans deliver 757 frames/s
        # against 649 for four split ones, VoxelNeXt 552 against 331 — parallel graph branches and second streams take the hardware
        # queues the other plans' graphs need.
        self._graph_now = self._warm_now = False
        self._static_batch = None
        self.slot = 0
        self.slot_generation = [0] * self.NSLOT
        self.use_graph = device.type == "cuda" and os.environ.get("AQ3D_GRAPH", "1") != "0"
        if self.use_graph and not _graph_replay_safe():
            import warnings
            warnings.warn("CUDA was initialised before this package was imported and DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is not set: "
                          "hipGraph replay is off (ROCm's packet-capture graph path corrupts captured kernel arguments once "
                          "memory is allocated after a capture); frames are launched eagerly, about 2x slower. Import the "
                          "package before the first CUDA call or export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.")
            self.use_graph = False
        self._graphs, self._pts_buf, self._pts_n, self._warm, self._calibrating = {}, None, 0, 0, None
        self.host_counts = torch.zeros(tuple(self.det_counts.shape), dtype=i32).pin_memory() if device.type == "cuda" else None
        self.host_stats = torch.zeros((self.NSLOT,) + tuple(self._stats_pool.shape), dtype=i32).pin_memory() if device.type == "cuda" else None
        self._slot_batch = [None] * self.NSLOT
        self._ev_done = [torch.cuda.Event() for _ in range(self.NSLOT)] if device.type == "cuda" else None
    # ------------------------------------------------------------------ helpers
    def _dev(self, arr, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    # ------------------------------------------------------------------ sparse plan primitives
    def _decide_row_order(self):
        """Class-sorted rows (AQ3D_LEVEL_SORTED: three more launches per level and frame) only where they pay: levels whose quantised
        submanifold convs have >= 128 input channels (W150k: 67 -> 54 us per 128-channel layer; 2 us per 32- / 64-channel layer, less
        than the sort costs; none at 16 channels)."""
        for li, lv in enumerate(self.levels):
            wide = any(r.get("coded") and r["level"] == li and r["nbr"] == li and r["cin"] >= 128 for r in self.sparse_convs)
            lv["index"].c.flags = 1 if wide else 0
            lv["index"].sorted_rows = wide

    def _level_cap(self, li, shape, bound=None):
        """Row capacity of level li (see __init__): override, else min(cells, bound, factor x level-0 capacity)."""
        if li in self.level_caps:
            return int(self.level_caps[li])
        if li == 0:
            return self.cap
        cells = int(np.prod(shape)) * self.batch_cap
        cap = min(cells, int(np.ceil(self.level_factor * self.cap)))
        if bound is not None:
            cap = min(cap, int(bound))
        return max(256, (cap + 255) // 256 * 256)

    def _new_level(self, shape, kvol=27, bound=None):
        dev = self.device
        li = len(self.levels)
        assert li < self._stats_pool.shape[0]
        cap = self._level_cap(li, shape, bound)
        ix = ops.IndexLevel(shape, self.batch_cap, cap, dev, stats=self._stats_pool[li])
        lv = dict(shape=list(shape), cap=cap, index=ix, coords=ix.coords, count=ix.count,
                  nbr=torch.full((cap, kvol), -1, dtype=i32, device=dev), subm_ksize=(3, 3, 3), jobs=None)
        self.levels.append(lv)
        return lv

    def _tensor(self, level, C, name):
        return dict(buf=torch.zeros((self.levels[level]["cap"], C), dtype=f32, device=self.device), C=C, level=level,
                    amax=self.arena.alloc(C), name=name)

    def _compile_spconv(self, mod, path, x, level_out, nbr, kvol, bn, relu, residual=None):
        """mod: SparseConvolution | QConvNd (w8a8, per input channel / per tensor) | QConv2d (weight-only).
        Returns the output tensor record. Output channels are padded to a multiple of 16 (MFMA tile), the pad
        columns carry zero weights / bias and unit BN scale."""
        dev, cap = self.device, self.levels[x["level"]]["cap"]     # codes / fake-quantised rows are the INPUT's rows
        q = mod if isinstance(mod, QConvNd) else None
        q2 = mod if isinstance(mod, QConv2d) else None
        conv = mod.module if (q is not None or q2 is not None) else mod
        assert isinstance(conv, SparseConvolution)
        cin, cout_real = conv.in_channels, conv.out_channels
        cout = (cout_real + 15) // 16 * 16
        w = _np(conv.weight).reshape(cout_real, kvol, cin)               # KRSC → [cout, kvol, cin]

        def padc(v, fill=0.0):
            v = np.asarray(v, dtype=np.float32)
            if v.shape[0] < cout:
                v = np.concatenate([v, np.full((cout - v.shape[0],) + v.shape[1:], fill, np.float32)], 0)
            return v

        rec = dict(path=path, cin=cin, cout=cout, cout_real=cout_real, kvol=kvol, x=x, nbr=nbr, level=level_out, relu=relu,
                   residual=residual, quant=None)
        wq_desc = q if q is not None else q2
        quant_on = wq_desc is not None and wq_desc.w_quant._if_quant
        if wq_desc is not None and wq_desc.w_quant._if_calib:   # constant weights: the max calibrator sees the same amax every batch
            wq_desc.w_quant.collect(torch.from_numpy(np.abs(np.moveaxis(w, 2, 1).reshape(cout_real, -1)).max(axis=1)))
        if quant_on:
            codes, wdq, _ = quantize_weight_rows(np.moveaxis(w, 2, 1).reshape(cout_real, -1), wq_desc.w_bits)   # [oc, ic*kvol] view
            codes = np.moveaxis(codes.reshape(cout_real, cin, kvol), 1, 2)                                    # back to [cout,kvol,cin]
            if wq_desc.w_bits <= 8 and cin % 16 == 0:
                wk = np.transpose(padc(codes.astype(np.float32)), (1, 0, 2))
                rec["weight_i8"] = self._dev(wk.astype(np.int8))              # canonical int8 codes [kvol, cout, cin]
                rec["weight"] = rec["weight_i8"]
                rec["w_dequant"] = self._dev(padc(wdq, 1.0))
            else:   # fake-quantised fp32 weights through the fp32 weight path
                wfq = (codes.astype(np.float32) * wdq[:, None, None]).astype(np.float32)
                rec["weight"] = self._dev(np.transpose(padc(wfq), (1, 0, 2)))
                rec["w_dequant"] = None
        else:
            rec["weight"] = self._dev(np.transpose(padc(w), (1, 0, 2)))
            rec["w_dequant"] = None
        if rec["weight"].dtype == f32 and cin == 128 and cout == 128:
            # an un-quantised 128 -> 128 layer runs on the fp16 pipe with split operands: its weights are split once, here
            rec["weight"] = ops.split_weight(rec["weight"].reshape(kvol * cout, cin))        # f16 [kvol * cout, 2, cin]
        rec["bias"] = self._dev(padc(_np(conv.bias))) if conv.bias is not None else None
        if bn is not None:
            s, b = fold_bn(bn)
            rec["post_scale"], rec["post_shift"] = self._dev(padc(s, 1.0)), self._dev(padc(b))
        else:
            rec["post_scale"] = rec["post_shift"] = None
        if q is not None:
            aq = q.act_quant
            rec["quant"] = dict(bits=q.act_bits, per_tensor=not q.cw, quantizer=aq, on=aq._if_quant,
                                calib=aq._if_calib,
                                static_amax=None if aq.amax is None else self._dev(_np(aq.amax).reshape(-1)))
            # production: int8 codes in HBM + per-frame weight copy (csrc/spconv_coded.hip); anything that path does not take
            # (act_bits = 16, channel counts it has no variant for) runs on fp32 fake-quantised rows through aq3d_spconv
            coded = q.act_bits <= 8 and rec["w_dequant"] is not None and ops.spconv_coded_supported(cin, cout, kvol)
            rec["coded"] = rec["quant"]["use_codes"] = coded
            if coded:
                rec["codes"] = torch.zeros((cap, cin), dtype=torch.int8, device=dev)
                # W'' = rint(code * (d_c / S) * 2^16) as three int8 planes + epilogue multipliers S * t_o / 2^16, rebuilt per frame by
                # aq3d_spconv_prep
                rec["wprep"] = torch.zeros((kvol, cout, 3, cin), dtype=torch.int8, device=dev)
                rec["out_mul"] = torch.zeros((cout,), dtype=f32, device=dev)
            else:
                rec["fq"] = torch.zeros((cap, cin), dtype=f32, device=dev)
                rec["dq"] = torch.zeros((cin,), dtype=f32, device=dev)
        out = self._tensor(level_out, cout, path)
        rec["out"] = out
        self.sparse_convs.append(rec)
        self.steps.append(("spconv", rec))
        return out

    def _compile_block(self, blk, path, x, level, nbr):
        assert isinstance(blk, SparseBasicBlock)
        t = self._compile_spconv(blk.conv1, path + ".conv1", x, level, nbr, 27, blk.bn1, True)
        return self._compile_spconv(blk.conv2, path + ".conv2", t, level, nbr, 27, blk.bn2, True, residual=x)

    def _compile_stages(self, bb, names):
        """conv_input + conv1 on level 0, then one strided SparseConv3d + residual blocks per named stage."""
        cap, dev = self.cap, self.device
        self.levels = []          # dicts: shape, cap, index (ops.IndexLevel), coords, count, nbr (submanifold table), down
        self._new_level(list(bb.sparse_shape))
        C0 = self.num_point_features
        self.vfe_feats = torch.zeros((cap, C0), dtype=f32, device=dev)
        self.vfe_npts = torch.zeros((cap,), dtype=i32, device=dev)
        self.sparse_convs = []    # compiled conv records (for introspection / tests)
        self.multi_scale = {}
        # level 0: voxels → hash + submanifold gather table (shared by conv_input 'subm1' and conv1 'res1')
        self.steps.append(("index_subm", 0))
        x_in = dict(buf=self.vfe_feats, C=C0, level=0, amax=self.arena.alloc(C0), name="vfe")
        self.steps.append(("absmax", x_in))      # only consumed if conv_input is quantised (sq=False recipe)
        ci = bb.conv_input
        x = self._compile_spconv(ci[0], "backbone_3d.conv_input.0", x_in, 0, 0, 27, ci[1], True)
        for bi, blk in enumerate(bb.conv1):
            x = self._compile_block(blk, "backbone_3d.conv1.%d" % bi, x, 0, 0)
        self.multi_scale["x_conv1"] = (x, 0)
        for si, name in enumerate(names):
            stage = getattr(bb, name)
            down = stage[0]
            conv = down[0].module if isinstance(down[0], QConvNd) else down[0]
            lv_in = self.levels[si]
            lv_out = self._new_level(ops.conv_out_shape(lv_in["shape"], conv.kernel_size, conv.stride, conv.padding))
            lv_out["down"] = dict(src=si, ksize=conv.kernel_size, stride=conv.stride, pad=conv.padding,
                                  nbr=torch.full((lv_out["cap"], 27), -1, dtype=i32, device=dev))
            self.steps.append(("index_down", si + 1))
            x = self._compile_spconv(down[0], "backbone_3d.%s.0.0" % name, x, si + 1, ("down", si + 1), 27, down[1], True)
            self.steps.append(("index_subm", si + 1))
            for bi in range(1, len(stage)):
                x = self._compile_block(stage[bi], "backbone_3d.%s.%d" % (name, bi), x, si + 1, si + 1)
            self.multi_scale["x_%s" % name] = (x, si + 1)
        return x

    # ------------------------------------------------------------------ CenterPoint sparse backbone
    def _build_sparse(self, bb):
        cap, dev = self.cap, self.device
        x = self._compile_stages(bb, ("conv2", "conv3", "conv4"))
        co = bb.conv_out
        conv = co[0].module if isinstance(co[0], QConvNd) else co[0]
        lv_in = self.levels[3]
        kvol = int(np.prod(conv.kernel_size))
        lv_out = self._new_level(ops.conv_out_shape(lv_in["shape"], conv.kernel_size, conv.stride, conv.padding))
        lv_out["down"] = dict(src=3, ksize=conv.kernel_size, stride=conv.stride, pad=conv.padding,
                              nbr=torch.full((lv_out["cap"], kvol), -1, dtype=i32, device=dev))
        self.steps.append(("index_down", 4))
        x = self._compile_spconv(co[0], "backbone_3d.conv_out.0", x, 4, ("down", 4), kvol, co[1], True)
        self.encoded = (x, 4)
        self._alloc_index_ws()
        self._plan_static_sparse()

    def _alloc_index_ws(self, n_points=0):
        """One scratch buffer for every index builder of the plan (they are ordered on the plan's stream)."""
        self._index_pts = int(n_points)
        self.index_ws = ops.index_workspace(max(lv["cap"] for lv in self.levels), max(lv["index"].words for lv in self.levels),
                                            self._index_pts, self.device)

    # ------------------------------------------------------------------ VoxelNeXt: sparse backbone, 2-D tail and head
    def _build_voxelnext(self, bb, head):
        """spconv_backbone_voxelnext.py:200-228 + voxelnext_head.py:535-560. 2-D sparse tensors are index levels of shape
        (1, H, W) with coords rows (b, 0, y, x), so that every 2-D conv runs through the same gather tables and kernels."""
        cap, dev = self.cap, self.device
        self._compile_stages(bb, ("conv2", "conv3", "conv4", "conv5", "conv6"))
        x4, l4 = self.multi_scale["x_conv4"]
        x5, l5 = self.multi_scale["x_conv5"]
        x6, l6 = self.multi_scale["x_conv6"]
        assert x4["C"] == x5["C"] == x6["C"], "bev_out concatenates rows: equal channel counts"
        _, H, W = self.levels[l4]["shape"]
        lv_m = self._new_level([1, H, W], kvol=1, bound=sum(self.levels[l]["cap"] for l in (l4, l5, l6)))
        mi = len(self.levels) - 1
        merged = self._tensor(mi, x4["C"], "backbone_3d.bev_out")
        L = ops._lib.load()
        self.merge = dict(srcs=[(x4, l4, 1), (x5, l5, 2), (x6, l6, 4)], out=merged, level=mi)
        self.steps.append(("bev_merge", None))
        self.steps.append(("absmax", merged))
        # conv_out: SparseConv2d 3x3 s1 p1 → dilated site set
        co = bb.conv_out
        conv = co[0].module if isinstance(co[0], QConv2d) else co[0]
        k3 = (1,) + tuple(conv.kernel_size)
        lv_o = self._new_level(ops.conv_out_shape([1, H, W], k3, (1,) + tuple(conv.stride), (0,) + tuple(conv.padding)),
                               kvol=int(np.prod(k3)))
        oi = len(self.levels) - 1
        lv_o["subm_ksize"] = k3
        lv_o["down"] = dict(src=mi, ksize=k3, stride=(1,) + tuple(conv.stride), pad=(0,) + tuple(conv.padding),
                            nbr=torch.full((lv_o["cap"], int(np.prod(k3))), -1, dtype=i32, device=dev))
        self.steps.append(("index_down", oi))
        x = self._compile_spconv(co[0], "backbone_3d.conv_out.0", merged, oi, ("down", oi), int(np.prod(k3)), co[1], True)
        # shared_conv: SubMConv2d 3x3 on the same sites
        sc = bb.shared_conv
        conv = sc[0].module if isinstance(sc[0], QConv2d) else sc[0]
        ks = (1,) + tuple(conv.kernel_size)
        lv_o["subm_ksize"] = ks
        if int(np.prod(ks)) != lv_o["nbr"].shape[1]:
            lv_o["nbr"] = torch.full((lv_o["cap"], int(np.prod(ks))), -1, dtype=i32, device=dev)
        self.steps.append(("index_subm", oi))
        x = self._compile_spconv(sc[0], "backbone_3d.shared_conv.0", x, oi, oi, int(np.prod(ks)), sc[1], True)
        self.encoded = (x, oi)
        self._alloc_index_ws()
        # head: per branch (SubMConv2d k + BN1d + ReLU)* + SubMConv2d 1x1; gather tables per distinct kernel size
        kh = int(head.model_cfg.get("KERNEL_SIZE_HEAD", 3))
        self.head_nbr = {}
        for k in sorted({kh, 1}):
            key = ("head", k)
            if (1, k, k) == ks:
                self.head_nbr[k] = oi
                continue
            self.head_nbr[k] = key
            lv_o.setdefault("extra_nbr", {})[k] = torch.full((lv_o["cap"], k * k), -1, dtype=i32, device=dev)
        self.steps.append(("index_head", oi))
        pp = head.model_cfg.POST_PROCESSING
        K = int(pp.MAX_OBJ_PER_SAMPLE)
        nms = pp.NMS_CONFIG
        assert nms.NMS_TYPE == "nms_gpu", "only class-agnostic rotated NMS (nms_gpu) is on the VoxelNeXt path"
        self.heads = []
        self.head_first_step = len(self.steps)
        import ctypes
        for hi, sep in enumerate(head.heads_list):
            outs = {}
            for name in sep.sep_head_dict:
                seq = getattr(sep, name)
                t = x
                path = "dense_head.heads_list.%d.%s" % (hi, name)
                for li, m in enumerate(seq):
                    if isinstance(m, SparseSequential):
                        conv = m[0].module if isinstance(m[0], QConv2d) else m[0]
                        k = conv.kernel_size[0]
                        t = self._compile_spconv(m[0], "%s.%d.0" % (path, li), t, oi, self.head_nbr[k], k * k, m[1], True)
                    else:
                        conv = m.module if isinstance(m, QConv2d) else m
                        k = conv.kernel_size[0]
                        t = self._compile_spconv(m, "%s.%d" % (path, li), t, oi, self.head_nbr[k], k * k, None, False)
                outs[name] = t
            ncls = len(head.class_names_each_head[hi])
            has_vel = "vel" in sep.sep_head_dict
            ld = {n: outs[n]["C"] for n in outs}
            assert len({ld[n] for n in outs if n != "hm"}) == 1
            cfg = ops.make_decode_cfg(H, W, ncls, K, has_vel, head.feature_map_stride, self.voxel_size[:2],
                                      self.pc_range[:2], [float(v) for v in pp.POST_CENTER_LIMIT_RANGE],
                                      pp.SCORE_THRESH, ld["hm"], ld["center"])
            nb = 9 if has_vel else 7
            B = self.batch_cap          # a batch is decoded sample by sample (voxelnext_head.py:418-488): aq3d_head_problem.sample
            rec = dict(outs=outs, cfg=cfg, has_vel=has_vel, nb=nb, K=K, level=oi,
                       thresh=float(nms.NMS_THRESH), pre=int(nms.NMS_PRE_MAXSIZE), post=int(nms.NMS_POST_MAXSIZE),
                       class_map=head.class_id_mapping_each_head[hi].to(dev).to(i32).contiguous(),
                       cand_boxes=torch.zeros((B, K, nb), dtype=f32, device=dev),
                       cand_scores=torch.zeros((B, K), dtype=f32, device=dev),
                       cand_labels=torch.zeros((B, K), dtype=i32, device=dev),
                       cand_count=torch.zeros((B, 1), dtype=i32, device=dev),
                       keep=torch.zeros((B, K), dtype=i32, device=dev),
                       out_boxes=torch.zeros((self.NSLOT, B, K, nb), dtype=f32, device=dev),
                       out_scores=torch.zeros((self.NSLOT, B, K), dtype=f32, device=dev),
                       out_labels=torch.zeros((self.NSLOT, B, K), dtype=torch.int64, device=dev),
                       dec_ws=torch.empty((B, max(L.aq3d_voxel_decode_workspace_bytes(ctypes.byref(cfg), self.levels[oi]["cap"]), 256)),
                                          dtype=torch.uint8, device=dev),
                       nms_ws=torch.empty((B, max(L.aq3d_nms_workspace_bytes(K), 256)), dtype=torch.uint8, device=dev))
            self.heads.append(rec)
        self.det_counts = torch.zeros((self.NSLOT, self.batch_cap, len(self.heads)), dtype=i32, device=dev)
        self.dense_convs = []
        self._group_sparse_head(self.head_first_step)
        self.steps.append(("voxel_decode_nms", None))
        self._plan_static_sparse()

    # ------------------------------------------------------------------ dense BEV backbone + head
    def _build_dense(self, b2d, head):
        dev, B = self.device, self.batch_cap
        D, H, W = self.levels[4]["shape"]
        Cenc = self.encoded[0]["C"]
        self.bev_shape = (D, H, W)
        self.spatial = torch.zeros((B, H, W, Cenc * D), dtype=f32, device=dev)
        self.steps.append(("scatter_dense", None))
        self.dense_convs = []
        self._pieces = {}
        self._producers = {}
        # piece maxima of a conv's input come out of the epilogue of the conv(s) that wrote it where that is cheap: int8 producers
        # feeding stride-1 consumers (register-reduced interior maximum, pieces_mode 1) and the fp32 deblocks (exact 16 classes merged
        # per workgroup in LDS); every other input gets the standalone aq3d_piece_absmax pass (per-element LDS atomics in the int8
        # epilogue measured slower than that pass)
        self._pieces_mode = {}

        def compile_conv2d(mod, path, x, xH, xW, x_ld, pad_extra, bn, relu, out=None, out_ld=None, out_coff=0):
            """mod: nn.Conv2d | SQConv2d | QuantConv2d | nn.ConvTranspose2d. x: [B, xH, xW, x_ld] buffer."""
            rec = dict(path=path, x=x, quant=None)
            if isinstance(mod, nn.ConvTranspose2d):
                d = mod.stride[0]
                assert mod.kernel_size[0] == d and mod.padding[0] == 0, "only kernel == stride transposed convs"
                cin, cout = mod.in_channels, mod.out_channels
                g = ops.conv_geom(xH, xW, cin, cout, 1, 1, 0, in_ld=x_ld, out_ld=out_ld or cout, out_coff=out_coff,
                                  deconv=d, relu=relu)
                rec["weight"] = ops.deconv_weight_to_k(mod.weight.detach().float()).to(dev)
                if cin % 32 == 0:       # the deblock GEMMs take their weights pre-split into fp16 planes (constant per model)
                    rec["w_split"] = ops.split_weight(rec["weight"])
                Ho, Wo = xH * d, xW * d
                rec["kind"] = "f32"
            else:
                k = mod.kernel_size if isinstance(mod.kernel_size, int) else mod.kernel_size[0]
                s = mod.stride if isinstance(mod.stride, int) else mod.stride[0]
                p = (mod.padding if isinstance(mod.padding, int) else mod.padding[0]) + pad_extra
                cin, cout = mod.in_channels, mod.out_channels
                g = ops.conv_geom(xH, xW, cin, cout, k, s, p, in_ld=x_ld, out_ld=out_ld or cout, out_coff=out_coff,
                                  relu=relu)
                Ho, Wo = g.Ho, g.Wo
                wk = ops.conv_weight_to_k(mod.weight.detach().float())
                is_q = isinstance(mod, _QuantConv2dBase) and mod._input_quantizer._if_quant
                if is_q and (cin % 64 != 0 or mod._weight_quantizer.num_bits > 8 or mod._input_quantizer.num_bits > 16):
                    raise NotImplementedError("quantised dense conv needs Cin %% 64 == 0, w_bits <= 8, act_bits <= 16 (%s)" % path)
                if is_q:
                    cpad = (cout + 3) // 4 * 4
                    if cpad != cout:    # pad the output channels to the 16-byte store granularity
                        wk = torch.cat([wk, torch.zeros((cpad - cout, wk.shape[1]), device=wk.device)], 0)
                        g.Cout = cpad
                    # more than 8 activation bits (the reference's A16): fake-quantised values on the fp16 pipe (aq3d_conv2d_sq16)
                    rec["kind"] = "i8" if mod._input_quantizer.num_bits <= 8 else "sq16"
                    if rec["kind"] == "sq16":
                        rec["sq16_ws"] = torch.empty((ops._lib.load().aq3d_conv2d_sq16_workspace_bytes(cin, g.Cout, k),),
                                                     dtype=torch.uint8, device=dev)
                    rec["weight"] = wk.contiguous().to(dev)
                    rec["w_colmax"] = rec["weight"].abs().max(dim=0)[0].contiguous()
                    rec["alpha"] = mod.scaling_factor
                    pkey = (x.data_ptr(), x_ld, cin, xH, xW)      # one abs-max pass per input tensor, shared by its consumers
                    if pkey in self._pieces and self._pieces_mode.get(pkey, 0) == 1 and s != 1:
                        pkey = pkey + ("exact",)                  # merged-interior maxima do not serve a strided consumer
                    rec["pieces_owner"] = pkey not in self._pieces
                    if rec["pieces_owner"]:
                        self._pieces[pkey] = self.arena.alloc(16 * cin)
                        # if the tensor is written by conv kernels, they produce its piece maxima in their epilogues
                        prods = self._producers.get(x.data_ptr(), [])
                        covered = sum(p_["geom"].Cout for p_ in prods)
                        fusable = prods and covered >= cin and x_ld == cin and xH >= 3 and xW >= 3 and len(pkey) == 5
                        if fusable and s == 1 and all(p_["kind"] == "i8" for p_ in prods):
                            for p_ in prods:
                                p_["pieces_out"] = (self._pieces[pkey], cin, 1)
                            self._pieces_mode[pkey] = 1
                            rec["pieces_owner"] = False
                        elif fusable and all(p_["kind"] == "f32" for p_ in prods):
                            for p_ in prods:        # fp32 producers (deblocks → bev_cat): merged per workgroup in LDS; interior classes
                                p_["pieces_out"] = (self._pieces[pkey], cin, 1 if s == 1 else 0)    # merged for a stride-1 consumer
                            if s == 1:
                                self._pieces_mode[pkey] = 1
                            rec["pieces_owner"] = False
                    rec["pieces"] = self._pieces[pkey]
                    rec["params"] = torch.empty((ops.sq_param_bytes(cin, g.Cout, k),), dtype=torch.uint8, device=dev)
                    rec["amax_x_acc"] = self.arena.alloc(1)
                    wq_, iq_ = mod._weight_quantizer, mod._input_quantizer
                    rec["w_bits"], rec["a_bits"] = wq_.num_bits, iq_.num_bits
                    rec["static_x"] = float(iq_.amax.reshape(-1)[0]) if iq_.amax is not None else 0.0
                    sw = None
                    if wq_.amax is not None:
                        sw = torch.zeros((g.Cout,), dtype=f32)
                        sw[:cout] = wq_.amax.reshape(-1)
                        sw = sw.to(dev)
                    rec["static_w"] = sw
                    rec["quantizers"] = (wq_, iq_)
                else:
                    rec["kind"] = "f32"
                    rec["weight"] = wk.contiguous().to(dev)
                    if isinstance(mod, _QuantConv2dBase) and (mod._input_quantizer._if_calib or mod._weight_quantizer._if_calib):
                        # calibration pass: run in fp32, harvest the amax the quantised layer would see
                        rec["calib"] = dict(pieces=self.arena.alloc(16 * cin), alpha=mod.scaling_factor,
                                            w_colmax=rec["weight"].abs().max(dim=0)[0].contiguous(),
                                            params=torch.empty((ops.sq_param_bytes(cin, (cout + 3) // 4 * 4, k),),
                                                               dtype=torch.uint8, device=dev),
                                            quantizers=(mod._weight_quantizer, mod._input_quantizer), k=k, cout=cout)
            bias = getattr(mod, "bias", None)
            cfull = g.Cout
            def padv(v, fill):
                v = np.asarray(v, dtype=np.float32)
                if len(v) < cfull:
                    v = np.concatenate([v, np.full((cfull - len(v),), fill, np.float32)])
                return self._dev(v)
            rec["bias"] = padv(_np(bias), 0.0) if bias is not None else None
            if bn is not None:
                s_, b_ = fold_bn(bn)
                rec["post_scale"], rec["post_shift"] = padv(s_, 1.0), padv(b_, 0.0)
            else:
                rec["post_scale"] = rec["post_shift"] = None
            rec["geom"] = g
            if out is None:
                out = torch.zeros((B, Ho, Wo, out_ld or g.Cout), dtype=f32, device=dev)
            rec["out"] = out
            rec["out_hw"] = (Ho, Wo)
            rec["pieces_out"] = None
            self._producers.setdefault(out.data_ptr(), []).append(rec)
            self.dense_convs.append(rec)
            self.steps.append(("conv2d", rec))
            return out, Ho, Wo

        def run_sequential(seq, path, x, xH, xW, x_ld, out=None, out_ld=None, out_coff=0):
            """nn.Sequential of [ZeroPad2d] Conv/Deconv [BN] [ReLU] groups (base_bev_backbone.py:30-66)."""
            mods = list(seq.named_children())
            i, pad_extra = 0, 0
            convs = []
            while i < len(mods):
                name, m = mods[i]
                if isinstance(m, nn.ZeroPad2d):
                    pad_extra = int(m.padding[0])
                    i += 1
                    continue
                if isinstance(m, nn.Sequential):        # nested conv-bn-relu group (head branches)
                    convs.append((path + "." + name, m, None))
                    i += 1
                    continue
                assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, _QuantConv2dBase)), (path, name, type(m))
                bn, relu, j = None, False, i + 1
                if j < len(mods) and isinstance(mods[j][1], nn.BatchNorm2d):
                    bn, j = mods[j][1], j + 1
                if j < len(mods) and isinstance(mods[j][1], nn.ReLU):
                    relu, j = True, j + 1
                convs.append((path + "." + name, m, (bn, relu, pad_extra)))
                pad_extra = 0
                i = j
            for ci_, (p_, m, extra) in enumerate(convs):
                last = ci_ == len(convs) - 1
                kw = dict(out=out, out_ld=out_ld, out_coff=out_coff) if last else {}
                if extra is None:
                    x, xH, xW = run_sequential(m, p_, x, xH, xW, x_ld, **kw)
                else:
                    bn, relu, pe = extra
                    x, xH, xW = compile_conv2d(m, p_, x, xH, xW, x_ld, pe, bn, relu, **kw)
                x_ld = x.shape[-1]
            return x, xH, xW

        ups_c = [int(c) for c in (b2d.model_cfg.get("NUM_UPSAMPLE_FILTERS", None) or [])]
        x, xH, xW, x_ld = self.spatial, H, W, Cenc * D
        self.bev_cat = None
        coff = 0
        for bi, blk in enumerate(b2d.blocks):
            x, xH, xW = run_sequential(blk, "backbone_2d.blocks.%d" % bi, x, xH, xW, x_ld)
            x_ld = x.shape[-1]
            if len(b2d.deblocks) > 0:
                de = b2d.deblocks[bi]
                first = de[0]
                d = first.stride[0] if isinstance(first, nn.ConvTranspose2d) else 1
                if self.bev_cat is None:
                    self.bev_cat = torch.zeros((B, xH * d, xW * d, sum(ups_c)), dtype=f32, device=dev)
                run_sequential(de, "backbone_2d.deblocks.%d" % bi, x, xH, xW, x_ld, out=self.bev_cat,
                               out_ld=sum(ups_c), out_coff=coff)
                coff += ups_c[bi]
        if self.bev_cat is None:
            self.bev_cat = x
        catH, catW, cat_ld = self.bev_cat.shape[1], self.bev_cat.shape[2], self.bev_cat.shape[3]
        self.head_first_step = len(self.steps)
        sh, sH, sW = run_sequential(head.shared_conv, "dense_head.shared_conv", self.bev_cat, catH, catW, cat_ld)
        self.head_hw = (sH, sW)
        # head outputs: one [B, L, 24] buffer per head, 4-aligned slots per branch
        slots = {"hm": 0, "center": 4, "center_z": 8, "dim": 12, "rot": 16, "vel": 20}
        self.HEAD_LD = 24
        pp = head.model_cfg.POST_PROCESSING
        K = int(pp.MAX_OBJ_PER_SAMPLE)
        nms = pp.NMS_CONFIG
        assert nms.NMS_TYPE == "nms_gpu", "only class-agnostic rotated NMS (nms_gpu) is on the CenterPoint path"
        self.heads = []
        head_branch_first = len(self.steps)
        for hi, sep in enumerate(head.heads_list):
            hbuf = torch.zeros((B, sH, sW, self.HEAD_LD), dtype=f32, device=dev)
            for name in sep.sep_head_dict:
                assert name in slots, "unsupported head branch %s" % name
                run_sequential(getattr(sep, name), "dense_head.heads_list.%d.%s" % (hi, name), sh, sH, sW, sh.shape[-1],
                               out=hbuf, out_ld=self.HEAD_LD, out_coff=slots[name])
            ncls = len(head.class_names_each_head[hi])
            has_vel = "vel" in sep.sep_head_dict
            cfg = ops.make_decode_cfg(sH, sW, ncls, K, has_vel, head.feature_map_stride, self.voxel_size[:2],
                                      self.pc_range[:2], [float(v) for v in pp.POST_CENTER_LIMIT_RANGE],
                                      pp.SCORE_THRESH, self.HEAD_LD, self.HEAD_LD)
            nb = 9 if has_vel else 7
            L_ = ops._lib.load()
            import ctypes
            rec = dict(buf=hbuf, cfg=cfg, slots=slots, has_vel=has_vel, nb=nb, K=K,
                       thresh=float(nms.NMS_THRESH), pre=int(nms.NMS_PRE_MAXSIZE), post=int(nms.NMS_POST_MAXSIZE),
                       class_map=head.class_id_mapping_each_head[hi].to(dev).to(i32).contiguous(),
                       cand_boxes=torch.zeros((B, K, nb), dtype=f32, device=dev),
                       cand_scores=torch.zeros((B, K), dtype=f32, device=dev),
                       cand_labels=torch.zeros((B, K), dtype=i32, device=dev),
                       cand_count=torch.zeros((B, 1), dtype=i32, device=dev),
                       keep=torch.zeros((B, K), dtype=i32, device=dev),
                       out_boxes=torch.zeros((self.NSLOT, B, K, nb), dtype=f32, device=dev),
                       out_scores=torch.zeros((self.NSLOT, B, K), dtype=f32, device=dev),
                       out_labels=torch.zeros((self.NSLOT, B, K), dtype=torch.int64, device=dev),
                       # (one workspace per sample: the (sample, head) pairs of a batch share launches)
                       dec_ws=torch.empty((B, (max(L_.aq3d_center_decode_workspace_bytes(ctypes.byref(cfg)), 256) + 255) // 256 * 256),
                                          dtype=torch.uint8, device=dev),
                       nms_ws=torch.empty((B, (max(L_.aq3d_nms_workspace_bytes(K), 256) + 255) // 256 * 256), dtype=torch.uint8, device=dev))
            self.heads.append(rec)
        self.det_counts = torch.zeros((self.NSLOT, B, len(self.heads)), dtype=i32, device=dev)
        self._group_head_convs(head_branch_first, B)
        self.steps.append(("decode_nms", None))

    def _group_sparse_head(self, first):
        """VoxelNeXtHead: 6 heads x 6 branches x 2 small sparse convs on ONE index level. Reorder by depth and launch runs of
        equal-shape layers together (aq3d_spconv_group, <= 8 per launch): 72 launches become 10."""
        if os.environ.get("AQ3D_NO_GROUPS") is not None:
            return
        tail = self.steps[first:]
        if not tail or any(k != "spconv" for k, _ in tail):
            return
        recs = [r for _, r in tail]
        if any(r["quant"] is not None or r["residual"] is not None for r in recs):
            return                                   # activation-quantised layers go through the single-layer f16-split path
        by_out = {id(r["out"]): r for r in recs}
        depth = {}

        def depth_of(r):
            if id(r) not in depth:
                prod = by_out.get(id(r["x"]))
                depth[id(r)] = 0 if prod is None else depth_of(prod) + 1
            return depth[id(r)]

        def sig(r):
            return (depth_of(r), r["cin"], r["cout"], r["kvol"], r["level"], str(r["nbr"]), r["relu"], str(r["weight"].dtype),
                    r["w_dequant"] is not None)

        order = sorted(range(len(recs)), key=lambda i: (sig(recs[i]), i))
        new_steps, i = [], 0
        while i < len(order):
            j = i
            while j + 1 < len(order) and j + 1 - i < 8 and sig(recs[order[j + 1]]) == sig(recs[order[i]]):
                j += 1
            grp = [recs[k] for k in order[i:j + 1]]
            new_steps.append(("spconv", grp[0]) if len(grp) == 1 else ("spconv_group", dict(recs=grp, args=None)))
            i = j + 1
        self.steps[first:] = new_steps

    def _run_spconv_group(self, G):
        recs = G["recs"]
        r0 = recs[0]
        lv = self.levels[r0["level"]]
        if G["args"] is None:
            nbr = self._nbr_of(r0["nbr"], r0["level"])
            # the per-channel abs-max of an output is only produced for tensors some activation quantiser (or calibrator) reads: the
            # head's output convs feed the decode, and thousands of waves updating the same few words serialise
            read = self._amax_readers()
            G["args"] = ops.spconv_problems([dict(
                feats_in=r["x"]["buf"], weight=r["weight"], w_dequant=r["w_dequant"], bias=r["bias"], post_scale=r["post_scale"],
                post_shift=r["post_shift"], nbr=nbr, count_out=lv["count"], feats_out=r["out"]["buf"],
                absmax_out=self.arena.view(r["out"]["amax"]) if id(r["out"]) in read else None, cout_real=r["cout_real"]) for r in recs])
            G["w_mode"] = ops.W_I8 if r0["weight"].dtype == torch.int8 else (ops.W_F16S if r0["weight"].dtype == torch.float16 else ops.W_F32)
        with _Timed(self, "spconv_c%d" % r0["cout"]):
            ops.spconv_group(G["args"], ops.ACT_F32, G["w_mode"], lv["cap"], self.levels[r0["x"]["level"]]["cap"], r0["cin"], r0["cout"],
                             r0["kvol"], r0["relu"])

    def _group_head_convs(self, first, B):
        """The SeparateHead branches are independent chains of small convs (center_head.py:12-46): reorder them by depth
        (all first convs, then all second convs ...) and turn runs of same-class convs into ONE grouped launch each
        (aq3d_conv2d_*_group, aq3d_sq_prep_group; <= 8 problems per launch). The ctypes argument arrays are built here, once."""
        if os.environ.get("AQ3D_NO_GROUPS") is not None:
            return
        tail = self.steps[first:]
        if not tail or any(k != "conv2d" for k, _ in tail):
            return
        recs = [r for _, r in tail]
        by_out = {r["out"].data_ptr(): r for r in recs}
        depth = {}

        def depth_of(r):
            if id(r) not in depth:
                prod = by_out.get(r["x"].data_ptr())
                depth[id(r)] = 0 if prod is None or prod is r else depth_of(prod) + 1
            return depth[id(r)]

        def sig(r):
            g = r["geom"]
            base = (depth_of(r), r["kind"], g.H, g.W, g.Cin, g.ksize, g.stride, g.pad, g.deconv)
            if r["kind"] == "i8":
                return base + (g.Cout >= 128, r["w_bits"], r["a_bits"])
            rows = g.Cout * (g.deconv * g.deconv if g.deconv else 1)
            L = g.H * g.W if g.deconv else g.Ho * g.Wo
            return base + (0 if rows <= 16 else (1 if (rows <= 64 or L < 128 * 150) else 2), r.get("calib") is not None)

        order = sorted(range(len(recs)), key=lambda i: (depth_of(recs[i]), sig(recs[i]), i))
        new_steps, i = [], 0
        while i < len(order):
            j = i
            while j + 1 < len(order) and j + 1 - i < 8 and sig(recs[order[j + 1]]) == sig(recs[order[i]]):
                j += 1
            grp = [recs[k] for k in order[i:j + 1]]
            if len(grp) == 1 or grp[0].get("calib") is not None or grp[0]["kind"] == "sq16":
                new_steps.extend(("conv2d", r) for r in grp)
            else:
                new_steps.append(("conv2d_group", self._make_group(grp, B)))
            i = j + 1
        self.steps[first:] = new_steps

    def _make_group(self, grp, B):
        kind = grp[0]["kind"]
        G = dict(kind=kind, recs=grp)

        def problems(b):
            out = []
            for r in grp:
                po = None
                if r["pieces_out"] is not None:
                    po = self.arena.view(r["pieces_out"][0], (16, r["pieces_out"][1]))
                out.append(dict(x=r["x"][b], w=r["params"] if kind == "i8" else r["weight"], g=r["geom"], bias=r["bias"],
                                post_scale=r["post_scale"], post_shift=r["post_shift"], y=r["out"][b], pieces_out=po,
                                pieces_mode=r["pieces_out"][2] if r["pieces_out"] is not None else 0))
            return out
        G["problems"] = problems        # resolved after the arena exists (first run)
        G["conv_args"] = None
        G["sq_args"] = None
        return G

    def _run_conv2d_group(self, G, B):
        recs, kind = G["recs"], G["kind"]
        if G["conv_args"] is None:       # pointers are stable from here on: build the ctypes arrays once
            G["conv_args"] = {}
            if kind == "i8":
                G["sq_args"] = ops.sq_prep_problems([dict(
                    pieces=self.arena.view(r["pieces"], (16, r["geom"].Cin)), weight=r["weight"], w_colmax=r["w_colmax"], g=r["geom"],
                    alpha=r["alpha"], static_x=r["static_x"], static_w=r["static_w"], amax_x_acc=self.arena.view(r["amax_x_acc"]),
                    params=r["params"]) for r in recs])
        if kind == "i8":
            for r in recs:
                if r["pieces_owner"]:
                    g = r["geom"]
                    with _Timed(self, "piece_absmax"):
                        for b in range(B):
                            ops.piece_absmax(r["x"][b], g.H, g.W, g.Cin, g.in_ld, self.arena.view(r["pieces"], (16, g.Cin)))
            with _Timed(self, "sq_prep"):
                ops.sq_prep_group(G["sq_args"], recs[0]["w_bits"], recs[0]["a_bits"])
            with _Timed(self, "conv2d_i8"):
                for args in self._batched_problems(G["conv_args"], G["problems"], B):
                    ops.conv2d_i8_group(args)
        else:
            with _Timed(self, "conv2d_f32"):
                for args in self._batched_problems(G["conv_args"], G["problems"], B):
                    ops.conv2d_f32_group(args)

    @staticmethod
    def _batched_problems(cache, problems_of, B):
        """The (conv, sample) problems of one plan step for a batch of B samples as ctypes arrays of <= 8 problems — one launch each
        (a launch's problems share the kernel class; the samples of a batch are just more problems: B images per launch instead of B
        launches). Built once per batch size."""
        if B not in cache:
            flat = [p for b in range(B) for p in problems_of(b)]
            cache[B] = [ops.conv_problems(flat[i:i + 8]) for i in range(0, len(flat), 8)]
        return cache[B]

    # ------------------------------------------------------------------ execution
    def _voxel_cfg(self, stride, has_b):
        return ops.make_voxel_cfg(self.pc_range, self.voxel_size, self.grid_size, self.num_point_features, stride, has_b,
                                  self.max_points_per_voxel, self.max_voxels)

    def _voxelize(self, batch_dict, B, exact_limit=False):
        """Level 0 of the frame: active voxels, their rows and mean features (+ the level's gather tables).
        Raw points (the usual case): hard voxelization on the device straight into the index level. Pre-voxelized batch_dict
        (pcdet contract: voxels [M,P,C], voxel_num_points [M], voxel_coords [M,4]): MeanVFE + the level from the coordinates.
        exact_limit: the frame has more than MAX_NUMBER_OF_VOXELS voxels — reproduce the reference's first-seen truncation with
        the order-preserving voxelizer (aq3d_voxelize), then proceed as for pre-voxelized input."""
        lv0 = self.levels[0]
        ix = lv0["index"]
        if "voxels" in batch_dict and "voxel_coords" in batch_dict and not getattr(self.det, "voxelize_on_device", False):
            vox = batch_dict["voxels"].to(self.device, f32).contiguous()
            m = vox.shape[0]
            if m > self.cap:
                raise ValueError("batch has %d voxels, engine capacity is %d" % (m, self.cap))
            coords = batch_dict["voxel_coords"].to(self.device).to(i32).contiguous()
            self._need_index_ws(m)
            rows = ops.level_from_coords(coords, ix, workspace=self.index_ws)
            if m > 0:
                ops.mean_vfe(vox, batch_dict["voxel_num_points"].to(self.device), rows=rows, out=self.vfe_feats)
            self._level_tables(0)
            return
        pts = batch_dict["points"]
        assert pts.is_cuda, "points must be on the device (load_data_to_gpu)"
        stride = pts.shape[1]
        has_b = stride == self.num_point_features + 1
        cfg = self._voxel_cfg(stride, has_b)
        if not exact_limit:
            self._need_index_ws(pts.shape[0])
            ops.level_from_points(pts, cfg, ix, feats_out=self.vfe_feats, npts_out=self.vfe_npts, workspace=self.index_ws)
            self._level_tables(0)
            return
        # first-seen order + MAX_NUMBER_OF_VOXELS per sample, sample by sample (data_processor.py:45-61 runs per sample)
        tmp = (torch.zeros((self.cap, 4), dtype=i32, device=self.device), torch.zeros((self.cap, self.num_point_features), dtype=f32, device=self.device),
               torch.zeros((self.cap,), dtype=i32, device=self.device), torch.zeros((1,), dtype=i32, device=self.device))
        if has_b and B > 1:
            offs = batch_dict.get("points_batch_offsets", None)
            if offs is None:
                cnt = torch.bincount(pts[:, 0].to(torch.int64), minlength=B).cpu().tolist()
                offs = np.concatenate([[0], np.cumsum(cnt)]).tolist()
            for b in range(B):
                ops.voxelize(pts[offs[b]:offs[b + 1]], cfg, self.cap, batch_index=b, append=b > 0, out=tmp)
        else:
            ops.voxelize(pts, cfg, self.cap, batch_index=0, append=False, out=tmp)
        m = int(tmp[3].item())
        self._need_index_ws(m)
        rows = ops.level_from_coords(tmp[0][:m].contiguous(), ix, workspace=self.index_ws)
        if m > 0:      # rows of the level <- the voxel means (x / 1 is exact)
            ops.mean_vfe(tmp[1][:m].contiguous().view(m, 1, -1), torch.ones((m,), dtype=f32, device=self.device), rows=rows, out=self.vfe_feats)
        self._level_tables(0)

    def _need_index_ws(self, n_points):
        if n_points > self._index_pts:
            self._alloc_index_ws((int(n_points) + 16383) // 16384 * 16384)
            self._graphs = {}

    def _level_tables(self, li):
        """Every gather table that hangs on level li (submanifold tables, the strided conv that produces the level, head kernels)
        in one launch per four tables; the job arrays are built once."""
        lv = self.levels[li]
        if lv["jobs"] is None:
            jobs = []
            ix = lv["index"]
            d = lv.get("down")
            if d is not None:
                jobs.append((self.levels[d["src"]]["index"], d["ksize"], d["stride"], d["pad"], d["nbr"]))
            if ("index_subm", li) in [(k_, a_) for k_, a_ in self.steps if k_ == "index_subm"]:
                ks = lv["subm_ksize"]
                assert lv["nbr"].shape[1] == int(np.prod(ks))
                jobs.append((ix, ks, (1, 1, 1), tuple(k // 2 for k in ks), lv["nbr"]))
            for k, nbr in lv.get("extra_nbr", {}).items():
                jobs.append((ix, (1, k, k), (1, 1, 1), (0, k // 2, k // 2), nbr))
            lv["jobs"] = ops.nbr_jobs(jobs) if jobs else ()
        if len(lv["jobs"]):
            ops.level_neighbors(lv["index"], lv["jobs"])

    def start_timing(self, categories=None):
        self.timers, self.timer_filter = {}, (set(categories) if categories else None)

    def stop_timing(self):
        """→ {category: (n_launch_groups, total_ms)}; synchronises."""
        torch.cuda.synchronize(self.device)
        out = {k: (len(v), float(sum(a.elapsed_time(b) for a, b in v))) for k, v in (self.timers or {}).items()}
        self.timers = None
        return out

    def _nbr_of(self, ref, level=None):
        if isinstance(ref, tuple):
            if ref[0] == "head":
                return self.levels[level]["extra_nbr"][ref[1]]
            return self.levels[ref[1]]["down"]["nbr"]
        return self.levels[ref]["nbr"]

    def _plan_static_sparse(self):
        """Static PTQ (quant_centerpoint.py:127-129: calibrated amax) for the coded sparse layers: their quantiser scales are
        constants, so (i) the weight copy W' is built ONCE, here, instead of per frame, (ii) the quantiser of a layer runs in the
        epilogue of the conv that produces its input (next_scale / next_codes of aq3d_spconv_coded and of the stem layer) — the
        20 quantiser launches of a frame disappear — and (iii) fp32 rows nobody else reads (the conv1 output of a residual
        block) are not stored at all. Layers whose producer cannot carry the hook keep a rows-only quantiser launch."""
        recs = self.sparse_convs
        static = [r for r in recs if r.get("coded") and r["quant"]["on"] and r["quant"]["static_amax"] is not None]
        if not static:
            return
        keep_f32 = bool(getattr(self.det, "static_keep_f32", False))
        by_out = {id(r["out"]): r for r in recs}
        users = {}
        for r in recs:
            users.setdefault(id(r["x"]), []).append(r)
            if r["residual"] is not None:
                users.setdefault(id(r["residual"]), []).append(r)
        pinned = {id(t) for t, _ in self.multi_scale.values()} | {id(self.encoded[0])}
        if getattr(self, "merge", None):
            pinned |= {id(x) for x, _, _ in self.merge["srcs"]}
        eps = np.float32(1.0 / (1 << 24))
        for r in static:
            q = r["quant"]
            amax = _np(q["static_amax"]).reshape(-1).astype(np.float32)
            if q["per_tensor"] or amax.size == 1:
                amax = np.full((r["cin"],), amax.max(), np.float32)
            qmax = np.float32((1 << (q["bits"] - 1)) - 1)
            with np.errstate(divide="ignore"):
                scale = np.where(amax <= eps, np.float32(0), qmax / amax).astype(np.float32)
            r["static"] = dict(amax=self._dev(amax), scale=self._dev(scale), fused=False)
            # (i) weight copy once
            ops.spconv_prep(None, r["static"]["amax"], None, r["weight_i8"], r["w_dequant"], bits=q["bits"], per_tensor=False,
                            out=(None, r["wprep"], r["out_mul"]))
            # (ii) quantiser into the producer's epilogue
            p = by_out.get(id(r["x"]))
            if p is None or "next" in p:
                continue
            stem = p["quant"] is None and p["w_dequant"] is None and p["cin"] <= 8 and p["cout"] % 4 == 0 and p["cout"] <= 32
            if not (p.get("coded") or stem) or p["cout"] != r["cin"]:
                continue
            p["next"] = dict(scale=r["static"]["scale"], codes=r["codes"], bits=q["bits"])
            r["static"]["fused"] = True
            # (iii) nobody but this quantiser reads the producer's fp32 rows
            others = [u for u in users.get(id(p["out"]), []) if u is not r]
            p["skip_f32"] = bool(p.get("coded")) and not others and id(p["out"]) not in pinned and not keep_f32

    def _amax_readers(self):
        """ids of the sparse tensors whose per-channel abs-max some activation quantiser / calibrator reads."""
        if getattr(self, "_amax_read", None) is None or self._amax_read[0] != len(self.sparse_convs):
            self._amax_read = (len(self.sparse_convs), {id(c["x"]) for c in self.sparse_convs if c["quant"] is not None})
        return self._amax_read[1]

    def _amax_out(self, out):
        return self.arena.view(out["amax"]) if id(out) in self._amax_readers() else None

    def _run_spconv(self, rec):
        lv = self.levels[rec["level"]]
        x = rec["x"]
        nbr = self._nbr_of(rec["nbr"], rec["level"])
        in_count = self.levels[x["level"]]["count"]
        feats, act_dq = x["buf"], None
        q = rec["quant"]
        if q is not None and q["calib"]:
            amax = self.arena.view(x["amax"])
            q["quantizer"].collect(amax.max().reshape(1) if q["per_tensor"] else amax)
        if q is not None and q["on"]:
            amax = q["static_amax"] if q["static_amax"] is not None else self.arena.view(x["amax"])
            if q["static_amax"] is not None and q["static_amax"].numel() == 1 and x["C"] > 1:
                amax = q["static_amax"].expand(x["C"]).contiguous()
            if rec.get("coded"):
                st = rec.get("static")
                if st is None:
                    with _Timed(self, "quantize_rows"):      # quantiser + per-frame weight copy, one launch
                        ops.spconv_prep(x["buf"], amax, in_count, rec["weight_i8"], rec["w_dequant"], bits=q["bits"],
                                        per_tensor=q["per_tensor"], out=(rec["codes"], rec["wprep"], rec["out_mul"]))
                elif not st["fused"] or self._force_rows:
                    with _Timed(self, "quantize_rows"):      # static, producer without the hook: rows only (W' is constant)
                        ops.spconv_prep(x["buf"], st["amax"], in_count, None, None, bits=q["bits"], per_tensor=False,
                                        out=(rec["codes"], None, None))
                out = rec["out"]
                nx = rec.get("next")
                with _Timed(self, "spconv_c%d" % rec["cout"]):
                    ops.spconv_coded(rec["codes"], rec["wprep"], rec["out_mul"], nbr, lv["count"], bias=rec["bias"],
                                     post_scale=rec["post_scale"], post_shift=rec["post_shift"],
                                     residual=None if rec["residual"] is None else rec["residual"]["buf"], relu=rec["relu"],
                                     absmax_out=self._amax_out(out), out=out["buf"],
                                     next_scale=None if nx is None else nx["scale"], next_codes=None if nx is None else nx["codes"],
                                     next_bits=8 if nx is None else nx["bits"], skip_f32=bool(rec.get("skip_f32")),
                                     w16=self.sparse_w16)
                return
            L = ops._lib.load()
            with _Timed(self, "quantize_rows"):
                ops.check(L.aq3d_quantize_rows(ops._p(x["buf"]), ops._p(amax), ops._p(in_count), int(x["buf"].shape[0]), x["C"], q["bits"],
                                               int(q["per_tensor"]), None, ops._p(rec["fq"]), ops._p(rec["dq"]), ops._stream()),
                          "aq3d_quantize_rows")
            feats = rec["fq"]
        out = rec["out"]
        nx = rec.get("next")
        with _Timed(self, "spconv_c%d" % rec["cout"]):
            ops.spconv(feats, rec["weight"], nbr, lv["count"], rec["cout"], act_dequant=act_dq, w_dequant=rec["w_dequant"],
                       bias=rec["bias"], post_scale=rec["post_scale"], post_shift=rec["post_shift"],
                       residual=None if rec["residual"] is None else rec["residual"]["buf"], relu=rec["relu"],
                       absmax_out=self._amax_out(out), out=out["buf"],
                       next_scale=None if nx is None else nx["scale"], next_codes=None if nx is None else nx["codes"],
                       next_bits=8 if nx is None else nx["bits"])

    def _run_conv2d(self, rec, B):
        g = rec["geom"]
        x, out = rec["x"], rec["out"]
        po = None
        if rec["pieces_out"] is not None:
            po = self.arena.view(rec["pieces_out"][0], (16, rec["pieces_out"][1]))
        if rec["kind"] in ("i8", "sq16"):
            pieces = self.arena.view(rec["pieces"], (16, g.Cin))
            if rec["pieces_owner"]:
                with _Timed(self, "piece_absmax"):
                    for b in range(B):
                        ops.piece_absmax(x[b], g.H, g.W, g.Cin, g.in_ld, pieces)
            with _Timed(self, "sq_prep"):
                ops.sq_prep(pieces, rec["weight"], rec["w_colmax"], g, rec["alpha"], rec["params"], rec["w_bits"],
                            rec["a_bits"], rec["static_x"], rec["static_w"], self.arena.view(rec["amax_x_acc"]))
            if rec["kind"] == "sq16":
                with _Timed(self, "conv2d_sq16"):
                    for b in range(B):
                        ops.conv2d_sq16(x[b], rec["params"], g, rec["w_bits"], rec["a_bits"], rec["bias"], rec["post_scale"],
                                        rec["post_shift"], out=out[b], workspace=rec["sq16_ws"])
                return
            with _Timed(self, "conv2d_i8"):
                if B > 1:
                    for args in self._batched_problems(rec.setdefault("_batch_args", {}), lambda b: [dict(
                            x=x[b], w=rec["params"], g=g, bias=rec["bias"], post_scale=rec["post_scale"], post_shift=rec["post_shift"],
                            y=out[b], pieces_out=po, pieces_mode=rec["pieces_out"][2] if po is not None else 0)], B):
                        ops.conv2d_i8_group(args)
                else:
                    ops.conv2d_i8(x[0], rec["params"], g, rec["bias"], rec["post_scale"], rec["post_shift"], out=out[0],
                                  pieces_out=po, pieces_mode=rec["pieces_out"][2] if po is not None else 0)
        else:
            cal = rec.get("calib")
            if cal is not None:
                self._harvest_calib(rec, cal, B)
            with _Timed(self, "conv2d_f32"):
                if B > 1 and cal is None:
                    for args in self._batched_problems(rec.setdefault("_batch_args", {}), lambda b: [dict(
                            x=x[b], w=rec["weight"], g=g, bias=rec["bias"], post_scale=rec["post_scale"], post_shift=rec["post_shift"],
                            y=out[b], pieces_out=po, pieces_mode=rec["pieces_out"][2] if po is not None else 0,
                            w_split=rec.get("w_split"))], B):
                        ops.conv2d_f32_group(args)
                else:
                    for b in range(B):
                        ops.conv2d_f32(x[b], rec["weight"], g, rec["bias"], rec["post_scale"], rec["post_shift"], out=out[b],
                                       pieces_out=po, weight_split=rec.get("w_split"),
                                       pieces_mode=rec["pieces_out"][2] if po is not None else 0)

    def _harvest_calib(self, rec, cal, B):
        """Calibration forward of an SQConv2d/QuantConv2d: the amax its two quantizers would see this batch."""
        g = rec["geom"]
        pieces = self.arena.view(cal["pieces"], (16, g.Cin))
        for b in range(B):
            ops.piece_absmax(rec["x"][b], g.H, g.W, g.Cin, g.in_ld, pieces)
        cout4 = (cal["cout"] + 3) // 4 * 4
        g2 = ops.conv_geom(g.H, g.W, g.Cin, cout4, g.ksize, g.stride, g.pad, in_ld=g.in_ld)
        wk = rec["weight"]
        if wk.shape[0] < cout4:
            wk = torch.cat([wk, torch.zeros((cout4 - wk.shape[0], wk.shape[1]), device=wk.device)], 0).contiguous()
        ops.sq_prep(pieces, wk, cal["w_colmax"], g2, cal["alpha"], cal["params"])
        v = ops.sq_param_views(cal["params"], g.Cin, cout4, g.ksize)
        scal, amax_w = v["scal"], v["amax_w"][:cal["cout"]]
        wq_, iq_ = cal["quantizers"]
        if iq_._if_calib:
            iq_.collect(scal[1:2])
        if wq_._if_calib:
            wq_.collect(amax_w)

    def _head_batches(self, B, make):
        """ctypes problem arrays of aq3d_decode_nms_batch for result slot self.slot and a batch of B samples: <= 8 (sample, head)
        pairs per launch, built once."""
        key = (self.slot, B)
        if key not in self._head_args:
            recs = [make(hi, h, b) for b in range(B) for hi, h in enumerate(self.heads)]
            self._head_args[key] = [(ops.head_problems(recs[i:i + 8]), recs[i:i + 8]) for i in range(0, len(recs), 8)]
        return self._head_args[key]

    def _head_common(self, hi, h, b):
        cap_nms = min(h["K"], h["pre"])
        return dict(cfg=h["cfg"], cand=(h["cand_boxes"][b], h["cand_scores"][b], h["cand_labels"][b], h["cand_count"][b]),
                    dec_ws=h["dec_ws"][b], thresh=h["thresh"], nms_cap=cap_nms, post_max=h["post"], keep=h["keep"][b],
                    keep_count=self.det_counts[self.slot, b, hi:hi + 1], nms_ws=h["nms_ws"][b], class_map=h["class_map"],
                    out=(h["out_boxes"][self.slot, b], h["out_scores"][self.slot, b], h["out_labels"][self.slot, b]))

    def _decode_nms(self, B):
        """center_head.py:297-365 for every head: top-K, decode, rotated NMS, gather — five launches for up to 8 heads."""
        def make(hi, h, b):
            s = h["slots"]
            buf = h["buf"][b].view(-1, self.HEAD_LD)
            r = self._head_common(hi, h, b)
            r.update(hm=buf[:, s["hm"]:], center=buf[:, s["center"]:], center_z=buf[:, s["center_z"]:], dim=buf[:, s["dim"]:],
                     rot=buf[:, s["rot"]:], vel=buf[:, s["vel"]:] if h["has_vel"] else None)
            return r
        for args, _ in self._head_batches(B, make):
            ops.decode_nms_batch(args)

    def run_head_from(self, bev_nhwc):
        """Run only shared conv → head branches → decode → NMS on given BEV features [B, H, W, C] (parity tests feed
        the oracle's features here so that box-level comparisons are not swamped by upstream rounding cascades)."""
        B = bev_nhwc.shape[0]
        self.arena.clear()
        self.bev_cat[:B].copy_(bev_nhwc)
        # the piece maxima of bev_cat normally come out of the deblock epilogues, which do not run here
        for kind, rec in self.steps[self.head_first_step:]:
            if kind == "conv2d" and rec["kind"] in ("i8", "sq16") and rec["x"] is self.bev_cat and not rec["pieces_owner"]:
                g = rec["geom"]
                for b in range(B):
                    ops.piece_absmax(self.bev_cat[b], g.H, g.W, g.Cin, g.in_ld, self.arena.view(rec["pieces"], (16, g.Cin)))
                break
        self._run_steps(self.steps[self.head_first_step:], B)
        self._publish()
        return B

    def run_sparse_head_from(self, rows):
        """VoxelNeXt: run only the sparse head branches -> decode -> NMS on given rows of the shared_conv output (engine row order,
        f32 [M, C]); the frame's index levels must exist (run the frame once). Parity tests feed the oracle's features here."""
        assert self.fully_sparse
        x, lvl = self.encoded
        m = int(self.levels[lvl]["count"].item())
        assert rows.shape[0] == m and rows.shape[1] <= x["buf"].shape[1]
        self.arena.clear()
        x["buf"][:m].zero_()
        x["buf"][:m, :rows.shape[1]].copy_(rows)
        self._run_steps(self.steps[self.head_first_step:], 1)
        self._publish()
        return 1

    def run(self, batch_dict):
        """Enqueue one batch on the current stream. Results sit in self.heads[*]['out_*'] and self.det_counts (slot self.slot);
        Detector.forward_async calls this on the plan's own stream."""
        B = self.begin(batch_dict)
        self._slot_batch[self.slot] = batch_dict
        if self._graph_now:
            batch_dict = self._stage_points(batch_dict)
            self._static_batch = batch_dict
        self._phase(("all", self.slot, B), lambda: self._enqueue(batch_dict, B))
        self._publish()
        return B

    def begin(self, batch_dict):
        """Per-frame bookkeeping: picks the result slot and whether this frame replays its captured hipGraph (single-sample
        frames given as raw points: one host call instead of ~180 launches — the eager path is host-bound at ~18 us per Python
        launch) or is enqueued eagerly (timers, calibration, batches, pre-voxelized input)."""
        B = int(batch_dict["batch_size"])
        if B > self.batch_cap:
            raise ValueError("batch_size %d exceeds the engine's max_batch_size %d" % (B, self.batch_cap))
        self.slot = (self.slot + 1) % self.NSLOT
        self.slot_generation[self.slot] += 1        # PendingFrame handles of the slot's previous frame are stale from here on
        self._graph_now = self._graph_ok(batch_dict, B)
        self._warm_now = self._graph_now and self._warm < 1     # first frame: eager, settles workspaces / kernel attributes
        if self._warm_now:
            self._warm += 1
        return B

    def _phase(self, key, enqueue):
        if self._graph_now and not self._warm_now and self.use_graph:
            g = self._graphs.get(key)
            if g is None:
                try:
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    cur = torch.cuda.current_stream(self.device)
                    cap = None if cur == torch.cuda.default_stream(self.device) else cur     # the legacy default stream cannot capture
                    with torch.cuda.graph(g, stream=cap):
                        enqueue()
                    self._graphs[key] = g
                except Exception as e:      # CAPTURE is an optimisation: fall back to eager launches, once, loudly
                    import warnings
                    warnings.warn("hipGraph capture failed (%r); this engine continues with eager launches" % (e,))
                    self.use_graph = False
                    self._graphs = {}
                    g = None
            if g is not None:
                g.replay()                  # an error of the replay itself propagates
                return
        enqueue()

    def _enqueue(self, batch_dict, B, exact_limit=False):
        self.arena.clear()
        with _Timed(self, "voxelize"):
            self._voxelize(batch_dict, B, exact_limit)
        self._run_steps(self.steps, B)

    # ------------------------------------------------------------------ hipGraph replay
    def _graph_ok(self, batch_dict, B):
        if not self.use_graph or self.timers is not None or "points" not in batch_dict:
            return False
        if self._calibrating is None:   # calibration forwards harvest statistics on the host between launches: eager only
            self._calibrating = any(r["quant"] is not None and r["quant"]["calib"] for r in self.sparse_convs) or \
                any(r.get("calib") is not None for r in self.dense_convs)
        if self._calibrating:
            return False
        if "voxels" in batch_dict and not getattr(self.det, "voxelize_on_device", False):
            return False
        pts = batch_dict["points"]
        if B > 1 and pts.shape[1] != self.num_point_features + 1:      # a batch needs the sample index column (collate_batch)
            return False
        return pts.is_cuda and pts.dtype == f32 and pts.dim() == 2

    def _stage_points(self, batch_dict):
        """Static input buffer of the captured graphs (capacity rounded up, tail rows parked outside the point-cloud range so
        the voxelizer drops them — first-touch order and therefore every index stays bit-identical to the eager path)."""
        pts = batch_dict["points"]
        n, stride = int(pts.shape[0]), int(pts.shape[1])
        buf = self._pts_buf
        if buf is None or buf.shape[1] != stride or n > buf.shape[0]:
            cap_pts = max(16384, (n + 16383) // 16384 * 16384)
            buf = torch.empty((cap_pts, stride), dtype=f32, device=self.device)
            far = torch.tensor([0.0] * stride, dtype=f32)
            off = 1 if stride == self.num_point_features + 1 else 0
            for a in range(3):
                far[off + a] = float(self.pc_range[3 + a]) + 1000.0
            buf[:] = far.to(self.device)
            self._pts_far = far.to(self.device)
            self._pts_buf, self._pts_n, self._graphs = buf, 0, {}
            self._need_index_ws(cap_pts)            # allocate outside the capture that follows
        buf[:n].copy_(pts)
        if self._pts_n > n:
            buf[n:self._pts_n] = self._pts_far
        self._pts_n = n
        return {"points": buf, "batch_size": int(batch_dict["batch_size"])}

    def _publish(self):
        """Detection counts of this frame → pinned host memory, stream-ordered, plus an event: collect() waits for THIS frame
        only, so the caller may already have enqueued the next one (forward_async)."""
        self.host_counts[self.slot].copy_(self.det_counts[self.slot], non_blocking=True)
        self.host_stats[self.slot].copy_(self._stats_pool, non_blocking=True)       # every level's site totals (overflow check)
        self._ev_done[self.slot].record()

    INDEX_KINDS = ("index_subm", "index_down", "index_head")

    def _run_steps(self, steps, B):
        """Walks the plan in order on the current stream."""
        for si, (kind, arg) in enumerate(steps):
            if kind in self.INDEX_KINDS:
                self._run_index(kind, arg)
                continue
            if kind == "spconv":
                self._run_spconv(arg)
            elif kind == "conv2d":
                self._run_conv2d(arg, B)
            elif kind == "conv2d_group":
                self._run_conv2d_group(arg, B)
            elif kind == "spconv_group":
                self._run_spconv_group(arg)
            elif kind == "absmax":
                if any(r["quant"] is not None and (r["quant"]["on"] or r["quant"]["calib"]) and r["x"] is arg
                       for r in self.sparse_convs):
                    ops.absmax_cols(arg["buf"], self.levels[arg["level"]]["count"], self.arena.view(arg["amax"]))
            elif kind == "scatter_dense":
                x, lvl = self.encoded
                lv = self.levels[lvl]
                with _Timed(self, "scatter_dense"):
                    ops.scatter_dense(x["buf"], lv["coords"], lv["count"], self.bev_shape, B, out=self.spatial[:B])
            elif kind == "decode_nms":
                with _Timed(self, "decode_nms"):
                    self._decode_nms(B)
            elif kind == "bev_merge":
                with _Timed(self, "index_levels"):
                    self._bev_merge()
            elif kind == "voxel_decode_nms":
                with _Timed(self, "decode_nms"):
                    self._voxel_decode_nms(B)

    def _run_index(self, kind, arg):
        # "index_subm" / "index_head" only register tables at compile time: every table of a level is built with the level
        if kind == "index_down":
            with _Timed(self, "index_levels"):
                lv_out = self.levels[arg]
                d = lv_out["down"]
                ops.level_downsample(self.levels[d["src"]]["index"], d["ksize"], d["stride"], d["pad"], lv_out["index"],
                                     workspace=self.index_ws)
                self._level_tables(arg)

    def _bev_merge(self):
        m = self.merge
        lv = self.levels[m["level"]]
        levels = [self.levels[l]["index"] for _, l, _ in m["srcs"]]
        scales = [sc for _, _, sc in m["srcs"]]
        ops.level_merge(levels, scales, lv["index"], workspace=self.index_ws)
        ops.level_merge_sum(levels, [x["buf"] for x, _, _ in m["srcs"]], scales, lv["index"], feats_out=m["out"]["buf"])

    def _voxel_decode_nms(self, B=1):
        def make(hi, h, b):
            lv = self.levels[h["level"]]
            o = h["outs"]
            r = self._head_common(hi, h, b)
            r.update(hm=o["hm"]["buf"], center=o["center"]["buf"], center_z=o["center_z"]["buf"], dim=o["dim"]["buf"], rot=o["rot"]["buf"],
                     vel=o["vel"]["buf"] if h["has_vel"] else None, coords=lv["coords"], count=lv["count"],
                     sample=b if self.batch_cap > 1 else -1)      # the rows of sample b: the head level is raster-ordered (sorted by sample)
            return r
        assert B == 1 or not self.levels[self.heads[0]["level"]]["index"].sorted_rows, "a batch needs the head level in raster order"
        for args, _ in self._head_batches(B, make):
            ops.decode_nms_batch(args)

    # ------------------------------------------------------------------ results
    def counts(self, B, slot=None):
        """Per-head detection counts of the frame in `slot` (default: the latest) — waits for that frame only."""
        slot = self.slot if slot is None else slot
        self._ev_done[slot].synchronize()
        self._check_levels(slot, B)
        return self.host_counts[slot, :B].numpy().copy()

    def _check_levels(self, slot, B):
        """A level that outgrew its buffers invalidates the frame: raise (the kernels clamp, they never write out of bounds).
        More voxels than MAX_NUMBER_OF_VOXELS in a sample: the reference keeps the first-seen ones — redo the frame that way."""
        st = self.host_stats[slot].numpy()
        bd = self._slot_batch[slot]
        if int(st[0, 2]) > self.max_voxels and bd is not None and "points" in bd and "voxels" not in bd:
            self._slot_batch[slot] = None
            return self._rerun_first_seen(slot, bd, B)
        for li, lv in enumerate(self.levels):
            total = int(st[li, 1])
            if total > lv["cap"]:
                what = "voxel_capacity" if li == 0 else "level_capacity[%d] / level_capacity_factor" % li
                raise LevelOverflow(li, total, lv["cap"],
                                    "index level %d: %d active sites > capacity %d rows — the frame was NOT computed; raise "
                                    "detector.%s (spconv sizes its outputs dynamically, spconv_backbone.py:230-231)"
                                    % (li, total, lv["cap"], what))

    def _rerun_first_seen(self, slot, batch_dict, B):
        """Rare path: a sample holds more than MAX_NUMBER_OF_VOXELS voxels. Everything in flight on the device finishes, then the
        frame runs again, eagerly, into its own result slot with the order-preserving voxelizer."""
        torch.cuda.synchronize(self.device)
        stream = getattr(self, "stream", None) or torch.cuda.current_stream(self.device)
        keep = self.slot
        self.slot = slot
        try:
            with torch.cuda.stream(stream):
                self._enqueue(batch_dict, B, exact_limit=True)
                self._publish()
            self._ev_done[slot].synchronize()
        finally:
            self.slot = keep
        self._check_levels(slot, B)

    def collect(self, B, slot=None):
        """One device→host read (the detection counts), then slice the padded outputs into pred_dicts."""
        slot = self.slot if slot is None else slot
        counts = self.counts(B, slot)
        preds = []
        for b in range(B):
            bx, sc, lb = [], [], []
            for hi, h in enumerate(self.heads):
                n = int(counts[b, hi])
                bx.append(h["out_boxes"][slot, b, :n])
                sc.append(h["out_scores"][slot, b, :n])
                lb.append(h["out_labels"][slot, b, :n])
            if len(self.heads) == 1:
                preds.append({"pred_boxes": bx[0].clone(), "pred_scores": sc[0].clone(), "pred_labels": lb[0].clone()})
            else:
                nbmax = max(t.shape[1] for t in bx)
                bx = [t if t.shape[1] == nbmax else torch.cat([t, t.new_zeros((t.shape[0], nbmax - t.shape[1]))], 1) for t in bx]
                preds.append({"pred_boxes": torch.cat(bx, 0), "pred_scores": torch.cat(sc, 0), "pred_labels": torch.cat(lb, 0)})
        return preds

    def sparse_view(self, name):
        """'x_conv1'..'x_conv4' or 'encoded' → SparseTensorView over the live rows (host sync on access)."""
        x, lvl = self.encoded if name == "encoded" else self.multi_scale[name]
        lv = self.levels[lvl]
        return SparseTensorView(x["buf"], lv["coords"], lv["count"], lv["shape"], self.batch_cap)
