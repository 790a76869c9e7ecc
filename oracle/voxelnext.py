"""Whole-path CPU restatement of quantized VoxelNeXt inference (ORACLE — test infrastructure; PARITY UNPINNED for the
third-party spconv arithmetic, as in oracle/sparse.py).

Composition follows pcdet/models/detectors/voxelnext.py:9-44:
    MeanVFE → VoxelResBackBone8xVoxelNeXt.forward (backbones_3d/spconv_backbone_voxelnext.py:200-228, bev_out :149-198)
            → VoxelNeXtHead.forward / generate_predicted_boxes (dense_heads/voxelnext_head.py:535-560, 418-487)
            → centernet_utils.decode_bbox_from_voxels_nuscenes (model_utils/centernet_utils.py:290-354) → class-agnostic NMS.
Quantization recipe of quant/quant_voxelnext.py:170-176,281-292,330-337:
    every 3-D sparse conv except backbone_3d.conv_input.0 → QConv3d (per-oc int8 weights, activations per input channel when
    cw else per tensor); every 2-D sparse conv outside the exclusion list (:13-71: the final '<branch>.1' convs and the whole hm
    branch) → QConv2d, which as committed only fake-quantises the WEIGHT (:160-168 never writes the activations back).
2-D sparse tensors are handled as 3-D ones of depth 1 (indices rows (b, 0, y, x), kernels (1, k, k)).
"""
import numpy as np

from . import decode as D
from . import sparse as S
from . import voxel as V

HEAD_NO_SUFFIX = ("center.1", "center_z.1", "dim.1", "rot.1", "vel.1", "hm.0.0", "hm.1")   # quant_voxelnext.py:13-71


class Recipe:
    def __init__(self, w_bits=8, act_bits=8, cw=True):
        self.w_bits, self.act_bits, self.cw = w_bits, act_bits, cw


def _bn(state, prefix):
    return {k: state["%s.%s" % (prefix, k)].float().numpy() for k in ("weight", "bias", "running_mean", "running_var")}


def _conv(state, path, x, nbr, recipe, mode, bn=None, eps=1e-3, relu=True, residual=None, trace=None):
    """mode: 'q3d' (w8a8), 'w2d' (weight-only), None (fp32). Weight [Cout, *k, Cin] of any rank."""
    w = state[path + ".weight"].float().numpy()
    b = state.get(path + ".bias", None)
    b = None if b is None else b.float().numpy()
    if recipe is not None and mode is not None:
        w, _, _ = S.qconv_weight(w, recipe.w_bits)
        if mode == "q3d":
            x, _ = S.qconv_act(x, recipe.act_bits, cw=recipe.cw)
    y = S.gather_conv(x, w, nbr, b)
    if bn is not None:
        y = S.batchnorm(y, _bn(state, bn), eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = np.maximum(y, 0)
    y = y.astype(np.float32)
    if trace is not None:
        trace[path] = y
    return y


def bev_out(levels):
    """spconv_backbone_voxelnext.py:149-198,214-221: concatenate (x_conv4, x_conv5 at x2, x_conv6 at x4), drop z, sum the
    features of coinciding (b, y, x) — torch.unique(dim=0) order (sorted) and index_add_ in concatenation order."""
    feats = np.concatenate([f for f, _, _ in levels], axis=0)
    idx = np.concatenate([np.concatenate([i[:, :1], i[:, 2:4] * s], axis=1) for _, i, s in levels], axis=0)
    uniq, inv = np.unique(idx, axis=0, return_inverse=True)
    out = np.zeros((uniq.shape[0], feats.shape[1]), dtype=np.float32)
    np.add.at(out, inv.reshape(-1), feats.astype(np.float32))
    idx4 = np.concatenate([uniq[:, :1], np.zeros_like(uniq[:, :1]), uniq[:, 1:]], axis=1).astype(np.int32)
    return out, idx4


def sparse_backbone(state, feats, coords, sparse_shape, recipe, trace=None):
    P = "backbone_3d."
    q3 = lambda path: None if path == "conv_input.0" else "q3d"

    def conv(path, x, nbr, bnp, residual=None):
        return _conv(state, P + path, x, nbr, recipe, q3(path), bn=P + bnp, eps=1e-3, relu=True, residual=residual, trace=trace)

    def block(prefix, x, nbr):
        t = conv(prefix + ".conv1", x, nbr, prefix + ".bn1")
        return conv(prefix + ".conv2", t, nbr, prefix + ".bn2", residual=x)

    idx, shape = coords, list(sparse_shape)
    nbr = S.subm_neighbors(idx, shape)
    x = conv("conv_input.0", feats, nbr, "conv_input.1")
    x = block("conv1.0", x, nbr)
    x = block("conv1.1", x, nbr)
    levels = {"x_conv1": (x, idx, shape)}
    for name in ("conv2", "conv3", "conv4", "conv5", "conv6"):
        w = state[P + name + ".0.0.weight"]
        k = tuple(int(v) for v in w.shape[1:4])
        idx2, nbr_d, shape2 = S.conv_outputs(idx, shape, k, (2, 2, 2), tuple(v // 2 for v in k))
        x = conv(name + ".0.0", x, nbr_d, name + ".0.1")
        idx, shape = idx2, shape2
        nbr = S.subm_neighbors(idx, shape)
        x = block(name + ".1", x, nbr)
        x = block(name + ".2", x, nbr)
        levels["x_" + name] = (x, idx, shape)
    f, i4 = bev_out([(levels["x_conv4"][0], levels["x_conv4"][1], 1), (levels["x_conv5"][0], levels["x_conv5"][1], 2),
                     (levels["x_conv6"][0], levels["x_conv6"][1], 4)])
    shape2d = [1] + list(levels["x_conv4"][2][1:])
    if trace is not None:
        trace["bev_out"] = (f, i4, shape2d)
    w = state[P + "conv_out.0.weight"]
    k = (1,) + tuple(int(v) for v in w.shape[1:3])
    idx_o, nbr_o, shape_o = S.conv_outputs(i4, shape2d, k, (1, 1, 1), (0, k[1] // 2, k[2] // 2))
    x = _conv(state, P + "conv_out.0", f, nbr_o, recipe, "w2d", bn=P + "conv_out.1", eps=1e-3, trace=trace)
    w = state[P + "shared_conv.0.weight"]
    k = (1,) + tuple(int(v) for v in w.shape[1:3])
    nbr_s = S.subm_neighbors(idx_o, shape_o, k)
    x = _conv(state, P + "shared_conv.0", x, nbr_s, recipe, "w2d", bn=P + "shared_conv.1", eps=1e-5, trace=trace)
    return x, idx_o, shape_o, levels


def head(state, model_cfg, x, idx, shape, recipe, trace=None):
    """VoxelNeXtHead.forward up to the raw per-branch rows (voxelnext_head.py:13-47)."""
    hc = model_cfg.DENSE_HEAD
    P = "dense_head."
    nbr_cache = {}

    def nbr_for(k):
        if k not in nbr_cache:
            nbr_cache[k] = S.subm_neighbors(idx, shape, (1, k, k))
        return nbr_cache[k]

    outs = []
    for hi, _names in enumerate(hc.CLASS_NAMES_EACH_HEAD):
        hd = dict(hc.SEPARATE_HEAD_CFG.HEAD_DICT)
        branches = list(hd.keys()) + ["hm"]
        d = {}
        for br in branches:
            n_conv = int(hc.NUM_HM_CONV) if br == "hm" else int(hd[br]["num_conv"])
            pre = "%sheads_list.%d.%s." % (P, hi, br)
            t = x
            for li in range(n_conv - 1):
                path = "%s%d.0" % (pre, li)
                k = int(state[path + ".weight"].shape[1])
                mode = None if ("%s.%d.0" % (br, li)) in HEAD_NO_SUFFIX else "w2d"
                t = _conv(state, path, t, nbr_for(k), recipe, mode, bn="%s%d.1" % (pre, li), eps=1e-5, trace=trace)
            path = "%s%d" % (pre, n_conv - 1)
            mode = None if ("%s.%d" % (br, n_conv - 1)) in HEAD_NO_SUFFIX else "w2d"
            d[br] = _conv(state, path, t, nbr_for(1), recipe, mode, relu=False, trace=trace)
        outs.append(d)
    return outs


def decode_bbox_from_voxels(indices, hm, center, center_z, dim, rot, vel, point_cloud_range, voxel_size, feature_map_stride,
                            K, score_thresh, post_center_limit_range):
    """centernet_utils.py:243-354 for one sample (nuscenes=True branch of _topk_1d). hm [N, C] post-sigmoid, dim post-exp,
    rot columns (cos, sin); indices rows (b, [z,] y, x). Requires N >= K like the reference's final .view(batch, K)."""
    N, C = hm.shape
    Keff = min(K, N)
    scores_c = np.zeros((C, Keff), np.float32)
    inds_c = np.zeros((C, Keff), np.int64)
    for c in range(C):
        o = D._stable_desc(hm[:, c])[:Keff]
        scores_c[c], inds_c[c] = hm[o, c], o
    o2 = D._stable_desc(scores_c.reshape(-1))[:min(K, C * Keff)]
    score = scores_c.reshape(-1)[o2]
    cls = (o2 // Keff).astype(np.int32)
    ind = inds_c.reshape(-1)[o2]
    f32 = np.float32
    ctr, cz, dm, rt = center[ind], center_z[ind], dim[ind], rot[ind]
    sp = indices[ind].astype(np.float32)
    angle = np.arctan2(rt[:, 1], rt[:, 0]).astype(np.float32)
    xs = (sp[:, -1] + ctr[:, 0]) * f32(feature_map_stride) * f32(voxel_size[0]) + f32(point_cloud_range[0])
    ys = (sp[:, -2] + ctr[:, 1]) * f32(feature_map_stride) * f32(voxel_size[1]) + f32(point_cloud_range[1])
    parts = [xs[:, None], ys[:, None], cz, dm, angle[:, None]]
    if vel is not None:
        parts.append(vel[ind])
    boxes = np.concatenate(parts, axis=1).astype(np.float32)
    r = np.asarray(post_center_limit_range, dtype=np.float32)
    mask = np.all(boxes[:, :3] >= r[None, :3], axis=1) & np.all(boxes[:, :3] <= r[None, 3:], axis=1)
    if score_thresh is not None:
        mask &= score > f32(score_thresh)
    return boxes[mask], score[mask], cls[mask]


def generate_predicted_boxes(model_cfg, class_names, point_cloud_range, voxel_size, head_outs, idx):
    """voxelnext_head.py:418-487 for batch size 1 → (boxes, scores, labels 1-based)."""
    hc = model_cfg.DENSE_HEAD
    pp = hc.POST_PROCESSING
    stride = hc.TARGET_ASSIGNER_CONFIG.FEATURE_MAP_STRIDE
    boxes_all, scores_all, labels_all = [], [], []
    for hi, d in enumerate(head_outs):
        names = [n for n in hc.CLASS_NAMES_EACH_HEAD[hi] if n in class_names]
        cmap = np.array([list(class_names).index(n) for n in names])
        hm = D.sigmoid(d["hm"])
        dim = np.exp(d["dim"].astype(np.float32))
        b, s, l = decode_bbox_from_voxels(idx, hm, d["center"], d["center_z"], dim, d["rot"], d.get("vel"), point_cloud_range,
                                          voxel_size, stride, pp.MAX_OBJ_PER_SAMPLE, pp.SCORE_THRESH, pp.POST_CENTER_LIMIT_RANGE)
        l = cmap[l] if len(l) else l.astype(np.int64)
        sel = D.class_agnostic_nms(s, b, pp.NMS_CONFIG.NMS_THRESH, pp.NMS_CONFIG.NMS_PRE_MAXSIZE, pp.NMS_CONFIG.NMS_POST_MAXSIZE)
        boxes_all.append(b[sel]); scores_all.append(s[sel]); labels_all.append(l[sel])
    return np.concatenate(boxes_all), np.concatenate(scores_all), np.concatenate(labels_all).astype(np.int64) + 1


def run(state, cfg, points, recipe=Recipe(), trace=None):
    dc = cfg.DATA_CONFIG
    pr = [float(v) for v in dc.POINT_CLOUD_RANGE]
    vox = {c.NAME: c for c in dc.DATA_PROCESSOR}["transform_points_to_voxels"]
    vs = [float(v) for v in vox.VOXEL_SIZE]
    feats, coords, _ = V.voxelize_mean(points, pr, vs, vox.MAX_POINTS_PER_VOXEL, vox.MAX_NUMBER_OF_VOXELS["test"])
    grid = V.grid_size_of(pr, vs)
    if trace is not None:
        trace["vfe"] = (feats, coords)
    sparse_shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
    x, idx, shape, levels = sparse_backbone(state, feats, coords, sparse_shape, recipe, trace)
    if trace is not None:
        trace["levels"] = levels
        trace["encoded"] = (x, idx, shape)
    outs = head(state, cfg.MODEL, x, idx, shape, recipe, trace)
    if trace is not None:
        trace["heads"] = outs
    return generate_predicted_boxes(cfg.MODEL, cfg.CLASS_NAMES, pr, vs, outs, idx)
