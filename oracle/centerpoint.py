"""Whole-path CPU restatement of quantized CenterPoint inference (ORACLE — test infrastructure).

Composition follows pcdet/models/detectors/centerpoint.py:9-50 with the module order of
detector3d_template.py:23-26, and the quantization recipe of quant/quant_centerpoint.py:74-131:
    sq=True : QConvNd(cw=True) on every sparse conv except backbone_3d.conv_input.0; SQConv2d(alpha) on every nn.Conv2d
              except the head's final convs and the whole hm branch; ConvTranspose2d stays fp32.
    sq=False: QConvNd(cw=False) on every sparse conv; per-tensor/per-oc fake quant on every nn.Conv2d.
    None    : fp32 model.
Input: a pcdet-style state dict (name → torch tensor), the model cfg, the dataset geometry and ONE sample's points.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import decode as D
from . import dense as DN
from . import sparse as S
from . import voxel as V

BACKBONE_NO_LIST = ["backbone_3d.conv_input.0"]                                  # quant_centerpoint.py:24-26
HEAD_NO_SUFFIX = ("center.1", "center_z.1", "dim.1", "rot.1", "vel.1", "hm.0.0", "hm.1")   # quant_centerpoint.py:28-71


def _bn(state, prefix):
    return {k: state["%s.%s" % (prefix, k)].float() for k in ("weight", "bias", "running_mean", "running_var")}


def _bn_np(state, prefix):
    return {k: v.numpy() for k, v in _bn(state, prefix).items()}


class Recipe:
    """static: {module path: {'act': amax [C] or scalar (QConvNd) | 'x': float, 'w': [O] (SQConv2d / quant conv)}} — calibrated
    amax of quant_centerpoint.py:127-129 (collect_stats + compute_amax), the quantizers then clip instead of following the
    frame. collect: a dict that a CALIBRATION forward fills with the running element-wise max of what each quantizer sees
    (MaxCalibrator); in that forward nothing is quantised (quantize.py:178-181: enable_calib + disable_quant)."""

    def __init__(self, w_bits=8, act_bits=8, sq=True, alpha=0.5, static=None, collect=None):
        self.w_bits, self.act_bits, self.sq, self.alpha = w_bits, act_bits, sq, alpha
        self.static, self.collect = static, collect


def _running_max(store, path, key, value):
    value = np.asarray(value, dtype=np.float32)
    d = store.setdefault(path, {})
    d[key] = value.copy() if key not in d else np.maximum(d[key], value)


def calibrate(state, cfg, frames, recipe):
    """collect_stats (quant/quantize.py:175-195) over `frames` (list of point arrays) → the `static` dict for Recipe."""
    store = {}
    cal = Recipe(recipe.w_bits, recipe.act_bits, recipe.sq, recipe.alpha, collect=store)
    for pts in frames:
        run(state, cfg, pts, cal)
    return store


def sparse_backbone(state, feats, coords, sparse_shape, recipe, trace=None):
    """VoxelResBackBone8x.forward (spconv_backbone.py:243-295). Returns (features, indices, shape) of conv_out."""
    P = "backbone_3d."

    def conv(path, x, nbr, bn_prefix, relu=True, residual=None):
        w = state[P + path + ".weight"].float().numpy()
        b = state.get(P + path + ".bias", None)
        b = None if b is None else b.float().numpy()
        quantized = recipe is not None and not (recipe.sq and (P + path) in BACKBONE_NO_LIST)
        if quantized and recipe.collect is not None:
            from . import quantizer as Q
            _running_max(recipe.collect, P + path, "act", Q.reduce_amax(x, axis=1 if recipe.sq else None).reshape(-1))
        elif quantized:
            w, _, _ = S.qconv_weight(w, recipe.w_bits)      # constant weights: calibrated amax == dynamic amax
            amax = None
            if recipe.static is not None:
                amax = np.asarray(recipe.static[P + path]["act"], dtype=np.float32)
                amax = amax.reshape(1, -1) if recipe.sq else np.float32(amax.reshape(-1)[0])
            x, _ = S.qconv_act(x, recipe.act_bits, cw=recipe.sq, amax=amax)
        y = S.gather_conv(x, w, nbr, b)
        y = S.batchnorm(y, _bn_np(state, P + bn_prefix), 1e-3)
        if residual is not None:
            y = y + residual
        if relu:
            y = np.maximum(y, 0)
        if trace is not None:
            trace[P + path] = y
        return y.astype(np.float32)

    def block(prefix, x, nbr):
        t = conv(prefix + ".conv1", x, nbr, prefix + ".bn1")
        return conv(prefix + ".conv2", t, nbr, prefix + ".bn2", residual=x)

    idx, shape = coords, list(sparse_shape)
    nbr = S.subm_neighbors(idx, shape)
    x = conv("conv_input.0", feats, nbr, "conv_input.1")
    x = block("conv1.0", x, nbr)
    x = block("conv1.1", x, nbr)
    levels = {"x_conv1": (x, idx, shape)}
    for name, pad in (("conv2", (1, 1, 1)), ("conv3", (1, 1, 1)), ("conv4", (0, 1, 1))):
        idx2, nbr_d, shape2 = S.conv_outputs(idx, shape, (3, 3, 3), (2, 2, 2), pad)
        x = conv(name + ".0.0", x, nbr_d, name + ".0.1")
        idx, shape = idx2, shape2
        nbr = S.subm_neighbors(idx, shape)
        x = block(name + ".1", x, nbr)
        x = block(name + ".2", x, nbr)
        levels["x_" + name] = (x, idx, shape)
    idx2, nbr_d, shape2 = S.conv_outputs(idx, shape, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    x = conv("conv_out.0", x, nbr_d, "conv_out.1")
    return x, idx2, shape2, levels


def _conv2d(state, path, x, stride, padding, recipe, quantized):
    w = state[path + ".weight"].float()
    b = state.get(path + ".bias", None)
    b = None if b is None else b.float()
    if quantized and recipe.collect is not None:
        y, st = DN.sq_conv2d(x, w, b, stride, padding, recipe.alpha if recipe.sq else None, recipe.w_bits, recipe.act_bits,
                             return_stats=True, quantize=False)
        _running_max(recipe.collect, path, "x", st["amax_x"])
        _running_max(recipe.collect, path, "w", st["amax_w"])
        return y
    if quantized:
        st = recipe.static[path] if recipe.static is not None else {}
        return DN.sq_conv2d(x, w, b, stride, padding, recipe.alpha if recipe.sq else None, recipe.w_bits, recipe.act_bits,
                            amax_x=st.get("x"), amax_w=st.get("w"))
    return F.conv2d(x, w, b, stride, padding)


def bev_backbone(state, model_cfg, x, recipe, trace=None):
    """BaseBEVBackbone.forward (base_bev_backbone.py:81-112) for the [ZeroPad, Conv, BN, ReLU, (Conv, BN, ReLU)*n] blocks."""
    c2 = model_cfg.BACKBONE_2D
    P = "backbone_2d."
    ups = []
    q = recipe is not None
    for bi, (n, s) in enumerate(zip(c2.LAYER_NUMS, c2.LAYER_STRIDES)):
        pre = "%sblocks.%d." % (P, bi)
        x = F.pad(x, (1, 1, 1, 1))                                            # nn.ZeroPad2d(1)
        x = torch.relu(DN.bn2d(_conv2d(state, pre + "1", x, s, 0, recipe, q), _bn(state, pre + "2"), 1e-3))
        if trace is not None:
            trace[pre + "1"] = x
        for k in range(n):
            ci = 4 + 3 * k
            x = torch.relu(DN.bn2d(_conv2d(state, pre + str(ci), x, 1, 1, recipe, q), _bn(state, pre + str(ci + 1)), 1e-3))
            if trace is not None:
                trace[pre + str(ci)] = x
        d = c2.UPSAMPLE_STRIDES[bi]
        wd = state["%sdeblocks.%d.0.weight" % (P, bi)].float()
        u = F.conv_transpose2d(x, wd, None, stride=d)                          # stays fp32 (not an nn.Conv2d)
        u = torch.relu(DN.bn2d(u, _bn(state, "%sdeblocks.%d.1" % (P, bi)), 1e-3))
        if trace is not None:
            trace["%sdeblocks.%d.0" % (P, bi)] = u
        ups.append(u)
    return torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]


def center_head(state, model_cfg, class_names, x, recipe, trace=None):
    """CenterHead.forward up to the raw per-head maps (center_head.py:385-391, SeparateHead :12-46)."""
    hc = model_cfg.DENSE_HEAD
    P = "dense_head."
    eps = hc.get("BN_EPS", 1e-5)
    q = recipe is not None
    x = torch.relu(DN.bn2d(_conv2d(state, P + "shared_conv.0", x, 1, 1, recipe, q), _bn(state, P + "shared_conv.1"), eps))
    if trace is not None:
        trace[P + "shared_conv.0"] = x
    outs = []
    for hi, names in enumerate(hc.CLASS_NAMES_EACH_HEAD):
        branches = list(hc.SEPARATE_HEAD_CFG.HEAD_ORDER) + ["hm"]
        d = {}
        for br in branches:
            pre = "%sheads_list.%d.%s." % (P, hi, br)
            # sq recipe: '<br>.0.0' quantised except hm; '<br>.1' never. sq=False recipe: everything quantised.
            q0 = q and (not recipe.sq or (br + ".0.0") not in HEAD_NO_SUFFIX)
            q1 = q and (not recipe.sq)
            t = torch.relu(DN.bn2d(_conv2d(state, pre + "0.0", x, 1, 1, recipe, q0), _bn(state, pre + "0.1"), eps))
            d[br] = _conv2d(state, pre + "1", t, 1, 1, recipe, q1)
            if trace is not None:
                trace[pre + "0.0"] = t
                trace[pre + "1"] = d[br]
        outs.append(d)
    return outs


def generate_predicted_boxes(model_cfg, class_names, point_cloud_range, voxel_size, head_outs):
    """center_head.py:297-365 for batch size 1 → (boxes, scores, labels 1-based)."""
    hc = model_cfg.DENSE_HEAD
    pp = hc.POST_PROCESSING
    stride = hc.TARGET_ASSIGNER_CONFIG.FEATURE_MAP_STRIDE
    boxes_all, scores_all, labels_all = [], [], []
    for hi, d in enumerate(head_outs):
        names = [n for n in hc.CLASS_NAMES_EACH_HEAD[hi] if n in class_names]
        cmap = np.array([list(class_names).index(n) for n in names])
        hm = D.sigmoid(d["hm"][0].numpy())
        dim = np.exp(d["dim"][0].numpy().astype(np.float32))
        vel = d["vel"][0].numpy() if "vel" in d else None
        b, s, l = D.decode_bbox_from_heatmap(hm, d["center"][0].numpy(), d["center_z"][0].numpy(), dim, d["rot"][0].numpy(),
                                             vel, point_cloud_range, voxel_size, stride, pp.MAX_OBJ_PER_SAMPLE,
                                             pp.SCORE_THRESH, pp.POST_CENTER_LIMIT_RANGE)
        l = cmap[l] if len(l) else l.astype(np.int64)
        sel = D.class_agnostic_nms(s, b, pp.NMS_CONFIG.NMS_THRESH, pp.NMS_CONFIG.NMS_PRE_MAXSIZE, pp.NMS_CONFIG.NMS_POST_MAXSIZE)
        boxes_all.append(b[sel]); scores_all.append(s[sel]); labels_all.append(l[sel])
    nb = max(b.shape[1] for b in boxes_all)
    boxes_all = [b if b.shape[1] == nb else np.concatenate([b, np.zeros((len(b), nb - b.shape[1]), np.float32)], 1) for b in boxes_all]
    return np.concatenate(boxes_all), np.concatenate(scores_all), np.concatenate(labels_all).astype(np.int64) + 1


def run(state, cfg, points, recipe=Recipe(), trace=None):
    """cfg: full config (CLASS_NAMES, DATA_CONFIG, MODEL). points f32 [N, C] of one sample (already range-masked)."""
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
    sp = DN.height_compression(x, idx, shape, 1)
    if trace is not None:
        trace["spatial_features"] = sp
    bev = bev_backbone(state, cfg.MODEL, sp, recipe, trace)
    if trace is not None:
        trace["spatial_features_2d"] = bev
    heads = center_head(state, cfg.MODEL, cfg.CLASS_NAMES, bev, recipe, trace)
    if trace is not None:
        trace["heads"] = heads
    return generate_predicted_boxes(cfg.MODEL, cfg.CLASS_NAMES, pr, vs, heads)
