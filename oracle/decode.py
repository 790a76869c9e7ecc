"""CenterHead decode + class-agnostic rotated NMS restated (numpy + the C geometry of iou3d_nms.c).

Follows pcdet/models/model_utils/centernet_utils.py:155-241 (_topk, decode_bbox_from_heatmap),
pcdet/models/dense_heads/center_head.py:297-365 (sigmoid / exp, label mapping, +1),
pcdet/models/model_utils/model_nms_utils.py:6-25 and pcdet/ops/iou3d_nms/iou3d_nms_utils.py:120-135.
torch.topk leaves the order of equal scores unspecified; the restatement fixes it to (score desc, index asc).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "_build", "liboracle_iou.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_iou.so")
        if not os.path.isfile(path):
            build_c()
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_iou_bev_matrix.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, fp]
        L.oracle_overlap_bev_matrix.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, fp]
        L.oracle_nms.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
        L.oracle_nms.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def iou_bev(boxes_a, boxes_b):
    a = np.ascontiguousarray(boxes_a, dtype=np.float32)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if a.shape[0] and b.shape[0]:
        _lib().oracle_iou_bev_matrix(_fptr(a), a.shape[0], a.shape[1], _fptr(b), b.shape[0], b.shape[1], _fptr(out))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-81 — BEV overlap x height overlap / union volume."""
    a = np.ascontiguousarray(boxes_a, dtype=np.float32)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32)
    ov = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if a.shape[0] and b.shape[0]:
        _lib().oracle_overlap_bev_matrix(_fptr(a), a.shape[0], a.shape[1], _fptr(b), b.shape[0], b.shape[1], _fptr(ov))
    a_max, a_min = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    b_max, b_min = (b[:, 2] + b[:, 5] / 2)[None, :], (b[:, 2] - b[:, 5] / 2)[None, :]
    ov_h = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None)
    ov3d = ov * ov_h
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (ov3d / np.clip(va + vb - ov3d, 1e-6, None)).astype(np.float32)


def nms_sorted(boxes, thresh):
    """boxes already sorted by descending score → kept indices in keep order."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    keep = np.zeros((max(b.shape[0], 1),), dtype=np.int32)
    n = _lib().oracle_nms(_fptr(b), b.shape[0], b.shape[1], np.float32(thresh),
                          keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return keep[:n]


def _stable_desc(values):
    return np.argsort(-values.astype(np.float64), kind="stable")   # score desc, index asc on ties


def sigmoid(x):
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def decode_bbox_from_heatmap(hm, center, center_z, dim, rot, vel, point_cloud_range, voxel_size, feature_map_stride, K,
                             score_thresh, post_center_limit_range):
    """One sample. hm [C,H,W] (post-sigmoid), center [2,H,W], center_z [1,H,W], dim [3,H,W] (post-exp),
    rot [2,H,W] (cos, sin), vel [2,H,W] | None → boxes [n,7|9], scores [n], labels [n] (head-local)."""
    C, H, W = hm.shape
    flat = hm.reshape(C, H * W)
    Keff = min(K, H * W)
    scores_c = np.zeros((C, Keff), np.float32)
    inds_c = np.zeros((C, Keff), np.int64)
    for c in range(C):
        o = _stable_desc(flat[c])[:Keff]
        scores_c[c], inds_c[c] = flat[c][o], o
    o2 = _stable_desc(scores_c.reshape(-1))[:Keff]
    score = scores_c.reshape(-1)[o2]
    cls = (o2 // Keff).astype(np.int32)
    ind = inds_c.reshape(-1)[o2]
    ys = (ind // W).astype(np.float32)
    xs = (ind % W).astype(np.float32)
    g = lambda m: m.reshape(m.shape[0], -1)[:, ind].T.astype(np.float32)
    ctr, cz, dm, rt = g(center), g(center_z), g(dim), g(rot)
    angle = np.arctan2(rt[:, 1], rt[:, 0]).astype(np.float32)
    f32 = np.float32
    xs = (xs + ctr[:, 0]) * f32(feature_map_stride) * f32(voxel_size[0]) + f32(point_cloud_range[0])
    ys = (ys + ctr[:, 1]) * f32(feature_map_stride) * f32(voxel_size[1]) + f32(point_cloud_range[1])
    parts = [xs[:, None], ys[:, None], cz, dm, angle[:, None]]
    if vel is not None:
        parts.append(g(vel))
    boxes = np.concatenate(parts, axis=1).astype(np.float32)
    r = np.asarray(post_center_limit_range, dtype=np.float32)
    mask = np.all(boxes[:, :3] >= r[None, :3], axis=1) & np.all(boxes[:, :3] <= r[None, 3:], axis=1)
    if score_thresh is not None:
        mask &= score > f32(score_thresh)
    return boxes[mask], score[mask], cls[mask]


def class_agnostic_nms(scores, boxes, thresh, pre_max, post_max):
    """model_nms_utils.py:6-25 with score_thresh=None → selected indices into boxes."""
    if scores.shape[0] == 0:
        return np.zeros((0,), np.int64)
    order = _stable_desc(scores)[:min(pre_max, scores.shape[0])]
    keep = nms_sorted(boxes[order][:, :7], thresh)
    return order[keep[:post_max]]
