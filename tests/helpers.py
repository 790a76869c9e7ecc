"""Shared test helpers: small CenterPoint configs, synthetic model + state, box matching."""
import copy

import numpy as np


def small_cfg(kind="waymo", half_range=12.8):
    """The real model config on a shrunken point-cloud range (grid 256x256x40 → BEV 32x32) so the CPU oracle runs in seconds."""
    from amdq3d.config import AttrDict, cfg_from_yaml_file
    path = {"waymo": "cfgs/waymo_models/centerpoint.yaml",
            "nuscenes": "cfgs/nuscenes_models/cbgs_voxel0075_res3d_centerpoint.yaml",
            "voxelnext": "cfgs/nuscenes_models/cbgs_voxel0075_voxelnext.yaml"}[kind]
    cfg = cfg_from_yaml_file(path, AttrDict())
    r = cfg.DATA_CONFIG.POINT_CLOUD_RANGE
    if kind in ("nuscenes", "voxelnext"):
        half_range = 12.0          # 0.075 m voxels → 320 cells
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-half_range, -half_range, r[2], half_range, half_range, r[5]]
    pp = cfg.MODEL.DENSE_HEAD.POST_PROCESSING
    pp.POST_CENTER_LIMIT_RANGE = [-half_range, -half_range, pp.POST_CENTER_LIMIT_RANGE[2], half_range, half_range,
                                  pp.POST_CENTER_LIMIT_RANGE[5]]
    return cfg


def build_model(cfg, n_points, n_voxels, n_frames=2, seed=4):
    from amdq3d.datasets import SyntheticDataset
    from amdq3d.detector import build_network
    from amdq3d.synthetic import init_synthetic_weights
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=n_frames, n_points=n_points, n_voxels=n_voxels, seed0=seed)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    init_synthetic_weights(model, seed)
    model.eval()
    return ds, model


def match_boxes(boxes_a, scores_a, labels_a, boxes_b, scores_b, labels_b, iou_tol=1e-3, score_tol=1e-3, z_tol=1e-3):
    """Fraction of a's boxes that have a partner in b with BEV IoU >= 1 - iou_tol, same label, |dscore| <= score_tol
    and |dz|, |dh| within z_tol (the BEV IoU ignores the vertical extent)."""
    from oracle import decode as OD
    if len(boxes_a) == 0:
        return 1.0
    if len(boxes_b) == 0:
        return 0.0
    iou = OD.iou_bev(boxes_a[:, :7], boxes_b[:, :7])
    ok = iou >= 1.0 - iou_tol
    ok &= labels_a[:, None] == labels_b[None, :]
    ok &= np.abs(scores_a[:, None] - scores_b[None, :]) <= score_tol
    ok &= np.abs(boxes_a[:, None, 2] - boxes_b[None, :, 2]) <= z_tol * np.maximum(1.0, np.abs(boxes_a[:, None, 2]))
    ok &= np.abs(boxes_a[:, None, 5] - boxes_b[None, :, 5]) <= z_tol * np.maximum(1.0, np.abs(boxes_a[:, None, 5]))
    return float(ok.any(axis=1).mean())
