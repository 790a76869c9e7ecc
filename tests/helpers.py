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


def calibrate_hm_with_oracle(model, cfg, points, fraction=0.001):
    """The heat-map branch's BatchNorm statistics (amdq3d.synthetic.calibrate_heat_map) from one fp32 forward of the CPU oracle:
    the tests' route to the hidden hm activations (bench.py and tools/eval.py take them from the device engine)."""
    from amdq3d.synthetic import calibrate_heat_map, hm_hidden_paths
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tr = {}
    if cfg.MODEL.NAME == "VoxelNeXt":
        from oracle import voxelnext as OV
        OV.run(state, cfg, points, None, trace=tr)
    else:
        from oracle import centerpoint as OC
        OC.run(state, cfg, points, None, trace=tr)
    hidden = {}
    for path in hm_hidden_paths(model):
        t = tr[path]
        t = t.numpy() if hasattr(t, "numpy") else np.asarray(t)
        hidden[path] = np.moveaxis(t, 1, -1).reshape(-1, t.shape[1]) if t.ndim == 4 else t
    return calibrate_heat_map(model, hidden, fraction=fraction)


def build_model(cfg, n_points, n_voxels, n_frames=2, seed=4):
    from amdq3d.datasets import SyntheticDataset
    from amdq3d.detector import build_network
    from amdq3d.synthetic import init_synthetic_weights
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=n_frames, n_points=n_points, n_voxels=n_voxels, seed0=seed)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    init_synthetic_weights(model, seed)
    model.eval()
    cal = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=1, n_points=n_points, n_voxels=n_voxels, seed0=1000 + seed)
    # a frame that is not among the test's own; the small test ranges (BEV 32 x 32 .. 40 x 40) get a larger firing fraction so that
    # every frame has tens of boxes to compare
    calibrate_hm_with_oracle(model, cfg, cal[0]["points"], fraction=0.001 if n_voxels >= 15000 else 0.005)
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


def sane_boxes(boxes, min_dim=0.05, max_dim=60.0, max_aspect=40.0):
    """Boxes whose BEV rectangle is of a size and aspect ratio a detector produces on real data. The random-weight synthetic
    models also emit 0.03 m x 212 m slivers, for which a rotated IoU of 1 - 1e-3 resolves parameter differences of 1e-7 —
    below the fp32 summation-order noise of any two correct implementations."""
    d = boxes[:, 3:5]
    return (d.min(axis=1) >= min_dim) & (d.max(axis=1) <= max_dim) & (d.max(axis=1) <= max_aspect * d.min(axis=1))


def match_params(boxes_a, scores_a, labels_a, boxes_b, scores_b, labels_b, rel=1e-4, score_tol=1e-3):
    """Fraction of a's boxes with a partner in b whose PARAMETERS agree: centre within rel * max(1, largest extent), extents and
    velocity within rel (relative), heading within rel rad (mod 2 pi), same label, |dscore| <= score_tol."""
    if len(boxes_a) == 0:
        return 1.0
    if len(boxes_b) == 0:
        return 0.0
    a, b = boxes_a[:, None, :], boxes_b[None, :, :]
    ext = np.maximum(1.0, np.abs(boxes_a[:, 3:6]).max(axis=1))[:, None]
    ok = (np.abs(a[..., :3] - b[..., :3]).max(axis=-1) <= rel * ext)
    ok &= (np.abs(a[..., 3:6] - b[..., 3:6]) <= rel * np.maximum(1e-3, np.abs(a[..., 3:6]))).all(axis=-1)
    dth = np.abs(a[..., 6] - b[..., 6])
    ok &= np.minimum(dth, np.abs(dth - 2 * np.pi)) <= rel
    if boxes_a.shape[1] > 7:
        ok &= (np.abs(a[..., 7:] - b[..., 7:]) <= rel * np.maximum(1.0, np.abs(a[..., 7:]))).all(axis=-1)
    ok &= labels_a[:, None] == labels_b[None, :]
    ok &= np.abs(scores_a[:, None] - scores_b[None, :]) <= score_tol
    return float(ok.any(axis=1).mean())
