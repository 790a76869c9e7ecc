"""tools/eval.py — the counterpart of tools/eval_utils/eval_utils.py:20-139: iterate a dataset, model(batch_dict),
generate_prediction_dicts, result.pkl, --infer_time on HIP events."""
import argparse
import importlib.util
import logging
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import build_model, small_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_eval():
    spec = importlib.util.spec_from_file_location("aq3d_tools_eval", os.path.join(ROOT, "tools", "eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_eval_loop_writes_the_reference_result_records(dev, tmp_path):
    from amdq3d import quant
    ev = _load_eval()
    cfg = small_cfg("waymo")
    ds, model = build_model(cfg, n_points=9000, n_voxels=2500, n_frames=3)
    quant.quant(model, 8, 8, sq=True, alpha=0.5, static=False)
    model.cuda()
    args = argparse.Namespace(infer_time=True)
    ret, annos = ev.eval_one_epoch(cfg, args, model, ds, [0, 1, 2], logging.getLogger("t"), str(tmp_path), dev)
    assert ret["infer_time_ms"] > 0 and ret["avg_pred_objects"] > 1
    assert all(k in ret for k in ("recall/roi_0.3", "recall/rcnn_0.5", "recall/rcnn_0.7"))
    with open(tmp_path / "result.pkl", "rb") as f:
        recs = pickle.load(f)
    assert len(recs) == 3
    for r, fid in zip(recs, ("synthetic_0000", "synthetic_0001", "synthetic_0002")):
        assert set(r) >= {"name", "score", "boxes_lidar", "pred_labels", "frame_id"} and r["frame_id"] == fid
        assert r["boxes_lidar"].shape == (len(r["score"]), 7) and r["pred_labels"].dtype == np.int64
        assert all(n in cfg.CLASS_NAMES for n in r["name"])
    # the records are what a plain forward gives
    from amdq3d.datasets import load_data_to_gpu
    b = ds.collate_batch([ds[1]])
    load_data_to_gpu(b, dev)
    with torch.no_grad():
        pred, _ = model(b)
    np.testing.assert_array_equal(recs[1]["score"], pred[0]["pred_scores"].cpu().numpy())


def test_eval_loop_with_batches(dev, tmp_path):
    """tools/test.py --batch_size 2: the fp32 model's records equal the frame-by-frame ones (frames are independent); three frames in
    batches of two (2 + 1)."""
    ev = _load_eval()
    cfg = small_cfg("waymo")
    ds, model = build_model(cfg, n_points=9000, n_voxels=2500, n_frames=3)
    model.cuda()
    r1, a1 = ev.eval_one_epoch(cfg, argparse.Namespace(infer_time=False), model, ds, [0, 1, 2], logging.getLogger("t"), str(tmp_path / "a"), dev)
    r2, a2 = ev.eval_one_epoch(cfg, argparse.Namespace(infer_time=False, batch_size=2), model, ds, [0, 1, 2], logging.getLogger("t"), str(tmp_path / "b"), dev)
    assert len(a1) == len(a2) == 3 and model.max_batch_size == 2
    for x, y in zip(a1, a2):
        assert x["frame_id"] == y["frame_id"]
        np.testing.assert_array_equal(x["score"], y["score"])
        np.testing.assert_array_equal(x["boxes_lidar"], y["boxes_lidar"])
