"""GPU: static PTQ (quant(..., static=True) → collect_stats + compute_amax, quant/quantize.py:175-207) through the engine."""
import numpy as np
import pytest
import torch

from helpers import build_model, small_cfg
from oracle import centerpoint as OC

pytestmark = pytest.mark.gpu


def test_static_calibration_collects_running_max_and_runs(dev):
    from amdq3d import quant as Qz
    from amdq3d.datasets import load_data_to_gpu
    cfg = small_cfg("waymo")
    ds, model = build_model(cfg, n_points=7000, n_voxels=2000, n_frames=3)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    model.cuda()
    loader = [ds.collate_batch([ds[i]]) for i in range(3)]
    Qz.quant(model, 8, 8, sq=True, alpha=0.5, static=True, test_loader=loader, n_batches=1)   # 3 forwards: i = 0, 1, 2
    mods = dict(model.named_modules())
    quantizers = [m for m in mods.values() if isinstance(m, Qz.TensorQuantizer)]
    assert len(quantizers) == 2 * 20 + 2 * 17
    assert all(q.amax is not None and not q._if_calib and q._if_quant for q in quantizers)
    # the activation quantizer of the first quantized sparse conv saw conv_input's output of every calibration frame (fp32 pass)
    ref = None
    for i in range(3):
        tr = {}
        OC.run(state, cfg, ds[i]["points"], None, tr)
        a = np.abs(tr["backbone_3d.conv_input.0"]).max(axis=0)
        ref = a if ref is None else np.maximum(ref, a)
    got = mods["backbone_3d.conv1.0.conv1"].act_quant.amax.numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    # first BEV conv: per-tensor input amax = max over frames of max_j act_scale_j / s_j (fp32 features) — positive, finite
    sq0 = mods["backbone_2d.blocks.0.1"]
    assert sq0._input_quantizer.amax.numel() == 1 and float(sq0._input_quantizer.amax) > 0
    assert sq0._weight_quantizer.amax.numel() == 128 and bool(torch.isfinite(sq0._weight_quantizer.amax).all())
    # static forward runs and stays close to the dynamic one on a calibration frame
    batch = ds.collate_batch([ds[0]]); load_data_to_gpu(batch, dev)
    with torch.no_grad():
        pred_s, _ = model(batch)
    bev_static = model.engine().bev_cat[0].clone()
    ds2, model2 = build_model(cfg, n_points=7000, n_voxels=2000, n_frames=3)
    Qz.quant(model2, 8, 8, sq=True, alpha=0.5, static=False)
    model2.cuda()
    batch = ds2.collate_batch([ds2[0]]); load_data_to_gpu(batch, dev)
    with torch.no_grad():
        pred_d, _ = model2(batch)
    bev_dyn = model2.engine().bev_cat[0]
    rel = float((bev_static - bev_dyn).pow(2).mean().sqrt() / bev_dyn.pow(2).mean().sqrt())
    assert rel < 0.2
    assert pred_s[0]["pred_boxes"].shape[1] == 7 and pred_s[0]["pred_scores"].numel() > 0
