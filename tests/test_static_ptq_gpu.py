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


def _set_static(model, static, Qz):
    """Calibrated amax of the oracle → the descriptors' quantizers (what compute_amax would have loaded)."""
    mods = dict(model.named_modules())
    for path, d in static.items():
        m = mods[path]
        if "act" in d:
            m.act_quant._amax = torch.from_numpy(np.asarray(d["act"], np.float32).reshape(-1).copy())
        else:
            m._input_quantizer._amax = torch.from_numpy(np.asarray(d["x"], np.float32).reshape(-1).copy())
            m._weight_quantizer._amax = torch.from_numpy(np.asarray(d["w"], np.float32).reshape(-1).copy())
    Qz._touch()


@pytest.mark.parametrize("shrink", [1.0, 0.5])
def test_static_ptq_parity_with_the_oracle(dev, shrink):
    """quant(..., static=True) against oracle.centerpoint.calibrate / Recipe(static=...):
      (a) every quantizer's calibrated amax (running max over the calibration frames of what it sees in the fp32 forward);
      (b) the static forward, teacher-forced layer by layer on a frame that was NOT in the calibration set, so values beyond
          the calibrated range clip at -128 / 127; shrink = 0.5 halves every amax to make clipping the common case."""
    import teacher
    from amdq3d import quant as Qz
    from amdq3d.datasets import load_data_to_gpu
    cfg = small_cfg("waymo")
    ds, model = build_model(cfg, n_points=7000, n_voxels=2000, n_frames=3)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.cuda()
    loader = [ds.collate_batch([ds[i]]) for i in (1, 2)]
    Qz.quant(model, 8, 8, sq=True, alpha=0.5, static=True, test_loader=loader, n_batches=5)
    rec = OC.Recipe(8, 8, True, 0.5)
    static = OC.calibrate(state, cfg, [ds[i]["points"] for i in (1, 2)], rec)
    mods = dict(model.named_modules())
    assert len(static) == 37
    for path, d in static.items():
        m = mods[path]
        if "act" in d:
            np.testing.assert_allclose(m.act_quant.amax.numpy(), d["act"], rtol=2e-5, atol=1e-7, err_msg=path)
        else:
            np.testing.assert_allclose(m._input_quantizer.amax.numpy().reshape(-1), np.asarray(d["x"]).reshape(-1), rtol=2e-5, err_msg=path)
            np.testing.assert_allclose(m._weight_quantizer.amax.numpy(), d["w"], rtol=2e-5, atol=1e-7, err_msg=path)
    if shrink != 1.0:
        static = {p: {k: np.asarray(v, np.float32) * np.float32(shrink) for k, v in d.items()} for p, d in static.items()}
    model.static_keep_f32 = True                           # teacher forcing reads every layer's fp32 output
    _set_static(model, static, Qz)                         # identical amax on both sides: codes must agree bit for bit
    batch = ds.collate_batch([ds[0]]); load_data_to_gpu(batch, dev)
    with torch.no_grad():
        model(batch)
    trace = {}
    OC.run(state, cfg, ds[0]["points"], OC.Recipe(8, 8, True, 0.5, static=static), trace)
    # clipping is exercised: the test frame exceeds the calibrated range somewhere
    x1 = trace["backbone_3d.conv_input.0"]
    clipped = int((np.abs(x1) > np.asarray(static["backbone_3d.conv1.0.conv1"]["act"])[None, :]).sum())
    assert clipped > 0
    eng = model.engine()
    idx_sets = [trace["vfe"][1]] + [trace["levels"][n][1] for n in ("x_conv2", "x_conv3", "x_conv4")] + [trace["encoded"][1]]
    maps = teacher.row_maps(eng, idx_sets)
    res = teacher.sparse_layers(eng, maps, lambda n: trace["vfe"][0] if n == "vfe" else trace[n])
    res += teacher.dense_layers(eng, lambda n: trace[n])
    teacher.report("static PTQ (amax x %.1f), teacher-forced layers; %d first-layer inputs beyond the calibrated range" % (shrink, clipped), res)
    bad = [(p, e) for p, k, e, _ in res if not e <= 2e-5]
    assert not bad, bad


def test_static_fast_path_equals_the_launch_per_quantiser_plan(dev):
    """Static PTQ fast path: consumer quantisers in producer epilogues, the weight copy built once, fp32 rows
    nobody reads skipped — against the same calibrated model with every quantiser as its own launch on stored fp32 rows:
    identical codes for every coded layer, identical detections; and the per-frame quantiser launches are gone."""
    from amdq3d import quant as Qz
    from amdq3d.datasets import load_data_to_gpu
    cfg = small_cfg("waymo")
    ds, model = build_model(cfg, n_points=7000, n_voxels=2000, n_frames=3)
    model.cuda()
    loader = [ds.collate_batch([ds[i]]) for i in (1, 2)]
    Qz.quant(model, 8, 8, sq=True, alpha=0.5, static=True, test_loader=loader, n_batches=5)

    def run(force_rows):
        eng = model.engine()
        eng._force_rows = force_rows
        eng.use_graph = False
        batch = ds.collate_batch([ds[0]]); load_data_to_gpu(batch, dev)
        eng.start_timing(["quantize_rows"])
        with torch.no_grad():
            pred, _ = model(batch)
        n_q = eng.stop_timing().get("quantize_rows", (0, 0.0))[0]
        codes = {}
        for r in eng.sparse_convs:
            if r.get("coded"):
                m = int(eng.levels[r["x"]["level"]]["count"].item())
                codes[r["path"]] = r["codes"][:m].clone()
        return eng, pred[0], codes, n_q

    eng, pred_fast, codes_fast, nq_fast = run(False)
    fused = [r["path"] for r in eng.sparse_convs if r.get("static") and r["static"]["fused"]]
    skipped = [r["path"] for r in eng.sparse_convs if r.get("skip_f32")]
    assert len(fused) == 20 and len(skipped) == 8, (len(fused), skipped)      # every coded layer; the conv1 of the 8 residual blocks
    assert nq_fast == 0
    model.static_keep_f32 = True
    model.invalidate_engine()
    eng2, pred_ref, codes_ref, nq_ref = run(True)
    assert nq_ref == 20 and not any(r.get("skip_f32") for r in eng2.sparse_convs)
    for path in codes_ref:
        assert torch.equal(codes_fast[path], codes_ref[path]), path
    for k in ("pred_boxes", "pred_scores", "pred_labels"):
        assert torch.equal(pred_fast[k], pred_ref[k]), k
