#!/usr/bin/env python
"""Times the stem layer (conv_input: SubMConv3d 5 -> 16, fp32) of a W150k frame on its real index level, and checks it against the
oracle's gather_conv. python scratch/stem_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import numpy as np, torch
from amdq3d import ops
from amdq3d.config import AttrDict, cfg_from_yaml_file
from amdq3d.synthetic import make_frame
from oracle import sparse as OS, voxel as V
dev = torch.device("cuda:0")
cfg = cfg_from_yaml_file(os.path.join(ROOT, "cfgs/waymo_models/centerpoint.yaml"), AttrDict())
pr, vs = [float(v) for v in cfg.DATA_CONFIG.POINT_CLOUD_RANGE], [0.1, 0.1, 0.15]
pts = make_frame(4, 150000, 30000, pr, vs, [1504, 1504, 40])
feats, idx, _ = V.voxelize_mean(pts, pr, vs, 5, 150000)
m = len(idx); shape = [41, 1504, 1504]
cap = 40960
lv = ops.IndexLevel(shape, 1, cap, dev)
rows = ops.level_from_coords(torch.from_numpy(idx.astype(np.int32)).to(dev), lv).cpu().numpy()
nbr = ops.subm_neighbors(lv)
inv = np.argsort(rows)
rng = np.random.default_rng(0)
w = (rng.standard_normal((16, 3, 3, 3, 5)) / np.sqrt(13 * 5)).astype(np.float32)
x = feats[inv].astype(np.float32)
xd = torch.zeros((cap, 5), device=dev); xd[:m] = torch.from_numpy(x).to(dev)
wk = torch.from_numpy(np.ascontiguousarray(w.reshape(16, 27, 5).transpose(1, 0, 2))).to(dev)
count = lv.count
am = torch.zeros(16, device=dev)
out = torch.zeros((cap, 16), device=dev)
def run():
    ops.spconv(xd, wk, nbr, count, 16, relu=True, absmax_out=am, out=out)
run(); torch.cuda.synchronize()
nbr_ref = OS.subm_neighbors(idx, shape)
ref = np.maximum(OS.gather_conv(feats.astype(np.float32), w, nbr_ref, None), 0)[inv]
got = out[:m].cpu().numpy()
print("max err", np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), "absmax ok", np.allclose(am.cpu().numpy(), np.abs(ref).max(axis=0), rtol=1e-6))
t0 = time.time()
while time.time() - t0 < 0.05:
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record(); torch.cuda.synchronize()
print("stem 5->16 @ %d rows: %.2f us per launch (host-paced eager loop; rocprof has the kernel time)" % (m, e0.elapsed_time(e1) * 5))
