"""What the bench's (random-weight) center head hands the top-K: per class, how many scores pass the threshold, how they spread over
the select's 32-bins-per-octave histogram, how many share the pivot bin.  python scratch/hm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build(os.path.join(ROOT, "cfgs/waymo_models/centerpoint.yaml"), dev)
from amdq3d.datasets import load_data_to_gpu
b = ds.collate_batch([ds[0]])
load_data_to_gpu(b, dev)
with torch.no_grad():
    model(b)
eng = model.engine()
torch.cuda.synchronize()
for h in eng.heads:
    buf = h["buf"][0].reshape(-1, eng.HEAD_LD).cpu().numpy()
    ncls = h["cfg"].num_classes if "cfg" in h else 3
    for c in range(ncls):
        s = (1.0 / (1.0 + np.exp(-buf[:, c].astype(np.float32)))).astype(np.float32)
        keys = s.view(np.uint32)
        cand = keys[s > 0.1]
        bins = np.minimum(cand >> 18, 4095)
        hist = np.bincount(bins, minlength=4096)
        cum = np.cumsum(hist[::-1])
        K = min(500, len(cand))
        piv = 4095 - int(np.searchsorted(cum, K)) if K else -1
        print("class", c, "logit mean %.3f std %.3f" % (buf[:, c].mean(), buf[:, c].std()), "above 0.1:", len(cand), "distinct bins", int((hist > 0).sum()),
              "pivot bin count", int(hist[piv]) if K else 0, "max bin", int(hist.max()))
