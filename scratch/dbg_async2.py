import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
with torch.no_grad():
    for i in range(7):
        t = time.time()
        p = model.forward_async(dict(batches[i % 4]))
        c = p.counts()
        torch.cuda.synchronize()
        e = p.eng
        print("frame", i, "engine", 0 if e is model._engine else 1, "slot", p.slot, "counts", int(c.sum()), "%.1f ms" % ((time.time() - t) * 1e3),
              "levels", [int(lv["count"].item()) for lv in e.levels], "graphs", list(e._graphs.keys()),
              "ptsbuf sum %.3f" % float(e._pts_buf[:1000].sum()) if e._pts_buf is not None else "", flush=True)
