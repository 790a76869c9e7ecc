import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
print("built", flush=True)
with torch.no_grad():
    for i in range(12):
        t = time.time()
        p = model.forward_async(dict(batches[i % 4]))
        c = p.counts()
        print("frame", i, "counts", int(c.sum()), "%.1f ms" % ((time.time() - t) * 1e3), flush=True)
    t = time.time(); prev = None; n = 0
    for i in range(0):
        cur = model.forward_async(dict(batches[i % 4]))
        if prev is not None: n += int(prev.counts().sum())
        prev = cur
        if i % 10 == 0: print("timed", i, flush=True)
    pass
    torch.cuda.synchronize()
    print("done")
