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
    for i in range(int(os.environ.get("NFR", "10"))):
        t = time.time()
        p = model.forward_async(dict(batches[i % 4]))
        t1 = time.time()
        p.eng.stream.synchronize(); t2 = time.time()
        lv = [int(l["count"].item()) if hasattr(l.get("count", None), "item") else -1 for l in getattr(p.eng, "levels", [])][:6]
        p.eng.main_stream.synchronize(); t3 = time.time()
        c = p.counts()
        print("frame", i, "counts", int(c.sum()), "host %.1f idx %.1f main %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), lv, sorted(map(str, p.eng._graphs)), flush=True)
    print("done")
