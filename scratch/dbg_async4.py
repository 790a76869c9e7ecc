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
def lvc(e):
    return [int(l["count"].item()) for l in e.levels][:6]
with torch.no_grad():
    for i in range(int(os.environ.get("NSYNC", "5"))):
        t = time.time()
        r = model(dict(batches[i % 4]))
        print("sync", i, "boxes", len(r[0][0]["pred_scores"]), "%.1f ms" % ((time.time() - t) * 1e3), lvc(model.engine()), sorted(map(str, model.engine()._graphs)), flush=True)
    for i in range(8):
        t = time.time()
        p = model.forward_async(dict(batches[i % 4]))
        t1 = time.time()
        p.eng.stream.synchronize(); t2 = time.time()
        lv = lvc(p.eng)
        p.eng.main_stream.synchronize(); t3 = time.time()
        c = p.counts()
        e = p.eng
        torch.cuda.synchronize()
        print("   slot", p.slot, "det_counts", e.det_counts.flatten().tolist(), "host", e.host_counts.flatten().tolist(), "cand", [int(h["cand_count"][0].item()) for h in e.heads], "spatial %.3f bev %.3f" % (float(e.spatial.abs().sum()), float(e.bev_cat.abs().sum())))
        print("async", i, "counts", int(c.sum()), "host %.1f idx %.1f main %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), lv, sorted(map(str, p.eng._graphs)), flush=True)
    print("done")
