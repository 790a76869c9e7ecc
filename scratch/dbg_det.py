import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "centerpoint")
cfgp, npts, nvox, _ = bench.WORKLOADS[wl]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
NF = 8
batches = []
for i in range(NF):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
D = int(os.environ.get("INFLIGHT", "2")); steps = int(os.environ.get("STEPS", "200"))
with torch.no_grad():
    eng = model.engine(); eng.use_graph = False
    ref = []
    for i in range(NF):
        r = model(dict(batches[i]))[0][0]
        ref.append((r["pred_boxes"].cpu().numpy().copy(), r["pred_scores"].cpu().numpy().copy()))
    eng.use_graph = True
    print("ref counts", [len(r[1]) for r in ref], flush=True)
    pending = collections.deque(); bad = 0
    import numpy as np
    def check(i, p):
        global bad
        r = p.result()[0][0]
        b, s = r["pred_boxes"].cpu().numpy(), r["pred_scores"].cpu().numpy()
        rb, rs = ref[i % NF]
        if b.shape != rb.shape or not (np.array_equal(b, rb) and np.array_equal(s, rs)):
            bad += 1
            if bad <= 10:
                e = p.eng
                k = ([model._engine] + model._extra_engines).index(e)
                c = p.counts()
                torch.cuda.synchronize()
                lv = [int(l["count"].item()) for l in e.levels]
                common = min(len(s), len(rs))
                nd = int((s[:common] != rs[:common]).sum())
                print("MISMATCH frame", i, "n", len(s), "ref", len(rs), "engine", k, "slot", p.slot, "counts()", int(c.sum()), "levels now", lv, "score diffs in common prefix", nd, "max|ds|", float(np.abs(s[:common] - rs[:common]).max()), flush=True)
    for i in range(steps):
        pending.append((i, model.forward_async(dict(batches[i % NF]))))
        if len(pending) >= D:
            check(*pending.popleft())
    while pending:
        check(*pending.popleft())
    print("engines", 1 + len(model._extra_engines), "inflight", D, "mismatches", bad, "of", steps)
