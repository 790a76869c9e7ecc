import os, sys, time
if os.environ.get("LATE_ENV"):
    import torch
    print("cuda initialized before env:", torch.cuda.is_initialized(), flush=True)
    if os.environ["LATE_ENV"] == "2":
        torch.cuda.init(); print("forced init", torch.cuda.is_initialized())
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from amdq3d.datasets import load_data_to_gpu
from amdq3d.engine import Engine
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
with torch.no_grad():
    for i in range(5):
        r = model(dict(batches[i % 4]))
        print("sync", i, "boxes", len(r[0][0]["pred_scores"]), flush=True)
    mode = os.environ.get("MODE", "engine")
    if mode == "engine":
        e1 = Engine(model, dev)
    elif mode.startswith("alloc"):
        mk = (lambda n: torch.empty((n,), device=dev)) if mode.endswith("E") else (lambda n: torch.full((n,), float("nan"), device=dev))
        junk = []
        if "B" in mode:
            junk += [mk(1 << 20) for _ in range(2000)]
        if "S" in mode:
            junk += [mk(1 << 12) for _ in range(20000)]
        if "T" in mode:
            junk += [mk(16) for _ in range(20000)]
        torch.cuda.synchronize()
    elif mode == "stream":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            r = model(dict(batches[0]))
            print("on side stream boxes", len(r[0][0]["pred_scores"]), sorted(map(str, model.engine()._graphs)), flush=True)
    for i in range(4):
        r = model(dict(batches[i % 4]))
        print("sync after", i, "boxes", len(r[0][0]["pred_scores"]), flush=True)
    print("done")
