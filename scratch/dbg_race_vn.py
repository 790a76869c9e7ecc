"""VoxelNeXt: plan A replays one frame while plan B runs frames beside it; A's stages are checksummed against the first pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfgp, npts, nvox, _ = bench.WORKLOADS["voxelnext"]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
model.async_streams = 2
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
def sums(e):
    out = [("counts", tuple(int(l["count"].item()) for l in e.levels))]
    w = torch.arange(1, e.cap + 1, device=dev, dtype=torch.float64)
    for li, l in enumerate(e.levels):
        n = int(l["count"].item())
        out.append(("coords%d" % li, float((l["coords"][:n].double().sum(1) * w[:n]).sum())))
        if "nbr" in l and l["nbr"] is not None:
            out.append(("nbr%d" % li, float((l["nbr"][:n].double().sum(1) * w[:n]).sum())))
    for r in e.sparse_convs:
        n = int(e.levels[r["level"]]["count"].item())
        out.append((r["out"]["name"], float((r["out"]["buf"][:n].double().sum(1) * w[:n]).sum())))
    if getattr(e, "merge", None):
        m = e.merge; n = int(e.levels[m["level"]]["count"].item())
        out.append(("merge_out", float((m["out"]["buf"][:n].double().sum(1) * w[:n]).sum())))
    for hi, h in enumerate(e.heads):
        c = int(h["cand_count"][0].item())
        wv = torch.arange(1, c + 1, device=dev, dtype=torch.float64)
        out.append(("cand_count%d" % hi, c))
        out.append(("cand_scores_ord%d" % hi, float((h["cand_scores"][0][:c].double() * wv).sum())))
        out.append(("cand_boxes_ord%d" % hi, float((h["cand_boxes"][0][:c].double().sum(1) * wv).sum())))
    out.append(("det", tuple(e.det_counts[e.slot].flatten().tolist())))
    return out
with torch.no_grad():
    for i in range(12):
        model.forward_async(dict(batches[i % 4])).counts()
    A, Bn = model._engine, model._extra_engines[0]
    torch.cuda.synchronize()
    def run(e, b):
        e._ev_done[e.slot].synchronize()
        with torch.cuda.stream(e.stream):
            e.run(dict(batches[b]))
    run(A, int(os.environ.get("ABATCH", "2"))); torch.cuda.synchronize()
    slot0 = A.slot
    with torch.cuda.stream(A.stream):
        ref = {A.slot: sums(A)}
    run(A, int(os.environ.get("ABATCH", "2"))); torch.cuda.synchronize()
    with torch.cuda.stream(A.stream):
        ref[A.slot] = sums(A)
    print("ref ready", sorted(map(str, A._graphs)), flush=True)
    bad = 0; iters = int(os.environ.get("ITERS", "600"))
    for it in range(iters):
        run(Bn, (it + 1) % 4)
        run(A, int(os.environ.get("ABATCH", "2")))
        A.stream.synchronize()
        with torch.cuda.stream(A.stream):
            s = sums(A)
        d = [a[0] for a, c in zip(ref[A.slot], s) if a != c]
        if d:
            bad += 1
            if bad <= 8: print("iter", it, "first diffs", d[:6], flush=True)
    torch.cuda.synchronize()
    print("bad", bad, "of", iters)
