"""VoxelNeXt, two plans pipelined: every completed frame's stages are checksummed against a sequential pass of the same
batch on the same plan — names the first stage that differs when a pipelined frame deviates."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfgp, npts, nvox, _ = bench.WORKLOADS["voxelnext"]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
model.async_streams = 4
NF = 8
batches = []
for i in range(NF):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
def sums(e, slot):
    out = [("counts", tuple(int(l["count"].item()) for l in e.levels))]
    w = torch.arange(1, e.cap + 1, device=dev, dtype=torch.float64)
    n0 = int(e.levels[0]["count"].item())
    out.append(("vfe", float((e.vfe_feats[:n0].double().sum(1) * w[:n0]).sum())))
    for li, l in enumerate(e.levels):
        n = int(l["count"].item())
        out.append(("coords%d" % li, float((l["coords"][:n].double().sum(1) * w[:n]).sum())))
    for r in e.sparse_convs:
        n = int(e.levels[r["level"]]["count"].item())
        out.append((r["out"]["name"], float((r["out"]["buf"][:n].double().sum(1) * w[:n]).sum())))
    if getattr(e, "merge", None):
        m = e.merge; n = int(e.levels[m["level"]]["count"].item())
        out.append(("merge_out", float((m["out"]["buf"][:n].double().sum(1) * w[:n]).sum())))
    for hi, h in enumerate(e.heads):
        c = int(h["cand_count"][0].item())
        wv = torch.arange(1, c + 1, device=dev, dtype=torch.float64)
        out.append(("cand_scores_ord%d" % hi, float((h["cand_scores"][0][:c].double() * wv).sum())))
        out.append(("cand_boxes_ord%d" % hi, float((h["cand_boxes"][0][:c].double().sum(1) * wv).sum())))
        capn = min(h["K"], h["pre"]); cbn = (capn + 63) // 64
        mk = h["nms_ws"][: capn * cbn * 8].view(torch.int64).view(capn, cbn)[:c]
        out.append(("mask%d" % hi, int((mk >> 4).sum())))
        kc = int(e.det_counts[slot, 0, hi].item())
        out.append(("keepcount%d" % hi, kc))
        out.append(("keep%d" % hi, int(h["keep"][0][:kc].long().sum())))
    out.append(("det", tuple(e.det_counts[slot].flatten().tolist())))
    return out
with torch.no_grad():
    for i in range(16):
        model.forward_async(dict(batches[i % NF])).counts()
    engs = [model._engine] + model._extra_engines
    ref = {}; fin = {}; masks = {}
    def maskof(e, hi):
        h = e.heads[hi]; c = int(h['cand_count'][0].item()); capn = min(h['K'], h['pre']); cbn = (capn + 63) // 64
        return h['nms_ws'][: capn * cbn * 8].view(torch.int64).view(capn, cbn)[:c].clone()
    for b in range(NF):
        for k in range(len(engs)):
            while model._rr % len(engs) != k:
                model.forward_async(dict(batches[0])).counts()
            p = model.forward_async(dict(batches[b])); torch.cuda.synchronize()
            ref[(k, b)] = sums(p.eng, p.slot)[:-1]
            fin[(k, b)] = p.result()[0][0]["pred_scores"].cpu().numpy().copy()
            masks[(k, b)] = [maskof(p.eng, hi) for hi in range(len(p.eng.heads))]
    print("reference built", flush=True)
    pend = collections.deque(); bad = 0
    import numpy as np
    def check(i, k, p):
        global bad
        r = p.result()[0][0]
        sc = r["pred_scores"].cpu().numpy()
        if (k, i % NF) not in fin:
            return
        if sc.shape != fin[(k, i % NF)].shape or not np.array_equal(sc, fin[(k, i % NF)]):
            bad += 1
            s = sums(p.eng, p.slot)[:-1]
            d = [a[0] for a, c in zip(ref[(k, i % NF)], s) if a != c]
            if bad <= 10:
                print("frame", i, "engine", k, "batch", i % NF, "stage diffs", d[:8], flush=True)
                for hi in range(len(p.eng.heads)):
                    m = maskof(p.eng, hi); rm = masks[(k, i % NF)][hi]
                    if m.shape == rm.shape and not torch.equal(m, rm):
                        for r_, c_ in (m != rm).nonzero()[:2].tolist():
                            x = int(m[r_, c_]) ^ int(rm[r_, c_]); bit = (x & -x).bit_length() - 1
                            j = c_ * 64 + bit
                            bx = p.eng.heads[hi]["cand_boxes"][0]
                            print("   head", hi, "row", r_, "j", j, "got bit", (int(m[r_, c_]) >> bit) & 1, "\n     a", [float("%.9g" % v) for v in bx[r_].tolist()], "\n     b", [float("%.9g" % v) for v in bx[j].tolist()], flush=True)
    steps = int(os.environ.get("STEPS", "1500"))
    for i in range(steps):
        k = model._rr % len(engs)
        pend.append((i, k, model.forward_async(dict(batches[i % NF]))))
        if len(pend) >= 2:
            check(*pend.popleft())
    while pend: check(*pend.popleft())
    print("bad", bad, "of", steps)
    torch.cuda.synchronize()
    if os.environ.get("AQ3D_NMS_DBG"):
        os.environ["AQ3D_NMS_DBG_DUMP"] = "1"
        e = model._engine; e.use_graph = False
        model(dict(batches[0]))
