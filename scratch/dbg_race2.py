"""Engine A replays its main phase on fixed index data while engine B runs whole frames beside it; every A pass is
checksummed stage by stage against the first one, to find the kernel whose result depends on what runs next to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
def sums(e):
    out = [("counts", tuple(int(l["count"].item()) for l in e.levels))]
    n0 = int(e.levels[0]["count"].item())
    out.append(("vfe", float(e.vfe_feats[:n0].double().sum())))
    w = torch.arange(1, e.cap + 1, device=dev, dtype=torch.float64)
    for li, l in enumerate(e.levels):
        n = int(l["count"].item())
        out.append(("coords%d" % li, float((l["coords"][:n].double().sum(1) * w[:n]).sum())))
        out.append(("nbr%d" % li, float((l["nbr"][:n].double().sum(1) * w[:n]).sum())))
        if "down" in l:
            out.append(("down_nbr%d" % li, float((l["down"]["nbr"][:n].double().sum(1) * w[:n]).sum())))
    for r in e.sparse_convs:
        n = int(e.levels[r["level"]]["count"].item())
        out.append((r["out"]["name"], float(r["out"]["buf"][:n].double().sum())))
    out.append(("spatial", float(e.spatial.double().sum())))
    for i, r in enumerate(e.dense_convs):
        out.append(("dense%d" % i, float(r["out"].double().sum())))
    for hi, h in enumerate(e.heads):
        nk = h["cfg"].num_classes * h["cfg"].K
        ws = h["dec_ws"][: nk * 8].view(torch.int64)
        out.append(("topk_keys_sum%d" % hi, int((ws >> 8).sum())))
        out.append(("topk_nonzero%d" % hi, int((ws != 0).sum())))
        c = int(h["cand_count"][0].item())
        out.append(("cand_count%d" % hi, c))
        out.append(("cand_scores%d" % hi, float(h["cand_scores"][0][:c].double().sum())))
        out.append(("cand_boxes%d" % hi, float(h["cand_boxes"][0][:c].double().sum())))
        wv = torch.arange(1, c + 1, device=dev, dtype=torch.float64)
        out.append(("cand_scores_ord%d" % hi, float((h["cand_scores"][0][:c].double() * wv).sum())))
        out.append(("cand_boxes_ord%d" % hi, float((h["cand_boxes"][0][:c].double().sum(1) * wv).sum())))
        out.append(("cand_labels_ord%d" % hi, float((h["cand_labels"][0][:c].double() * wv).sum())))
        capn = min(h["K"], h["pre"]); cb = (capn + 63) // 64
        m = h["nms_ws"][: capn * cb * 8].view(torch.int64).view(capn, cb)[:c]
        out.append(("mask%d" % hi, int((m >> 4).sum())))
        out.append(("mask_pop%d" % hi, int((m != 0).sum())))
        kc = int(e.det_counts[e.slot, 0, hi].item())
        out.append(("keep%d" % hi, int(h["keep"][0][:kc].long().sum())))
    out.append(("det", tuple(e.det_counts[e.slot].flatten().tolist())))
    return out
with torch.no_grad():
    for i in range(12):
        model.forward_async(dict(batches[i % 4])).counts()
    A, Bn = model._engine, model._extra_engines[0]
    torch.cuda.synchronize()
    sA, mA = A.stream, A.main_stream
    with torch.cuda.stream(sA):
        nb = A.begin(dict(batches[0])); A.run_index(dict(batches[0]), nb)
    torch.cuda.synchronize()
    evA = torch.cuda.Event()
    def passA():
        with torch.cuda.stream(sA):
            A.run_index(dict(batches[0]), nb); evA.record(sA)
        mA.wait_event(evA)
        with torch.cuda.stream(mA):
            A.run_main(nb)
    passA(); torch.cuda.synchronize()
    def maskof(e):
        h = e.heads[0]; c = int(h["cand_count"][0].item())
        capn = min(h["K"], h["pre"]); cbn = (capn + 63) // 64
        return h["nms_ws"][: capn * cbn * 8].view(torch.int64).view(capn, cbn)[:c].clone(), h["cand_boxes"][0][:c].clone()
    with torch.cuda.stream(mA):
        ref = sums(A)
        ref_mask, ref_boxes = maskof(A)
    print("ref ready; graphs A", sorted(map(str, A._graphs)), flush=True)
    noise = os.environ.get("NOISE", "1") == "1"
    bad = 0
    iters = int(os.environ.get("ITERS", "1500"))
    for it in range(iters):
        if noise:
            for k in range(2):
                Bn._ev_done[Bn.slot].synchronize() if k == 0 else None
                with torch.cuda.stream(Bn.stream):
                    nbb = Bn.begin(dict(batches[(it + k) % 4])); Bn.run_index(dict(batches[(it + k) % 4]), nbb); Bn.ev_idx.record(Bn.stream)
                Bn.main_stream.wait_event(Bn.ev_idx)
                with torch.cuda.stream(Bn.main_stream):
                    Bn.run_main(nbb)
        passA()
        mA.synchronize()
        with torch.cuda.stream(mA):
            s = sums(A)
        d = [a[0] for a, c in zip(ref, s) if a != c]
        if d:
            bad += 1
            if bad <= 10:
                print("iter", it, "first diffs", d[:5], flush=True)
                with torch.cuda.stream(mA):
                    m, bx = maskof(A)
                    if m.shape == ref_mask.shape:
                        dd = (m != ref_mask).nonzero()
                        for r_, c_ in dd[:4].tolist():
                            print("   row", r_, "cb", c_, "got %016x ref %016x" % (int(m[r_, c_]) & (2**64 - 1), int(ref_mask[r_, c_]) & (2**64 - 1)), "boxes equal", bool(torch.equal(bx, ref_boxes)), flush=True)
    torch.cuda.synchronize()
    print("noise", noise, "bad", bad, "of", iters)
    os.environ["AQ3D_NMS_DBG_DUMP"] = "1"
    A.use_graph = False
    passA(); torch.cuda.synchronize()
