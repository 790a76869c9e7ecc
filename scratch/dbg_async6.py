import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
eng = model.engine()
def sums():
    torch.cuda.synchronize()
    out = [("vfe", float(eng.vfe_feats.double().abs().sum()))]
    for r in eng.sparse_convs:
        n = int(eng.levels[r["level"]]["count"].item())
        out.append((r["out"]["name"], float(r["out"]["buf"][:n].double().abs().sum())))
    out.append(("spatial", float(eng.spatial.double().abs().sum())))
    for i, r in enumerate(eng.dense_convs):
        out.append(("dense%d" % i, float(r["out"].double().abs().sum())))
    return out
with torch.no_grad():
    for i in range(5):
        r = model(dict(batches[0]))
    ref = sums()
    print("graphs", sorted(map(str, eng._graphs)), "boxes", len(r[0][0]["pred_scores"]))
    junk = [torch.full((1 << 12,), float("nan"), device=dev) for _ in range(20000)]
    torch.cuda.synchronize()
    seen = {}
    def walk(o, path, depth=0):
        if depth > 6 or id(o) in seen: return
        seen[id(o)] = 1
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                st = o.untyped_storage()
                found.append((st.data_ptr(), st.data_ptr() + st.nbytes(), path))
        elif isinstance(o, dict):
            for k, v in o.items(): walk(v, path + "." + str(k), depth + 1)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o): walk(v, path + "[%d]" % i, depth + 1)
        elif hasattr(o, "__dict__") and not isinstance(o, torch.nn.Module) and type(o).__module__.startswith(("amdq3d", "quantization")):
            for k, v in vars(o).items(): walk(v, path + "." + k, depth + 1)
    found = []
    walk(eng, "eng")
    ji = sorted((j.data_ptr(), j.data_ptr() + j.numel() * 4) for j in junk)
    import bisect
    starts = [a for a, _ in ji]
    nover = 0
    for a, b, path in found:
        k = bisect.bisect_right(starts, a) - 1
        for kk in (k, k + 1):
            if 0 <= kk < len(ji) and ji[kk][0] < b and a < ji[kk][1]:
                nover += 1
                if nover < 30: print("OVERLAP", path, hex(a), b - a)
                break
    print("tensors", len(found), "overlapping junk:", nover)
    eng.use_graph = False
    r = model(dict(batches[0]))
    print("eager after junk boxes", len(r[0][0]["pred_scores"]))
    eng.use_graph = True
    r = model(dict(batches[0])); r = model(dict(batches[0]))
    torch.cuda.synchronize()
    big = torch.cat([j.view(torch.int32) for j in junk])
    mod = (big != 0x7fc00000)
    print("junk words modified by graph replay:", int(mod.sum()), "of", big.numel())
    idx = mod.nonzero().flatten()
    if idx.numel():
        # contiguous runs
        d = (idx[1:] - idx[:-1]) > 1024
        starts = torch.cat([idx[:1], idx[1:][d]]); ends = torch.cat([idx[:-1][d], idx[-1:]])
        for a, b in list(zip(starts.tolist(), ends.tolist()))[:40]:
            seg = big[a:a + 8]
            print("  run words %d..%d (%d KB) junk#%d ptr %s ints %s floats %s" % (a, b, (b - a) * 4 // 1024, a // 4096, hex(junk[a // 4096].data_ptr()), seg.tolist()[:4], seg.view(torch.float32).tolist()[:4]))
    got = sums()
    print("after junk boxes", len(r[0][0]["pred_scores"]))
    for (n, a), (_, b) in zip(ref, got):
        print("%-28s %.6e %.6e %s" % (n, a, b, "" if a == b else "<<<< DIFF"))
