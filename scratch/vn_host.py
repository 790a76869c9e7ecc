import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfgp, npts, nvox, _ = bench.WORKLOADS["voxelnext"]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
batches = []
for i in range(4):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
with torch.no_grad():
    for i in range(12):
        model.forward_async(dict(batches[i % 4])).counts()
    torch.cuda.synchronize()
    eng = model.engine()
    print("graphs", sorted(map(str, eng._graphs)), "steps", len(eng.steps), flush=True)
    # host enqueue time vs GPU time
    t = time.time(); ps = []
    for i in range(20):
        t0 = time.time(); p = model.forward_async(dict(batches[i % 4])); t1 = time.time()
        ps.append((t1 - t0, p)); p.counts()
    print("serial: host enqueue ms avg %.2f, frame ms %.2f" % (1e3 * sum(a for a, _ in ps) / 20, 1e3 * (time.time() - t) / 20), flush=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = eng.stream if getattr(eng, "stream", None) is not None else torch.cuda.current_stream()
    ev0.record(s); p = model.forward_async(dict(batches[0])); ev1.record(s); p.counts(); torch.cuda.synchronize()
    print("gpu ms of one frame on its stream %.2f" % ev0.elapsed_time(ev1))
