import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
batches = []
for i in range(8):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
with torch.no_grad():
    for i in range(32):
        model.forward_async(dict(batches[i % 8])).counts()
    torch.cuda.synchronize()
    pend = collections.deque(); t_enq = 0.0; t_wait = 0.0; n = 400
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        pend.append(model.forward_async(dict(batches[i % 8])))
        b_ = time.perf_counter(); t_enq += b_ - a
        if len(pend) >= 4:
            pend.popleft().counts()
        t_wait += time.perf_counter() - b_
    while pend: pend.popleft().counts()
    tot = time.perf_counter() - t0
    print("frames/s %.1f  per frame: total %.3f ms, forward_async (incl. its own wait for the plan) %.3f ms, counts() wait %.3f ms" % (n / tot, 1e3 * tot / n, 1e3 * t_enq / n, 1e3 * t_wait / n))
    # pure enqueue cost: enqueue 4 frames on an idle GPU
    torch.cuda.synchronize(); a = time.perf_counter()
    ps = [model.forward_async(dict(batches[i])) for i in range(4)]
    b_ = time.perf_counter()
    for p in ps: p.counts()
    print("enqueue of 4 frames on an idle GPU: %.3f ms per frame" % (1e3 * (b_ - a) / 4))
