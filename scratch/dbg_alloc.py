import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
b = ds.collate_batch([ds[0]]); load_data_to_gpu(b, dev)
eng = model.engine()
def st():
    m = torch.cuda.memory_stats()
    return m["allocation.all.allocated"], m["segment.all.allocated"], m["segment.all.freed"], m["reserved_bytes.all.current"]
with torch.no_grad():
    print("start", st())
    for i in range(4):
        a = st()
        model(dict(b))
        print("frame", i, "graphs", len(eng._graphs), "delta allocs", st()[0] - a[0], st())
    a = st()
    B = eng.begin(dict(b)); eng._graph_now = False
    eng.run_index(dict(b), B)
    print("eager idx allocs", st()[0] - a[0]); a = st()
    eng.run_main(B)
    print("eager main allocs", st()[0] - a[0])
    torch.cuda.synchronize()
