import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amdq3d import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
cap, kvol = 65536, 27
def mk(cin, cout):
    m = 60000
    feats = torch.randn(cap, cin, device=dev)
    w = torch.randn(kvol, cout, cin, device=dev) * 0.1
    nbr = torch.randint(-1, m, (cap, kvol), device=dev, dtype=torch.int32)
    count = torch.tensor([m], device=dev, dtype=torch.int32)
    ps = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev) * 0.1
    out = torch.zeros(cap, cout, device=dev)
    amax = torch.zeros(cout, device=dev)
    return dict(feats=feats, w=w, nbr=nbr, count=count, ps=ps, sh=sh, out=out, amax=amax, cout=cout)
cases = {"stem5x16": mk(5, 16), "direct16x16": mk(16, 16), "wide64x64": mk(64, 64)}
s = torch.cuda.Stream()
graphs = {}
def run(c):
    ops.spconv(c["feats"], c["w"], c["nbr"], c["count"], c["cout"], post_scale=c["ps"], post_shift=c["sh"], relu=True,
               absmax_out=c["amax"], out=c["out"])
ref = {}
with torch.cuda.stream(s):
    for k, c in cases.items():
        run(c); torch.cuda.synchronize()
        ref[k] = float(c["out"].double().abs().sum()); c["out"].zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run(c)
        graphs[k] = g
        g.replay(); torch.cuda.synchronize()
        print(k, "eager", ref[k], "graph", float(c["out"].double().abs().sum()))
junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(500)]
junk += [torch.full((1 << 12,), float("nan"), device=dev) for _ in range(20000)]
junk += [torch.full((16,), float("nan"), device=dev) for _ in range(20000)]
torch.cuda.synchronize()
with torch.cuda.stream(s):
    for k, c in cases.items():
        c["out"].zero_()
        graphs[k].replay(); torch.cuda.synchronize()
        a = float(c["out"].double().abs().sum())
        print(k, "graph after junk", a, "SAME" if a == ref[k] else "DIFFERENT")
