"""Time the wide sparse-conv kernel against the number of active rows (tile quantisation over 256 CUs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch
from amdq3d import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
cin = int(os.environ.get("CIN", "128")); cout = int(os.environ.get("COUT", "128")); kvol = 27
cap = 70000
feats = (torch.randn(cap, 2 * cin, device=dev) * 0.5).to(torch.float16)
w = (torch.randint(-127, 128, (kvol, cout, cin), device=dev)).to(torch.float16)
wdq = torch.rand(cout, device=dev) * 0.01
out = torch.zeros(cap, cout, device=dev)
amax = torch.zeros(cout, device=dev)
for n in [int(a) for a in os.environ.get("NS", "16384,30000,32768,33000,34960,40000,49152,65536").split(",")]:
    # ~55% neighbour occupancy like conv4
    nbr = torch.randint(0, n, (cap, kvol), device=dev, dtype=torch.int32)
    nbr[torch.rand(cap, kvol, device=dev) > 0.55] = -1
    nbr[:, 13] = torch.arange(cap, device=dev, dtype=torch.int32)
    count = torch.tensor([n], device=dev, dtype=torch.int32)
    f = lambda: ops.spconv(feats, w, nbr, count, cout, w_dequant=wdq, relu=True, absmax_out=amax, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tiles = (n + 127) // 128
    print("n %6d tiles %4d  %.1f us   %.2f us/tile-round(256)  TFLOP/s(split) %.0f" % (n, tiles, us, us / ((tiles + 255) // 256), tiles * 27 * 128 * cin * cout * 2 * 2 / us / 1e6), flush=True)
