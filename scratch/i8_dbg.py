import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import numpy as np, torch
from amdq3d import ops
dev = torch.device("cuda:0")
def time_it(fn, iters=50):
    t0 = time.time()
    while time.time() - t0 < 0.05:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
Cin, Cout, H, W = 128, 128, 188, 188
g_ = torch.Generator().manual_seed(1)
x = (torch.relu(torch.randn(H, W, Cin, generator=g_))).to(dev)
w = torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)
wk = ops.conv_weight_to_k(w).to(dev)
for relu, kernel, name in ((True, 1, "legacy"), (True, 2, "new full"), (True, 3, "new, Q skipped"), (False, 2, "new, M skipped"), (False, 3, "new, weights from registers (no global weight loads)")):
    geom = ops.conv_geom(H, W, Cin, Cout, 3, 1, 1, relu=relu)
    params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
    y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
    fn = lambda: ops.conv2d_i8(x, params, geom, out=y, kernel=kernel)
    print(name, round(time_it(fn), 2), "us", flush=True)
