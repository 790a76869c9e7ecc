#!/usr/bin/env python
"""Times aq3d_conv2d_i8 per kernel form (aq3d_conv2d_problem.kernel) on the benchmark's layer shapes; outputs must be bit-identical.
    python scratch/i8_bench.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import numpy as np, torch
from amdq3d import ops
dev = torch.device("cuda:0")
SHAPES = [(128, 128, 188, 188, 1), (256, 128, 188, 188, 1), (128, 256, 188, 188, 2), (256, 256, 94, 94, 1), (512, 64, 188, 188, 1), (64, 64, 188, 188, 1),
          (512, 64, 180, 180, 1)]
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
for Cin, Cout, H, W, stride in SHAPES:
    g_ = torch.Generator().manual_seed(Cin + Cout)
    x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)
    geom = ops.conv_geom(H, W, Cin, Cout, 3, stride, 1, relu=True)
    wk = ops.conv_weight_to_k(w).to(dev)
    params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
    ps, pt = (torch.rand(Cout) + 0.5).to(dev), (torch.randn(Cout) * 0.1).to(dev)
    po = torch.zeros((16, Cout), device=dev)
    out = {}
    ref = None
    rec = {"shape": "%d->%d %dx%d s%d" % (Cin, Cout, H, W, stride)}
    for kernel in (1, 2, 3):
        y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
        fn = lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1, kernel=kernel)
        fn(); torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        rec["same_bits_%d" % kernel] = bool(torch.equal(ref, y))
        rec["us_kernel_%d" % kernel] = round(time_it(fn), 2)
    print(json.dumps(rec), flush=True)
