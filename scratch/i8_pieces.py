"""What the piece-maxima side output costs conv2d_i8_kernel: the BEV layers of the bench with pieces_out = None / merged interior
(pieces_mode 1) / exact 16 classes (mode 0).   python scratch/i8_pieces.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import torch
from amdq3d import ops

dev = torch.device("cuda:0")
g_ = torch.Generator().manual_seed(3)


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (H, W, Cin, Cout, s_) in [(188, 188, 128, 128, 1), (94, 94, 256, 256, 1), (188, 188, 256, 128, 1)]:
    x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / (9 * Cin) ** 0.5
    geom = ops.conv_geom(H, W, Cin, Cout, 3, s_, 1, relu=True)
    wk = ops.conv_weight_to_k(w).to(dev)
    params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
    ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
    y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
    po = torch.zeros((16, Cout), device=dev)
    t_none = timed(lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y))
    t_m1 = timed(lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1))
    po.zero_()
    t_m1z = timed(lambda: (po.zero_(), ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1)))
    t_m0 = timed(lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=0))
    t_m0z = timed(lambda: (po.zero_(), ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=0)))
    print("%dx%d %d->%d: no pieces %.1f us | merged interior %.1f (zeroed each time: %.1f) | 16 classes %.1f (zeroed: %.1f)"
          % (H, W, Cin, Cout, t_none, t_m1, t_m1z, t_m0, t_m0z))
