"""Microbenchmark of the per-layer SmoothQuant prep kernels (piece_absmax, sq_prep) at BEV-backbone shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amdq3d import ops
dev = torch.device("cuda:0")
def timeit(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W, C, Cout) in [(188, 188, 128, 128), (188, 188, 256, 128), (94, 94, 256, 256), (188, 188, 384, 64), (188, 188, 64, 64)]:
    x = torch.randn(H, W, C, device=dev)
    pieces = torch.zeros(16, C, device=dev)
    t_p = timeit(lambda: ops.piece_absmax(x, H, W, C, C, pieces))
    w = torch.randn(Cout, 9 * C, device=dev) * 0.05
    wc = w.abs().max(dim=0)[0].contiguous()
    g = ops.conv_geom(H, W, C, Cout, 3, 1, 1)
    params = torch.empty((ops.sq_param_bytes(C, Cout, 3),), dtype=torch.uint8, device=dev)
    acc = torch.zeros(1, device=dev)
    t_s = timeit(lambda: ops.sq_prep(pieces, w, wc, g, 0.5, params, 8, 8, 0.0, None, acc))
    print("H%d W%d C%d Cout%d: piece_absmax %.1f us (%.0f GB/s)  sq_prep %.1f us" % (H, W, C, Cout, t_p, H * W * C * 4 / t_p / 1e3, t_s), flush=True)
