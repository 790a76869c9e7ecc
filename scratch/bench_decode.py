"""Microbenchmark of aq3d_center_decode / NMS chain at head-map sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amdq3d import ops
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W, ncls, thr) in [(32, 32, 3, 0.1), (94, 94, 3, 0.1), (188, 188, 3, 0.1), (188, 188, 1, 0.1), (188, 188, 3, -1.0)]:
    LD = 24
    buf = torch.randn(H * W, LD, device=dev) * 1.5 - 2.0
    cfg = ops.make_decode_cfg(H, W, ncls, 500, False, 8, [0.1, 0.1], [-75.2, -75.2], [-80, -80, -10, 80, 80, 10], thr, LD, LD)
    out = None
    def f():
        return ops.center_decode(buf[:, 0:], buf[:, 4:], buf[:, 8:], buf[:, 12:], buf[:, 16:], None, cfg)
    b, s, l, c = f()
    t = timeit(f)
    n = int(c.item())
    keep = None
    def g():
        return ops.nms_rotated(b[:500], c, 0.7, 500)
    t2 = timeit(g)
    print("H%d W%d ncls%d thr%.1f: decode %.1f us (%d candidates)  nms %.1f us" % (H, W, ncls, thr, t, n, t2), flush=True)
