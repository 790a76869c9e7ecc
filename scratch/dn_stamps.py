"""Where the single-workgroup tail of a frame (top-K select, decode, NMS sweep) spends its time: builds the library with -DAQ3D_STAMPS
(scratch/diag/libaq3d_stamps.so, never loaded by the product), runs decode + NMS on a 188 x 188 x 3 head shaped like the bench's
random-weight head (narrow-band logits around the bias; scratch/hm_probe.py) and prints the 10 ns wall-clock deltas between the stamps.

    python scratch/dn_stamps.py [--peaks]      (--peaks: a trained-looking map, few hundred scores above the threshold)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scratch"))
import amdq3d  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from amdq3d import lib  # noqa: E402
import cw_stamps  # noqa: E402

NAMES = {0: ["counts+hist sum", "pivot", "zero sort buffer", "partition pass", "sort + take", "write-out"],
         1: ["load candidates", "sort", "gather + box", "block scan", "store"],
         2: ["count load", "mask -> LDS", "block 0: chain", "block 0: keep", "block 0: remv", "other blocks"]}
KERNEL = {0: "topk_select_kernel", 1: "decode_kernel", 2: "nms_sweep_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--peaks", action="store_true")
    args = ap.parse_args()
    path = cw_stamps.build_stamped()
    lib.load(path)
    from amdq3d import ops
    raw = ctypes.CDLL(path)
    dev = torch.device("cuda:0")
    H = W = 188
    rng = np.random.default_rng(0)
    buf = np.zeros((H * W, 24), np.float32)
    if args.peaks:
        buf[:, 0:3] = rng.normal(-6, 1, (H * W, 3))
        for _ in range(400):
            buf[rng.integers(0, H * W), rng.integers(0, 3)] = rng.uniform(-1, 4)
    else:
        for c, (mu, sd) in enumerate([(-2.052, 0.134), (-2.254, 0.090), (-2.077, 0.142)]):
            buf[:, c] = rng.normal(mu, sd, H * W)
    buf[:, 4:6] = rng.uniform(0, 1, (H * W, 2))
    buf[:, 6] = rng.uniform(-1.5, 3.5, H * W)
    buf[:, 8:11] = rng.normal(0.5, 0.4, (H * W, 3))
    buf[:, 12:14] = rng.normal(0, 1, (H * W, 2))
    t = torch.from_numpy(buf).to(dev)
    pc = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    cfg = ops.make_decode_cfg(H, W, 3, 500, False, 8, [0.1, 0.1], pc, pc, 0.1, 24, 24)
    stamps = torch.zeros((4, 16), dtype=torch.int64, device=dev)

    def run():
        boxes, scores, labels, count = ops.center_decode(t, t[:, 4:], t[:, 6:], t[:, 8:], t[:, 12:], None, cfg)
        keep, kc = ops.nms_rotated(boxes, count, 0.7, 500)
        return count, kc

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    assert raw.aq3d_diag_set_dn_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
    count, kc = run()
    torch.cuda.synchronize()
    raw.aq3d_diag_set_dn_stamps(ctypes.c_void_p(0))
    s = stamps.cpu().numpy()
    print("decoded %d boxes, kept %d" % (int(count.item()), int(kc.item())))
    for k in (0, 1, 2):
        row = s[k]
        n = len(NAMES[k])
        print("%s: %.1f us between first and last stamp" % (KERNEL[k], (row[n] - row[0]) / 100.0))
        for i, nm in enumerate(NAMES[k]):
            print("   %-20s %7.2f us" % (nm, (row[i + 1] - row[i]) / 100.0))


if __name__ == "__main__":
    main()
