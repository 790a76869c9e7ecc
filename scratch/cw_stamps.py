"""Where spconv_cw_kernel spends its cycles: builds the library once more with -DAQ3D_STAMPS (in-kernel s_memtime stamps, a separate
.so under scratch/diag/, never loaded by the product), runs one layer per channel count on LiDAR-sheet occupancy and prints the
per-wave averages: prologue (gather-table staging + live mask), pipeline fill, and per unit: A wait + code conversion, MFMA loop
(fragment reads + MFMAs), W' store (incl. the wait for its loads), barrier; epilogue.

    python scratch/cw_stamps.py [--voxels 35000] [--channels 32 64 128]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
import amdq3d  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from amdq3d import build as B, lib  # noqa: E402


def build_stamped():
    out_dir = os.path.join(ROOT, "scratch", "diag", "build_stamps")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(ROOT, "scratch", "diag", "libaq3d_stamps.so")
    objs = []
    for src in B._sources():
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        cmd = [B.HIPCC] + B.CFLAGS + ["-DAQ3D_STAMPS"] + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        if not os.path.isfile(obj) or os.path.getmtime(obj) < os.path.getmtime(src):
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stdout + r.stderr)
        objs.append(obj)
    r = subprocess.run([B.HIPCC, "-shared", "-fPIC", "--offload-arch=%s" % B.ARCH, "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    return out


def dense(path, ops):
    dev = torch.device("cuda:0")
    raw = ctypes.CDLL(path)
    g_ = torch.Generator().manual_seed(3)
    for (H, W, Cin, Cout, s_) in [(188, 188, 128, 128, 1), (188, 188, 256, 128, 1), (94, 94, 256, 256, 1), (188, 188, 512, 64, 1)]:
        x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
        w = torch.randn(Cout, Cin, 3, 3, generator=g_) / (9 * Cin) ** 0.5
        geom = ops.conv_geom(H, W, Cin, Cout, 3, s_, 1, relu=True)
        wk = ops.conv_weight_to_k(w).to(dev)
        params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
        ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
        y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
        po = torch.zeros((16, Cout), device=dev)
        stamps = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
        run = lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1)
        for _ in range(3):
            run()
        assert raw.aq3d_diag_set_i8_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        run()
        t1.record()
        torch.cuda.synchronize()
        raw.aq3d_diag_set_i8_stamps(ctypes.c_void_p(0))
        s = stamps.cpu().numpy()
        s = s[s[:, 7] != 0]
        steps = (s[:, 7] >> 48).astype(np.float64)
        total = (s[:, 7] & ((1 << 48) - 1)).astype(np.float64)
        names = ["prologue", "fill (first step)", "load issue", "LDS reads + MFMA", "wait + quantise + store", "barrier", "epilogue"]
        print("conv2d_i8 %dx%d %d->%d: %d live waves, kernel %.1f us (with stamps); ticks per wave, mean (share of the wave's total):"
              % (H, W, Cin, Cout, len(s), t0.elapsed_time(t1) * 1e3))
        for i, nme in enumerate(names):
            v = s[:, i].astype(np.float64)
            print("  %-24s %9.0f  (%4.1f %%)%s" % (nme, v.mean(), 100 * v.sum() / total.sum(),
                                                 "   per step %.0f" % (v.sum() / np.maximum(steps - 1, 1).sum()) if 2 <= i <= 5 else ""))
        print("  %-24s %9.0f   K steps %.0f" % ("total", total.mean(), steps.mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=35000)
    ap.add_argument("--channels", type=int, nargs="+", default=[32, 64, 128])
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--w16", action="store_true")
    ap.add_argument("--dense", action="store_true", help="conv2d_i8_kernel on the benchmark's BEV layers instead of spconv_cw_kernel")
    args = ap.parse_args()
    path = build_stamped()
    if args.build_only:
        print(path)
        return
    L = lib.load(path)
    from amdq3d import ops
    if args.dense:
        return dense(path, ops)
    import subm_layer as SL
    dev = torch.device("cuda:0")
    shape = [41, 1504, 1504]
    idx = SL.occupancy("sheet", args.voxels)
    m = len(idx)
    count = torch.tensor([m], dtype=torch.int32, device=dev)
    lv = ops.IndexLevel(shape, 1, m, dev)
    ops.level_from_coords(torch.from_numpy(idx).to(dev), lv)
    nbr = ops.subm_neighbors(lv)
    raw = ctypes.CDLL(path)
    for C in args.channels:
        g = torch.Generator().manual_seed(C)
        x = torch.relu(torch.randn(m, C, generator=g)).to(dev)
        wq = torch.randint(-127, 128, (27, C, C), generator=g, dtype=torch.int8).to(dev)
        wdq = (torch.rand(C, generator=g) * 1e-2).to(dev)
        amax = ops.absmax_cols(x, count)
        codes, wprep, out_mul = ops.spconv_prep(x, amax, count, wq, wdq, bits=8, per_tensor=False)
        ct = 64 if C >= 64 else 32
        grid = (m + 127) // 128 * (C // ct) + 8 * (C // ct)
        stamps = torch.zeros((2 * grid + 64, 8, 8), dtype=torch.int64, device=dev)
        assert raw.aq3d_diag_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        for _ in range(3):
            ops.spconv_coded(codes, wprep, out_mul, nbr, count, w16=args.w16)
        torch.cuda.synchronize()
        stamps.zero_()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.spconv_coded(codes, wprep, out_mul, nbr, count, w16=args.w16)
        t1.record()
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 8)
        s = s[s[:, 7] != 0]
        units = (s[:, 7] >> 48).astype(np.float64)
        total = (s[:, 7] & ((1 << 48) - 1)).astype(np.float64)
        names = ["prologue", "fill", "next unit + W load issue", "MFMA loop", "W store", "barrier", "epilogue"]
        print("C = %d, %d rows, %d live waves, kernel %.1f us (with stamps); stamp clock ticks per wave, mean (share of the wave's total):"
              % (C, m, len(s), t0.elapsed_time(t1) * 1e3))
        for i, nme in enumerate(names):
            v = s[:, i].astype(np.float64)
            print("  %-18s %9.0f  (%4.1f %%)%s" % (nme, v.mean(), 100 * v.sum() / total.sum(),
                                               "   per unit %.0f" % (v.sum() / units.sum()) if 2 <= i <= 5 else ""))
        print("  %-18s %9.0f   units per tile %.1f" % ("total", total.mean(), units.mean()))
        raw.aq3d_diag_set_stamps(ctypes.c_void_p(0))


if __name__ == "__main__":
    main()
