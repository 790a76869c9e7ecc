"""What bounds spconv_cw_kernel (the quantised sparse convs with Cin >= 32): knock-out builds (-DAQ3D_KO_CW=mask, see
csrc/spconv_coded.hip; timing only, results are wrong) of the product sources into scratch/diag/libaq3d_kocw<mask>.so, each timed on one
layer per channel count over a LiDAR-sheet index level in device (class-sorted) row order, plus where and when every live workgroup ran.

    python scratch/cw_knockout.py --build-only      # in the build container
    python scratch/cw_knockout.py [--voxels 38000] [--channels 128 64]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
MASKS = [0, 1, 2, 4, 12, 16, 3, 7, 15, 31]
NAMES = {1: "weight planes", 2: "code gathers", 4: "MFMAs", 8: "fragment reads", 16: "output stores"}


def build(masks):
    import amdq3d  # noqa: F401
    from amdq3d import build as B
    out_dir = os.path.join(ROOT, "scratch", "diag", "build_ko")
    os.makedirs(out_dir, exist_ok=True)
    B.build(verbose=False)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B._sources() if not s.endswith("spconv_coded.hip")]
    src = os.path.join(B.CSRC, "spconv_coded.hip")
    paths = {}
    for m in masks:
        obj = os.path.join(out_dir, "spconv_coded_ko%d.o" % m)
        out = os.path.join(ROOT, "scratch", "diag", "libaq3d_kocw%d.so" % m)
        r = subprocess.run([B.HIPCC] + B.CFLAGS + ["-DAQ3D_KO_CW=%d" % m, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        r = subprocess.run([B.HIPCC, "-shared", "-fPIC", "--offload-arch=%s" % B.ARCH, "-o", out, obj] + others, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        paths[m] = out
    return paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--masks", type=int, nargs="+", default=MASKS)
    ap.add_argument("--voxels", type=int, default=38000)
    ap.add_argument("--channels", type=int, nargs="+", default=[128, 64])
    ap.add_argument("--w16", action="store_true")
    args = ap.parse_args()
    if args.build_only:
        print("\n".join(build(args.masks).values()))
        return
    paths = {m: os.path.join(ROOT, "scratch", "diag", "libaq3d_kocw%d.so" % m) for m in args.masks}
    import numpy as np
    import torch
    import amdq3d  # noqa: F401
    from amdq3d import lib
    from conv2d_scaling import timed
    import subm_layer as SL
    dev = torch.device("cuda:0")
    idx = SL.occupancy("sheet", args.voxels)
    m_rows = len(idx)
    for m in args.masks:
        lib._lib = None
        lib.load(paths[m])
        from amdq3d import ops
        raw = ctypes.CDLL(paths[m])
        count = torch.tensor([m_rows], dtype=torch.int32, device=dev)
        lv = ops.IndexLevel([41, 1504, 1504], 1, m_rows, dev)
        ops.level_from_coords(torch.from_numpy(idx).to(dev), lv)
        nbr = ops.subm_neighbors(lv)
        for C in args.channels:
            g = torch.Generator().manual_seed(C)
            x = torch.relu(torch.randn(m_rows, C, generator=g)).to(dev)
            wq = torch.randint(-127, 128, (27, C, C), generator=g, dtype=torch.int8).to(dev)
            wdq = (torch.rand(C, generator=g) * 1e-2).to(dev)
            amax = ops.absmax_cols(x, count)
            codes, wprep, out_mul = ops.spconv_prep(x, amax, count, wq, wdq, bits=8, per_tensor=False)
            run = lambda: ops.spconv_coded(codes, wprep, out_mul, nbr, count, w16=args.w16)
            med, lo, hi = timed(run)
            rec = dict(knocked_out=[NAMES[b] for b in NAMES if m & b] or ["none"], mask=m, C=C, rows=m_rows, us=round(med, 2), us_p10=round(lo, 2))
            ct = 64 if C >= 64 else 32
            grid = ((m_rows + 127) // 128 + 7) // 8 * 8 * (C // ct)
            place = torch.zeros((grid + 64, 4), dtype=torch.int64, device=dev)
            assert raw.aq3d_diag_set_cw_place(ctypes.c_void_p(place.data_ptr())) == 0
            run()
            torch.cuda.synchronize()
            raw.aq3d_diag_set_cw_place(ctypes.c_void_p(0))
            p = place.cpu().numpy()
            p = p[p[:, 3] != 0]
            cu = (p[:, 0] & 15) * 4096 + ((p[:, 1] >> 8) & 0xff)
            inv = np.unique(cu, return_inverse=True)[1]
            per_cu = np.bincount(inv)
            t0 = p[:, 2].min()
            dur, start, end = (p[:, 3] - p[:, 2]) / 100.0, (p[:, 2] - t0) / 100.0, (p[:, 3] - t0) / 100.0
            late = end > end.max() - 2.0
            rec.update(workgroups=int(len(p)), cus_used=int(len(per_cu)), wgs_per_cu_hist=np.bincount(per_cu).tolist(),
                       wg_us_mean=round(float(dur.mean()), 2), wg_us_p10=round(float(np.percentile(dur, 10)), 2),
                       wg_us_p90=round(float(np.percentile(dur, 90)), 2), wg_us_max=round(float(dur.max()), 2),
                       start_us_p50=round(float(np.median(start)), 2), start_us_p90=round(float(np.percentile(start, 90)), 2),
                       start_us_max=round(float(start.max()), 2), end_us_p50=round(float(np.median(end)), 2),
                       end_us_p90=round(float(np.percentile(end, 90)), 2), span_us=round(float(end.max()), 2),
                       last_2us_wgs=int(late.sum()), late_start_wgs=int((start > 2.0).sum()),
                       wg_us_late_starters=round(float(dur[start > 2.0].mean()), 2) if (start > 2.0).any() else None)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
