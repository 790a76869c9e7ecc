"""What bounds conv2d_i8_kernel: knock-out builds (-DAQ3D_KO=mask, see csrc/conv2d.hip; timing only, results are wrong) of the product
sources into scratch/diag/libaq3d_ko<mask>.so, each timed on the benchmark's BEV layers, plus where and when every workgroup ran
(XCC / CU ids and 100 MHz wall-clock stamps written by the kernel) for the unmodified loop (mask 0).

    python scratch/i8_knockout.py --build-only      # in the build container
    python scratch/i8_knockout.py                   # on the GPU box
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
MASKS = [0, 1, 2, 4, 8, 16, 32, 3, 7, 15, 47, 63]
NAMES = {0: "none", 1: "s|1/s", 2: "X loads", 4: "W loads", 8: "quantise", 16: "7 of 8 MFMAs", 32: "code stores"}


def build(masks, exp=0):
    import amdq3d  # noqa: F401
    from amdq3d import build as B
    out_dir = os.path.join(ROOT, "scratch", "diag", "build_ko")
    os.makedirs(out_dir, exist_ok=True)
    B.build(verbose=False)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B._sources() if not s.endswith("conv2d.hip")]
    src = os.path.join(B.CSRC, "conv2d.hip")
    paths = {}
    for m in masks:
        tag = "ko%d" % m + ("_exp%d" % exp if exp else "")
        obj = os.path.join(out_dir, "conv2d_%s.o" % tag)
        out = os.path.join(ROOT, "scratch", "diag", "libaq3d_%s.so" % tag)
        if not os.path.isfile(out) or os.path.getmtime(out) < os.path.getmtime(src):
            r = subprocess.run([B.HIPCC] + B.CFLAGS + ["-DAQ3D_KO=%d" % m, "-DAQ3D_EXP=%d" % exp, "-c", src, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stdout + r.stderr)
            r = subprocess.run([B.HIPCC, "-shared", "-fPIC", "--offload-arch=%s" % B.ARCH, "-o", out, obj] + others, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stdout + r.stderr)
        paths[(m, exp)] = out
    return paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--masks", type=int, nargs="+", default=MASKS)
    ap.add_argument("--exp", type=int, nargs="+", default=[0], help="AQ3D_EXP switch sets (4 non-temporal output stores, 8 no row-register kernel), each built with every mask")
    args = ap.parse_args()
    paths = {}
    for e in args.exp:
        paths.update(build(args.masks, e))
    if args.build_only:
        print("\n".join(paths.values()))
        return
    import numpy as np
    import torch
    from amdq3d import lib
    from conv2d_scaling import timed
    dev = torch.device("cuda:0")
    cases = [(188, 188, 128, 128, False), (188, 188, 128, 128, True), (94, 94, 256, 256, False), (188, 188, 256, 128, False)]
    for (m, e) in paths:
        lib._lib = None
        lib.load(paths[(m, e)])
        from amdq3d import ops
        raw = ctypes.CDLL(paths[(m, e)])
        g_ = torch.Generator().manual_seed(3)
        for (H, W, Cin, Cout, pieces) in cases:
            x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
            w = torch.randn(Cout, Cin, 3, 3, generator=g_) / (9 * Cin) ** 0.5
            geom = ops.conv_geom(H, W, Cin, Cout, 3, 1, 1, relu=True)
            wk = ops.conv_weight_to_k(w).to(dev)
            params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
            ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
            y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
            po = torch.zeros((16, Cout), device=dev) if pieces else None
            run = lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1 if pieces else 0)
            med, lo, hi = timed(run)
            rec = dict(knocked_out=[NAMES[b] for b in NAMES if b and (m & b)] or ["none"], mask=m, exp=e, H=H, Cin=Cin, Cout=Cout, pieces=pieces,
                       us=round(med, 2), us_p10=round(lo, 2))
            if m == 0:
                place = torch.zeros((4096, 4), dtype=torch.int64, device=dev)
                assert raw.aq3d_diag_set_i8_place(ctypes.c_void_p(place.data_ptr())) == 0
                run()
                torch.cuda.synchronize()
                raw.aq3d_diag_set_i8_place(ctypes.c_void_p(0))
                p = place.cpu().numpy()
                p = p[p[:, 3] != 0]
                cu = (p[:, 0] & 15) * 4096 + ((p[:, 1] >> 8) & 0xff)        # XCC, (SE, SH, CU) of HW_ID
                t0 = p[:, 2].min()
                per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
                dur = (p[:, 3] - p[:, 2]) / 100.0
                start = (p[:, 2] - t0) / 100.0
                rec.update(workgroups=int(len(p)), cus_used=int(len(per_cu)), wgs_per_cu_hist=np.bincount(per_cu).tolist(),
                           wg_us_mean=round(float(dur.mean()), 2), wg_us_p10=round(float(np.percentile(dur, 10)), 2),
                           wg_us_p90=round(float(np.percentile(dur, 90)), 2), start_us_p50=round(float(np.median(start)), 2),
                           start_us_p90=round(float(np.percentile(start, 90)), 2), start_us_max=round(float(start.max()), 2),
                           span_us=round(float((p[:, 3].max() - t0) / 100.0), 2))
                # duration by how many workgroups shared the CU
                inv = np.unique(cu, return_inverse=True)[1]
                rec["wg_us_by_cu_load"] = {int(k): round(float(dur[per_cu[inv] == k].mean()), 2) for k in np.unique(per_cu)}
                end = (p[:, 3] - t0) / 100.0
                late = end > end.max() - 2.0
                rec.update(wg_us_max=round(float(dur.max()), 2), wg_us_p99=round(float(np.percentile(dur, 99)), 2),
                           end_us_p50=round(float(np.median(end)), 2), end_us_p90=round(float(np.percentile(end, 90)), 2),
                           last_2us_wgs=int(late.sum()), last_2us_cu_load_hist=np.bincount(per_cu[inv][late]).tolist(),
                           last_2us_xcc_hist=np.bincount((p[:, 0] & 15)[late], minlength=8).tolist())
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
