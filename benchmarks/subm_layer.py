#!/usr/bin/env python
"""BASELINE configs[1]: ONE SubMConv3d 3x3x3 layer, int8 weights x per-channel-quantised activations, on the MI355X
gather-GEMM kernel vs the CPU oracle — plus the voxel-count sweep (30 k → 1 M active voxels) that shows where the
kernel sits against the HBM / MFMA roofs once launch latency stops dominating (SURVEY.md §7 hard part 1, §8d).

    python benchmarks/subm_layer.py [--channels 16 64 128] [--voxels 30000 120000 480000 1000000] [--occupancy sheet|w150k]

Prints one JSON line per (channels, voxels). Timed region: aq3d_spconv_prep (per-channel quantiser to int8 codes + per-frame
weight copy) + aq3d_spconv_coded (fused bias/BN/ReLU epilogue) with the gather table already built, HIP events over `--iters`
launches; the two kernels are also timed separately. Algorithmic bytes: fp32 rows in + codes out (quantiser), codes + gather
table + int8 weights in, fp32 rows out (conv).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_GBS, I8_TOPS = 8000.0, 5000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s; dense int8 MFMA = 2 x the 2.5 PF of bf16


def _morton_key(idx):
    """Z-curve over (y>>3, x>>3) tiles, then z, y, x inside — the row order a spatial sort of the voxels would give."""
    ty, tx = idx[:, 2].astype(np.int64) >> 3, idx[:, 3].astype(np.int64) >> 3
    key = np.zeros(len(idx), np.int64)
    for b in range(9):
        key |= ((tx >> b) & 1) << (2 * b)
        key |= ((ty >> b) & 1) << (2 * b + 1)
    inner = ((idx[:, 1].astype(np.int64) * 8) + (idx[:, 2] & 7)) * 8 + (idx[:, 3] & 7)
    return key * 4096 + inner


def occupancy(kind, m, seed=4, order="random"):
    """i32 [m, 4] active sites (b, z, y, x) inside a [41, 1504, 1504] grid."""
    rng = np.random.default_rng(seed)
    if kind == "sheet":          # contiguous ground sheet, 3 z-levels: ~13 of 27 neighbours active (LiDAR-like surfaces)
        side = int(np.ceil(np.sqrt(m / 1.5)))
        ys, xs = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
        base = np.stack([np.zeros(side * side, np.int64), np.full(side * side, 12), ys.ravel() + 8, xs.ravel() + 8], 1)
        up = base[rng.random(len(base)) < 0.5].copy()
        up[:, 1] = 13
        idx = np.concatenate([base, up])[:m]
        assert side + 16 <= 1504
    else:                        # the W150k recipe: isolated cells, ~2 neighbours
        from amdq3d.config import AttrDict, cfg_from_yaml_file
        from amdq3d.synthetic import make_frame
        from oracle import voxel as V
        cfg = cfg_from_yaml_file("cfgs/waymo_models/centerpoint.yaml", AttrDict())
        pr, vs = cfg.DATA_CONFIG.POINT_CLOUD_RANGE, [0.1, 0.1, 0.15]
        pts = make_frame(seed, m * 5, m, pr, vs, [1504, 1504, 40])
        _, c, _ = V.voxelize_mean(pts, pr, vs, 5, 10 ** 9)
        idx = c.astype(np.int64)
    idx = idx[rng.permutation(len(idx))]
    if order == "morton":
        idx = idx[np.argsort(_morton_key(idx), kind="stable")]
    elif order == "linear":
        idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1]))]
    return idx.astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[16, 64, 128])
    ap.add_argument("--voxels", type=int, nargs="+", default=[30000, 120000, 480000, 1000000])
    ap.add_argument("--occupancy", default="sheet")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--order", default="random", choices=["random", "morton", "linear", "device"], help="row order of the voxels")
    ap.add_argument("--w16", action="store_true", help="16-bit fixed-point weights (two int8 planes) instead of 24-bit")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on the smallest size")
    args = ap.parse_args()
    from amdq3d import ops
    from oracle import sparse as OS
    dev = torch.device("cuda:0")
    shape = [41, 1504, 1504]
    # bring the GPU out of its idle clocks once (the first timing of a fresh process otherwise runs on the ramp: the 1 922 us
    # outlier of the round-1 sweep, and a 30 us "16-channel" line in round 2's first sweeps)
    junk = torch.empty((64 << 20,), dtype=torch.float32, device=dev)
    t_end = time.time() + 0.7
    while time.time() < t_end:
        junk.mul_(1.0001)
        torch.cuda.synchronize()
    del junk
    for m_req in args.voxels:
        idx = occupancy(args.occupancy, m_req, order=args.order)
        m = len(idx)
        cap = m
        count = torch.tensor([m], dtype=torch.int32, device=dev)
        if args.order == "device":      # the index builder's own row order (neighbourhood classes, heaviest first)
            lv = ops.IndexLevel(shape, 1, m, dev)
            ops.level_from_coords(torch.from_numpy(idx).to(dev), lv)
            nbr = ops.subm_neighbors(lv)
        else:                           # rows in the given order: the gather table comes from the oracle
            nbr = torch.from_numpy(OS.subm_neighbors(idx, shape)).to(dev)
        pairs = int((nbr >= 0).sum().item())
        for C in args.channels:
            g = torch.Generator().manual_seed(C)
            x = torch.relu(torch.randn(m, C, generator=g)).to(dev)
            w = torch.randn(C, 3, 3, 3, C, generator=g) / np.sqrt(13 * C)
            wq, codes, wdiv = OS.qconv_weight(w.numpy(), 8)
            wk = torch.from_numpy(np.ascontiguousarray(codes.reshape(C, 27, C).transpose(1, 0, 2))).to(dev)
            wk8 = wk.to(torch.int8)
            wdq = torch.from_numpy((np.float32(1) / wdiv).astype(np.float32)).to(dev)
            ps, pt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            amax = ops.absmax_cols(x, count)
            out = torch.empty((cap, C), device=dev)
            am_out = torch.zeros(C, device=dev)
            bufs = (torch.zeros((cap, C), dtype=torch.int8, device=dev), torch.empty((27, C, 3, C), dtype=torch.int8, device=dev),
                    torch.empty((C,), device=dev))

            def prep():
                ops.spconv_prep(x, amax, count, wk8, wdq, out=bufs)

            def conv():
                return ops.spconv_coded(bufs[0], bufs[1], bufs[2], nbr, count, post_scale=ps, post_shift=pt, relu=True, absmax_out=am_out, out=out, w16=args.w16)

            def timed(fn):
                # warm up for >= 30 ms: after the host-side set-up of a configuration (seconds of numpy) the GPU sits at idle clocks
                # and the first launches run several times slower — the 1 922 us outlier of the round-1 sweep was such a reading
                t0 = time.time()
                while time.time() - t0 < 0.03:
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / args.iters

            us_prep, us_conv = timed(prep), timed(conv)
            us = timed(lambda: (prep(), conv()))
            b_prep = m * C * 5 + 27 * C * C * 5
            b_conv = m * C * 1 + m * C * 4 + 27 * C * C + m * 27 * 4
            byts = b_prep + b_conv
            flops = 2.0 * pairs * C * C
            rec = {"layer": "SubMConv3d 3x3x3 %d->%d int8-w x per-channel-quantised a" % (C, C), "occupancy": args.occupancy, "order": args.order,
                   "voxels": m, "pairs": pairs, "us_per_layer": us, "us_prep": us_prep, "us_conv": us_conv,
                   "algorithmic_bytes": byts, "algorithmic_GBps": byts / us / 1e3, "frac_hbm": byts / us / 1e3 / HBM_GBS,
                   "conv_only_GBps": b_conv / us_conv / 1e3, "conv_only_frac_hbm": b_conv / us_conv / 1e3 / HBM_GBS,
                   # USEFUL ops (2 x pairs x C x C) over the int8 matrix peak — the pipe the layer runs on since round 3; the planes of the
                   # fixed-point weights (3, or 2 with --w16) are issued work, listed beside it, never as throughput
                   "useful_TOPs": flops / us / 1e6, "conv_only_useful_TOPs": flops / us_conv / 1e6,
                   "frac_mfma_i8_useful": flops / us_conv / 1e6 / I8_TOPS, "weight_planes": 2 if args.w16 else 3,
                   "issued_TOPs_lower_bound": (2 if args.w16 else 3) * flops / us_conv / 1e6, "voxels_per_s": m / us * 1e6}
            if args.cpu and m_req == args.voxels[0]:
                xa, _ = OS.qconv_act(x.cpu().numpy(), 8, cw=True)
                nb = nbr.cpu().numpy()
                t0 = time.time()
                OS.gather_conv(xa, wq, nb)
                rec["cpu_oracle_us"] = (time.time() - t0) * 1e6
                rec["cpu_cores"] = int(torch.get_num_threads())
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
