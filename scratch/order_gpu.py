"""GPU experiment: one submanifold layer per index level of a W150k frame (real channel counts), rows in random / raster /
plane-class-sorted order. Answers: what does the row order buy the existing gather-GEMM kernels (128-row-tile unit skipping)?"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import amdq3d
import torch
from amdq3d import ops
from amdq3d.synthetic import make_frame
from oracle import sparse as OS
from oracle import voxel as OV

RANGE = [-75.2, -75.2, -2, 75.2, 75.2, 4]
VS = [0.1, 0.1, 0.15]
grid = OV.grid_size_of(RANGE, VS)
pts = make_frame(4, 150000, 30000, RANGE, VS, grid)
f, c, n = OV.voxelize_mean(pts, RANGE, VS, 5, 150000)
shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
levels = [(c.astype(np.int32), shape)]
for k, s, p in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]:
    idx, nbr, osh = OS.conv_outputs(levels[-1][0], levels[-1][1], k, s, p)
    levels.append((idx, [int(v) for v in osh]))
dev = torch.device("cuda:0")
fp16w = "--w16" in sys.argv


def class_key(idx, sh):
    nbr = OS.subm_neighbors(idx, sh)
    m3 = (nbr >= 0).reshape(len(idx), 3, 3, 3)
    z = m3.any(axis=(2, 3)); y = m3.any(axis=(1, 3)); x = m3.any(axis=(1, 2))
    return (z * (1 << np.arange(3))).sum(1) * 16 + (y[:, [0, 2]] * (1 << np.arange(2))).sum(1) * 4 + (x[:, [0, 2]] * (1 << np.arange(2))).sum(1)


def timed(fn, iters=50):
    t0 = time.time()
    while time.time() - t0 < 0.05:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


rng = np.random.default_rng(0)
for (idx, sh), C in zip(levels, (16, 32, 64, 128)):
    M = len(idx)
    raster = np.argsort(OS._key(idx, sh), kind="stable")
    ck = class_key(idx, sh)
    pc = (OS.subm_neighbors(idx, sh) >= 0).sum(1)
    orders = {"random": rng.permutation(M), "class7": raster[np.argsort(ck[raster], kind="stable")],
              "class7 heavy-first": raster[np.argsort(-ck[raster], kind="stable")],
              "class7 by class weight desc": None, "popcount desc": raster[np.argsort(-pc[raster], kind="stable")], "device": None}
    # classes ordered by their mean neighbour count, heaviest first
    cls = np.unique(ck)
    wgt = {c_: pc[ck == c_].mean() for c_ in cls}
    rank = {c_: i for i, c_ in enumerate(sorted(cls, key=lambda c_: -wgt[c_]))}
    rk = np.array([rank[v] for v in ck])
    orders["class7 by class weight desc"] = raster[np.argsort(rk[raster], kind="stable")]
    for name, o in orders.items():
        if name != "device" and o is None:
            continue
        count = torch.tensor([M], dtype=torch.int32, device=dev)
        if name == "device":       # the index builder's own row order
            lv = ops.IndexLevel(sh, 1, M, dev)
            ops.level_from_coords(torch.from_numpy(np.ascontiguousarray(idx)).to(dev), lv)
            nbr = ops.subm_neighbors(lv)
        else:
            ref = OS.subm_neighbors(idx[o], sh)
            nbr = torch.from_numpy(ref).to(dev)
        g = torch.Generator().manual_seed(C)
        x = torch.relu(torch.randn(M, C, generator=g)).to(dev)
        w = torch.randn(C, 3, 3, 3, C, generator=g) / np.sqrt(13 * C)
        wq, codes, wdiv = OS.qconv_weight(w.numpy(), 8)
        wk8 = torch.from_numpy(np.ascontiguousarray(codes.reshape(C, 27, C).transpose(1, 0, 2))).to(dev).to(torch.int8)
        wdq = torch.from_numpy((np.float32(1) / wdiv).astype(np.float32)).to(dev)
        ps, pt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        amax = ops.absmax_cols(x, count)
        out = torch.empty((M, C), device=dev)
        am_out = torch.zeros(C, device=dev)
        bufs = (torch.zeros((M, C), dtype=torch.int8, device=dev), torch.empty((27, C, 3, C), dtype=torch.int8, device=dev), torch.empty((C,), device=dev))
        ops.spconv_prep(x, amax, count, wk8, wdq, out=bufs)
        us = timed(lambda: ops.spconv_coded(bufs[0], bufs[1], bufs[2], nbr, count, post_scale=ps, post_shift=pt, relu=True, absmax_out=am_out, out=out, w16=fp16w))
        nb = nbr.cpu().numpy() >= 0
        pad = (-M) % 128
        live = np.concatenate([nb, np.zeros((pad, 27), bool)]).reshape(-1, 128, 27).any(1).mean()
        print(json.dumps({"level": levels.index((idx, sh)) if False else None, "C": C, "rows": M, "order": name, "us_conv": round(us, 2), "live_units_128": round(float(live), 3), "density": round(float(nb.mean()), 3), "fp16w": fp16w}), flush=True)
