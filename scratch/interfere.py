#!/usr/bin/env python
"""Do the index kernels slow down OTHER kernels that share the chip? Stream A replays a graph of dense int8 convs (128 -> 128 at
188 x 188); stream B concurrently replays a graph of one index-building sequence. A's time per conv alone vs next to each B.
    python scratch/interfere.py"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import numpy as np, torch
from amdq3d import ops
from amdq3d.config import AttrDict, cfg_from_yaml_file
from amdq3d.synthetic import make_frame
dev = torch.device("cuda:0")
cfg = cfg_from_yaml_file(os.path.join(ROOT, "cfgs/waymo_models/centerpoint.yaml"), AttrDict())
pr, vs = [float(v) for v in cfg.DATA_CONFIG.POINT_CLOUD_RANGE], [0.1, 0.1, 0.15]
pts = torch.from_numpy(make_frame(4, 150000, 30000, pr, vs, [1504, 1504, 40])).to(dev)
vcfg = ops.make_voxel_cfg(pr, vs, [1504, 1504, 40], 5, 5, False, 5, 150000)
shape0 = [41, 1504, 1504]
lv0 = ops.IndexLevel(shape0, 1, 40960, dev, sorted_rows=False)
lv0.c.flags = 0
shape1 = ops.conv_out_shape(shape0, (3, 3, 3), (2, 2, 2), (1, 1, 1))
lv1 = ops.IndexLevel(shape1, 1, 102400, dev, sorted_rows=False)
lv1.c.flags = 0
ws = ops.index_workspace(102400, max(lv0.words, lv1.words), 163840, dev)
feats, npts = torch.zeros((40960, 5), device=dev), torch.zeros((40960,), dtype=torch.int32, device=dev)
nbr0 = torch.full((40960, 27), -1, dtype=torch.int32, device=dev)
jobs0 = ops.nbr_jobs([(lv0, (3, 3, 3), (1, 1, 1), (1, 1, 1), nbr0)])
big = torch.zeros((36 * 1024 * 1024 // 4,), device=dev)
probe = torch.zeros((lv0.words * 2,), dtype=torch.int32, device=dev)
ridx = torch.randint(0, lv0.words * 2, (150000,), device=dev)
ones = torch.ones((150000,), dtype=torch.int32, device=dev)
_f, _c, _ = __import__("oracle.voxel", fromlist=["x"]).voxelize_mean(pts.cpu().numpy(), pr, vs, 5, 150000)
coords30k = torch.from_numpy(np.concatenate([np.zeros((len(_c), 1), np.int32), _c.astype(np.int32)], 1)).to(dev).contiguous()
B_SEQS = {
    "level_from_points": lambda: ops.level_from_points(pts, vcfg, lv0, feats_out=feats, npts_out=npts, workspace=ws),
    "level_downsample": lambda: ops.level_downsample(lv0, (3, 3, 3), (2, 2, 2), (1, 1, 1), lv1, workspace=ws),
    "level_neighbors": lambda: ops.level_neighbors(lv0, jobs0),
    "memset_36MB": lambda: big.zero_(),
    # attribution probes: what in level_from_points disturbs its neighbours?
    "level_from_coords_30k": lambda: ops.level_from_coords(coords30k, lv0, workspace=ws),       # mark (30 k atomicOr) + count + rank + rows
    "stream_bitmap_11.6MB_read": lambda: torch.sum(lv0.bits, dtype=torch.int64),                 # one streaming read of the level-0 bitmap
    "zero_bitmap_11.6MB": lambda: lv0.bits.zero_(),
    "scatter_add_150k_atomics_into_11.6MB": lambda: probe.scatter_add_(0, ridx, ones),          # 150 k atomic adds on random words
}
# stream A: the conv
Cin = Cout = 128; H = W = 188
g_ = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(H, W, Cin, generator=g_)).to(dev)
wk = ops.conv_weight_to_k(torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)).to(dev)
geom = ops.conv_geom(H, W, Cin, Cout, 3, 1, 1, relu=True)
params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
y = torch.empty((H, W, Cout), device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
NA = 20
def capture(stream, fn, reps):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(reps):
            fn()
    return g
B_SEQS["level_from_points"](); torch.cuda.synchronize()
gA = capture(sA, lambda: ops.conv2d_i8(x, params, geom, out=y), NA)
def time_A(gB, n=30):
    stop = [False]
    def spin():
        with torch.cuda.stream(sB):
            while not stop[0]:
                gB.replay()
                sB.synchronize()
    th = None
    if gB is not None:
        th = threading.Thread(target=spin); th.start(); time.sleep(0.05)
    with torch.cuda.stream(sA):
        gA.replay(); sA.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            gA.replay()
        sA.synchronize()
        dt = time.perf_counter() - t0
    if th is not None:
        stop[0] = True; th.join()
    return dt / (n * NA) * 1e6
print(json.dumps({"A": "conv2d_i8 128->128 188x188", "alone_us": round(time_A(None), 2)}), flush=True)
for name, fn in B_SEQS.items():
    gB = capture(sB, fn, 1)
    # B alone
    with torch.cuda.stream(sB):
        gB.replay(); sB.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gB.replay()
        sB.synchronize()
        tb = (time.perf_counter() - t0) / 50 * 1e6
    print(json.dumps({"B": name, "B_alone_us": round(tb, 1), "A_next_to_B_us": round(time_A(gB), 2)}), flush=True)


# ---- the other direction: how long does each index sequence take while convs of ANOTHER stream keep the chip busy?
def time_B(gB, busy, n=30):
    stop = [False]
    def spin():
        with torch.cuda.stream(sA):
            while not stop[0]:
                gA.replay()
                sA.synchronize()
    th = None
    if busy:
        th = threading.Thread(target=spin); th.start(); time.sleep(0.05)
    with torch.cuda.stream(sB):
        gB.replay(); sB.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            gB.replay()
            sB.synchronize()
        dt = time.perf_counter() - t0
    if th is not None:
        stop[0] = True; th.join()
    return dt / n * 1e6
for name in ("level_from_points", "level_downsample", "level_neighbors", "level_from_coords_30k", "zero_bitmap_11.6MB"):
    gB = capture(sB, B_SEQS[name], 1)
    print(json.dumps({"B": name, "B_alone_us_incl_launch": round(time_B(gB, False), 1), "B_next_to_20_convs_in_flight_us": round(time_B(gB, True), 1)}), flush=True)
