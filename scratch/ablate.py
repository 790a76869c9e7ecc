#!/usr/bin/env python
"""Where does the frame rate go? The pipelined bench loop (4 frames in flight, hipGraph replay) with one kernel family at a time
turned into a no-op (results are garbage; only the timing is read): fps with the family removed -> its share of the CHIP time of a
frame in throughput mode, as opposed to its isolated kernel durations (rocprof), which count tails other frames fill.
    python scratch/ablate.py [family ...]"""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa
import numpy as np, torch
import bench
from amdq3d import ops
from amdq3d.datasets import load_data_to_gpu
FAMILIES = {
    "conv2d_i8": ["conv2d_i8", "conv2d_i8_group"],
    "conv2d_f32": ["conv2d_f32", "conv2d_f32_group"],
    "spconv_coded": ["spconv_coded"],
    "spconv_f32": ["spconv", "spconv_group"],
    "sp_prep": ["spconv_prep"],
    "sq_prep": ["sq_prep", "sq_prep_group", "piece_absmax"],
    "index_downsample": ["level_downsample"],
    "index_neighbors": ["level_neighbors"],
    "voxelize": ["level_from_points"],
    "decode_nms": ["decode_nms_batch"],
    "scatter_dense": ["scatter_dense"],
}
dev = torch.device("cuda:0")
steps = int(os.environ.get("STEPS", "200"))
want = sys.argv[1:] or ["none"] + list(FAMILIES)
orig = {n: getattr(ops, n) for names in FAMILIES.values() for n in names}
def measure(skip):
    for n, f in orig.items():
        setattr(ops, n, f)
    cfg_path, npts, nvox, _ = bench.WORKLOADS["centerpoint"]
    cfg, ds, model, state = bench.build(cfg_path, dev, npts, nvox, seed0=4)
    batches = []
    for i in range(8):
        b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
    def run(n):
        pend = collections.deque()
        for i in range(n):
            with torch.no_grad():
                pend.append(model.forward_async(dict(batches[i % 8])))
            if len(pend) >= 4:
                pend.popleft().counts()
        while pend:
            pend.popleft().counts()
    with torch.no_grad():
        model(dict(batches[0]))          # one complete frame first: index levels, tables and every buffer hold valid (then stale) data
    for n in FAMILIES.get(skip, []):
        setattr(ops, n, (lambda *a, **k: k.get("out")))
    if skip.startswith("dup:"):          # the family launched TWICE per frame: the cross-check of the removal figure
        for n in FAMILIES[skip[4:]]:
            f = orig[n]
            setattr(ops, n, (lambda f_: (lambda *a, **k: (f_(*a, **k), f_(*a, **k))[1]))(f))
    run(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)
base = None
for fam in want:
    try:
        fps = measure(fam)
    except Exception as e:
        print(json.dumps({"skipped": fam, "failed": repr(e)})); continue
    if fam == "none":
        base = fps
    rec = {"skipped": fam, "frames_per_s": round(fps, 1), "ms_per_frame": round(1e3 / fps, 4)}
    if base:
        rec["chip_ms_per_frame_of_family"] = round(1e3 / base - 1e3 / fps, 4)
    print(json.dumps(rec), flush=True)
