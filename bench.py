#!/usr/bin/env python
"""Headline benchmark: frames/s of int8 CenterPoint (SmoothQuant recipe of quant/quant_centerpoint.py) on synthetic
Waymo-shape frames (150 000 points / 30 000 voxels), one frame per step per GPU, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" + "cpu_baseline").
A step = one pass of the whole hot path (voxelize → sparse backbone → BEV backbone → head → decode → NMS) over one
frame, including the single device→host read of the detection counts that ends Detector.forward.
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import amdq3d  # noqa: E402,F401  (first: sets the HIP runtime's graph flag before anything initialises CUDA)
import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POINTS, N_VOXELS, N_FRAMES = 150000, 30000, 8
WORKLOADS = {      # --workload → (cfg, points, voxels, description); the headline (BASELINE metric) is "centerpoint"
    "centerpoint": ("cfgs/waymo_models/centerpoint.yaml", 150000, 30000,
                    "Full CenterPoint VoxelNet sparse backbone + dense int8 BEV head, synthetic Waymo-shape 150k-pt / 30k-voxel "
                    "frames (BASELINE configs[2]), SmoothQuant w8a8 dynamic, 1 frame/step/GPU"),
    "voxelnext": ("cfgs/nuscenes_models/cbgs_voxel0075_voxelnext.yaml", 40000, 20000,
                  "VoxelNeXt fully sparse int8 (quant/quant_voxelnext.py recipe: w8a8 per-channel 3-D convs + weight-only 2-D convs), "
                  "synthetic nuScenes-shape 40k-pt / 20k-voxel frames (BASELINE configs[4]), 1 frame/step/GPU"),
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TOPS = {"i8": 5000.0, "f16": 2500.0, "f32": 157.3}   # dense peaks: int8 = 2x f16/bf16 (2.5 PF); f32 MFMA = vector rate


def build(cfg_path, dev, n_points=N_POINTS, n_voxels=N_VOXELS):
    from amdq3d import quant
    from amdq3d.config import AttrDict, cfg_from_yaml_file
    from amdq3d.datasets import SyntheticDataset
    from amdq3d.detector import build_network
    from amdq3d.synthetic import init_synthetic_weights
    cfg = cfg_from_yaml_file(cfg_path, AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=N_FRAMES, n_points=n_points, n_voxels=n_voxels, seed0=4)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    init_synthetic_weights(model, 4)
    model.eval()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if cfg.MODEL.NAME == "VoxelNeXt":
        quant.dynamic_quant_voxelnext(model, 8, 8, sq=True)
    else:
        quant.quant(model, 8, 8, sq=True, alpha=0.5, static=False)
    model.voxel_capacity = int(os.environ.get("AQ3D_VOXEL_CAP", "65536"))
    model.to(dev)
    return cfg, ds, model, state


def kernel_work(eng, cat):
    """Algorithmic work of ONE frame for a timed category: (bytes, ops, n_launches). Per-unit figures: DESIGN.md §Kernels."""
    if cat == "conv2d_i8":
        ops = 0
        byts = 0
        n = 0
        for r in eng.dense_convs:
            if r["kind"] != "i8":
                continue
            g = r["geom"]
            L, K = g.Ho * g.Wo, g.Cin * g.ksize * g.ksize
            ops += 2 * L * K * g.Cout
            byts += g.H * g.W * g.Cin * 4 + L * g.Cout * 4 + K * g.Cout
            n += 1
        return byts, ops, n
    if cat == "conv2d_f32":
        ops = byts = n = 0
        for r in eng.dense_convs:
            if r["kind"] != "f32":
                continue
            g = r["geom"]
            rows = g.Cout * (g.deconv * g.deconv if g.deconv else 1)
            L, K = g.Ho * g.Wo, g.Cin * g.ksize * g.ksize
            ops += 2 * L * K * rows
            byts += g.H * g.W * g.Cin * 4 + L * rows * 4 + K * rows * 4
            n += 1
        return byts, ops, n
    if cat.startswith("spconv_c"):
        cout = int(cat[len("spconv_c"):])
        ops = byts = n = 0
        for r in eng.sparse_convs:
            if r["cout"] != cout:
                continue
            lv_out, lv_in = eng.levels[r["level"]], eng.levels[r["x"]["level"]]
            m_out, m_in = int(lv_out["count"].item()), int(lv_in["count"].item())
            nbr = eng._nbr_of(r["nbr"], r["level"])[:m_out]
            pairs = int((nbr >= 0).sum().item())
            # algorithmic activation bytes: the int8 code of the quantised input (1 B) or fp32 (4 B) — what the layer needs,
            # not what the chosen staging format (fp16 hi/lo pair) moves
            abytes = 1 if r["quant"] is not None and r["quant"]["on"] and r["quant"]["use_codes"] else 4
            wbytes = 1 if r["w_dequant"] is not None else 4
            ops += 2 * pairs * r["cin"] * r["cout"]
            byts += m_in * r["cin"] * abytes + m_out * r["cout"] * 4 + r["kvol"] * r["cin"] * r["cout"] * wbytes \
                + m_out * r["kvol"] * 4 + (m_out * r["cout"] * 4 if r["residual"] is not None else 0)
            n += 1
        return byts, ops, n
    return 0, 0, 0


def cpu_baseline(cfg, ds, state):
    """The CPU oracle (a port, not the reference binary) on ONE frame of the same workload, host cores of this box."""
    pts = ds[0]["points"]
    t0 = time.time()
    if cfg.MODEL.NAME == "VoxelNeXt":
        from oracle import voxelnext as OV
        b, s, l = OV.run(state, cfg, pts, OV.Recipe(8, 8, True))
        name = "oracle.voxelnext.run"
    else:
        from oracle import centerpoint as OC
        b, s, l = OC.run(state, cfg, pts, OC.Recipe(8, 8, True, 0.5))
        name = "oracle.centerpoint.run"
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "1 frame (%d pts) through %s, %.1f s, %d boxes" % (len(pts), name, dt, len(s))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="centerpoint", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=0, help="frames enqueued before the oldest one is read back "
                    "(default: one per device plan of the detector — 4 for CenterPoint, 2 for VoxelNeXt)")
    ap.add_argument("--breakdown", action="store_true", help="time every kernel family in an extra pass (stderr)")
    args = ap.parse_args()

    from amdq3d import dist as D
    from amdq3d import lib
    from amdq3d.datasets import load_data_to_gpu
    world, rank, local = D.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D.init("nccl")
    lib.load()
    cfg_path, n_points, n_voxels, wl_desc = WORKLOADS[args.workload]
    cfg, ds, model, state = build(cfg_path, dev, n_points, n_voxels)

    # frames resident in HBM before the timed region; frame i → rank i mod world
    mine = D.frames_for_rank(max(N_FRAMES, world), rank, world) if world <= N_FRAMES else [rank % N_FRAMES]
    batches = []
    for i in mine:
        b = ds.collate_batch([ds[i % N_FRAMES]])
        load_data_to_gpu(b, dev)
        batches.append(b)

    def step(i):
        with torch.no_grad():
            return model(dict(batches[i % len(batches)]))

    eng = model.engine()
    # warm-up: first pass times every kernel family to find the dominant one
    eng.start_timing()
    for i in range(max(args.warmup, 1)):
        step(i)
    warm = eng.stop_timing()
    n_warm = max(args.warmup, 1)
    dominant = max(warm.items(), key=lambda kv: kv[1][1])[0] if warm else None

    if args.inflight <= 0:
        args.inflight = 4 if getattr(eng, "split_index", False) else 2
    # graph warm-up (untimed): the first frames without per-kernel events capture the per-slot hipGraphs the timed region replays
    with torch.no_grad():
        for i in range(4 * eng.NSLOT * max(1, args.inflight)):
            model.forward_async(dict(batches[i % len(batches)])).counts()
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # two frames in flight: frame i+1 is enqueued before the detections of frame i are read back (every frame is still
    # completed and read inside this region)
    n_det = 0
    pending = collections.deque()
    for i in range(args.steps):
        with torch.no_grad():
            pending.append(model.forward_async(dict(batches[i % len(batches)])))
        if len(pending) >= args.inflight:
            n_det += int(pending.popleft().counts().sum())
    while pending:
        n_det += int(pending.popleft().counts().sum())
    torch.cuda.synchronize(dev)
    D.barrier()
    elapsed = time.perf_counter() - t0
    # roofline leg: the same frames once more with HIP events around every launch of the dominant kernel family (events cannot
    # bracket single kernels inside a replayed graph, so this pass enqueues eagerly; same kernels, same inputs, same stream)
    n_roof = max(1, min(args.steps, 50))
    eng.start_timing([dominant] if dominant else None)
    for i in range(n_roof):
        step(i)
    timed = eng.stop_timing()
    total_frames, tmax, fps = D.aggregate_throughput(args.steps, elapsed, dev)

    if rank == 0:
        roof = None
        if dominant and dominant in timed:
            groups, ms = timed[dominant]
            byts, ops, n_launch = kernel_work(eng, dominant)
            per_frame_ms = ms / n_roof
            avg_launch_s = per_frame_ms * 1e-3 / max(n_launch, 1)
            if dominant == "conv2d_i8":
                ach = ops / n_launch / avg_launch_s / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TOPS["i8"], "unit": "TFLOP/s",
                        "frac": ach / MFMA_PEAK_TOPS["i8"], "traffic": None}
            elif dominant == "conv2d_f32":
                ach = ops / n_launch / avg_launch_s / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TOPS["f32"], "unit": "TFLOP/s",
                        "frac": ach / MFMA_PEAK_TOPS["f32"], "traffic": None}
            else:
                # sparse conv family: bound by whichever roof its arithmetic intensity puts it under
                gbs = byts / max(n_launch, 1) / avg_launch_s / 1e9 if n_launch else 0.0
                tfs = ops / max(n_launch, 1) / avg_launch_s / 1e12 if n_launch else 0.0
                ridge = MFMA_PEAK_TOPS["f16"] * 1e12 / (HBM_PEAK_GBS * 1e9)          # FLOP per byte where the roofs meet
                if byts > 0 and ops / byts > ridge:
                    roof = {"bound": "mfma", "achieved": tfs, "peak": MFMA_PEAK_TOPS["f16"], "unit": "TFLOP/s",
                            "frac": tfs / MFMA_PEAK_TOPS["f16"], "traffic": None, "hbm_GBps_algorithmic": gbs}
                else:
                    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": None, "useful_TFLOPs": tfs}
                roof["intensity_flop_per_byte"] = ops / byts if byts else None
            try:      # HBM bytes per launch measured with rocprofv3 --pmc in a separate run (profiles/pmc_traffic.json)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    roof["traffic"] = json.load(f).get(dominant, None)
            except Exception:
                pass
            roof.update({"kernel": dominant, "launches_per_frame": n_launch, "avg_launch_us": avg_launch_s * 1e6,
                         "ms_per_frame": per_frame_ms})
        out = {
            "metric": "frames/sec/GPU int8 CenterPoint @150k pts; mAP delta vs fp32 ref" if args.workload == "centerpoint"
            else "frames/sec/GPU int8 VoxelNeXt @40k pts",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": wl_desc, "frames": N_FRAMES, "points": n_points, "voxels": n_voxels, "parallelism": "replicas(frames)x%d" % world,
                       "avg_boxes_per_frame": n_det / max(args.steps, 1)},
            "roofline": roof,
            "execution": "%s, %d frames in flight" % ("hipGraph replay" if getattr(eng, "_graphs", None) else "eager launches", args.inflight),
            "warmup_breakdown_ms_per_frame": {k: v[1] / n_warm for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, ds, state)
            except Exception as e:       # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
