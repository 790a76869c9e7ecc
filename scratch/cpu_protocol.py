"""SURVEY §8d's CPU-baseline protocol for the oracle (a restatement, not the upstream binary): torch.set_num_threads(T) for
T in {all host cores, 1}, warm-up + timed frames of W150k, median ms/frame and frames/s, CPU model and core count.
    python scratch/cpu_protocol.py [--frames-all 10 --warm-all 3 --frames-one 3 --warm-one 1]  → one JSON line"""
import argparse, json, os, platform, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import numpy as np, torch
import bench
from amdq3d.config import AttrDict, cfg_from_yaml_file
from amdq3d.datasets import SyntheticDataset
from amdq3d.detector import build_network
from amdq3d.synthetic import init_synthetic_weights
from oracle import centerpoint as OC

ap = argparse.ArgumentParser()
ap.add_argument("--frames-all", type=int, default=10); ap.add_argument("--warm-all", type=int, default=3)
ap.add_argument("--frames-one", type=int, default=3); ap.add_argument("--warm-one", type=int, default=1)
args = ap.parse_args()
cfg = cfg_from_yaml_file(bench.WORKLOADS["centerpoint"][0], AttrDict())
ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=8, n_points=150000, n_voxels=30000, seed0=4)
model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
init_synthetic_weights(model, 4)
state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
cpu = "unknown"
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
all_threads = torch.get_num_threads()
out = {"what": "oracle.centerpoint.run (SmoothQuant w8a8 dynamic), W150k frames seeded 4..11; a CPU restatement, not the upstream binary",
       "cpu": cpu, "logical_cpus": os.cpu_count(), "torch_default_threads": all_threads, "runs": []}
for T, warm, frames in ((all_threads, args.warm_all, args.frames_all), (1, args.warm_one, args.frames_one)):
    torch.set_num_threads(T)
    os.environ["OMP_NUM_THREADS"] = str(T)
    times = []
    for i in range(warm + frames):
        t0 = time.time()
        OC.run(state, cfg, ds[i % 8]["points"], OC.Recipe(8, 8, True, 0.5))
        if i >= warm:
            times.append(time.time() - t0)
        print("T=%d frame %d: %.1f s" % (T, i, time.time() - t0), file=sys.stderr, flush=True)
    out["runs"].append({"threads": T, "warmup_frames": warm, "timed_frames": frames, "median_s_per_frame": float(np.median(times)),
                        "frames_per_s": 1.0 / float(np.median(times)), "all_s": [round(t, 2) for t in times]})
print(json.dumps(out), flush=True)
