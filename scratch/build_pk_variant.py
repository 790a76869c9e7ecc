"""Builds scratch/build_pk/libaq3d_pk.so: the same sources WITH packed-fp32 code generation (the round-1 flags), for A/B runs
(AQ3D_LIB=scratch/build_pk/libaq3d_pk.so python scratch/numerics_check.py)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quantization-on-3d-object-detection_amd", "csrc")
OUT = os.path.join(ROOT, "scratch", "build_pk")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off"]
os.makedirs(OUT, exist_ok=True)
srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def cc(f):
    o = os.path.join(OUT, f + ".o")
    r = subprocess.run(["hipcc"] + FLAGS + (["-x", "hip"] if f.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, f), "-o", o], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return o


with ThreadPoolExecutor(6) as ex:
    objs = list(ex.map(cc, srcs))
r = subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(OUT, "libaq3d_pk.so")] + objs, capture_output=True, text=True)
assert r.returncode == 0, r.stderr
print(os.path.join(OUT, "libaq3d_pk.so"))
