"""A/B builds of the product library: csrc/<file>.hip recompiled with extra -D switches, linked with the product's other objects into
scratch/diag/libaq3d_ab_<tag>.so. scratch/ab_run.sh copies each over libaq3d.so in turn on the GPU box and runs the bench.

    python scratch/ab_build.py conv2d.hip base= nt=-DAQ3D_EXP=4 old=-DAQ3D_EXP=8
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdq3d  # noqa: E402,F401
from amdq3d import build as B  # noqa: E402


def main():
    fname, variants = sys.argv[1], sys.argv[2:]
    B.build(verbose=False)
    out_dir = os.path.join(ROOT, "scratch", "diag", "build_ko")
    os.makedirs(out_dir, exist_ok=True)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B._sources() if not s.endswith(fname)]
    src = os.path.join(B.CSRC, fname)
    for v in variants:
        tag, _, defs = v.partition("=")
        obj = os.path.join(out_dir, "ab_%s.o" % tag)
        out = os.path.join(ROOT, "scratch", "diag", "libaq3d_ab_%s.so" % tag)
        r = subprocess.run([B.HIPCC] + B.CFLAGS + [d for d in defs.split(",") if d] + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        r = subprocess.run([B.HIPCC, "-shared", "-fPIC", "--offload-arch=%s" % B.ARCH, "-o", out, obj] + others, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        print(out)


if __name__ == "__main__":
    main()
