"""Builds scratch/diag/libaq3d_diag.so (diagnostic kernels; never loaded by the product)."""
import os, subprocess, shutil, sys
HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OUT = os.path.join(HERE, "libaq3d_diag.so")


def build():
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off",
           os.path.join(HERE, "nms_diag.hip"), os.path.join(HERE, "op_probe.hip"), "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build())
