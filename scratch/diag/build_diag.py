"""Builds scratch/diag/libaq3d_diag.so (diagnostic kernels; never loaded by the product)."""
import os, subprocess, shutil, sys
HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OUT = os.path.join(HERE, "libaq3d_diag.so")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off"]
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs = []
    for src, extra, tag in (("nms_diag.hip", NOPK, ""), ("op_probe.hip", [], ""), ("op_probe2.hip", ["-DPROBE_TAG=pk"], "_pk"),
                            ("op_probe2.hip", ["-DPROBE_TAG=nopk"] + NOPK, "_nopk")):
        obj = os.path.join(HERE, "build", src + tag + ".o")
        r = subprocess.run([HIPCC] + FLAGS + extra + ["-c", os.path.join(HERE, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        objs.append(obj)
    r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build())
