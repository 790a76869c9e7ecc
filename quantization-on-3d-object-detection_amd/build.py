"""Builds csrc/*.hip + api.cpp into the in-tree shared library libaq3d.so for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
REPO = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libaq3d.so")
OBJ_DIR = os.path.join(CSRC, "build")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
# No -ffast-math: voxel indices and integer codes must be bit-exact (IEEE divide, round-half-even).
# No packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 / v_pk_mov_b32 ...): on this stack (gfx950, ROCm 7.2) the
# forms the compiler generates for the rotated-overlap geometry are not a pure function of their inputs while kernels of other
# HIP streams share the CU — the cause of the NMS masks that differed between identical frames in round 1. Measured with
# scratch/diag/op_probe2.hip, the same source built twice: 100 k - 440 k mismatching lanes of 2.9e8 with packed ops, 0 without
# (8 formulations, profiles/r02_nms/rootcause_evidence.txt). They buy nothing here either (round 1: ~4x the issue cost of a plain op).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-ffp-contract=off",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"] + NO_PACKED_FP32


def _sources():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))
    return [os.path.join(CSRC, n) for n in names]


def _deps_digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(REPO, "include")):
        for n in sorted(os.listdir(root)):
            p = os.path.join(root, n)
            if os.path.isfile(p) and n.endswith((".h", ".hip", ".cpp")):
                h.update(n.encode())
                with open(p, "rb") as f:
                    h.update(f.read())
    h.update(" ".join(CFLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    cmd = [HIPCC] + CFLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)   # host pass of the device-only feature flag
    if err.strip():
        sys.stderr.write(err + "\n")
    return obj


def build(force=False, verbose=True):
    """Compile every kernel file and link libaq3d.so. Returns the library path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "digest.txt")
    digest = _deps_digest()
    if not force and os.path.isfile(LIB_PATH) and os.path.isfile(stamp) and open(stamp).read() == digest:
        return LIB_PATH
    srcs = _sources()
    if verbose:
        print("[aq3d] hipcc --offload-arch=%s: %s" % (ARCH, ", ".join(os.path.basename(s) for s in srcs)), flush=True)
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    tmp = LIB_PATH + ".tmp"
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, LIB_PATH)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
