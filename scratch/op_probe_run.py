"""Which instruction class is not reproducible next to other streams' kernels? Launches scratch/diag/op_probe.hip after every
NMS of pipelined VoxelNeXt frames (2 plans) and, for comparison, alone. Prints per-class mismatch counters.

    STEPS=3000 python scratch/op_probe_run.py
"""
import collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d import ops
from amdq3d.datasets import load_data_to_gpu

NAMES = ["fma chain", "packed fp32 (v_pk_*)", "IEEE division", "sqrtf", "sinf/cosf", "atan2f", "LDS swap chain",
         "divergent loop + division", "rotated_overlap (ill-conditioned pair)", "rotated_overlap (sane pair)",
         "seg_intersection x16 + corner tests (no LDS)", "box_corners"]
dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "voxelnext")
cfgp, npts, nvox, _ = bench.WORKLOADS[wl]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
if wl == "voxelnext":
    model.async_streams = 2
diag = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libaq3d_diag.so"))
P = ctypes.c_void_p
NAMES2 = ["rotated_overlap (production function)", "seg_intersection x16 + corner tests", "corner tests x8 only",
          "seg_intersection x16 only", "branch-free seg_intersection x16", "accept tests of seg_intersection only (no division)",
          "the divisions only (unconditional)", "ONE seg_intersection"]
BLOCKS = int(os.environ.get("BLOCKS", "256"))
PROBES = []
for sym, names in (("aq3d_diag_op_probe", NAMES), ("aq3d_diag_op_probe2_pk", NAMES2), ("aq3d_diag_op_probe2_nopk", NAMES2),
                   # only formulations 6 and 7 per kernel — 98 (pk) / 38 (nopk) VGPRs, no scratch in either build
                   ("aq3d_diag_op_probe2_pk_small", NAMES2), ("aq3d_diag_op_probe2_nopk_small", NAMES2)):
    fn = getattr(diag, sym)
    fn.argtypes = [P, ctypes.c_int32, ctypes.c_int32, P, P]
    fn.restype = ctypes.c_int32
    PROBES.append(dict(sym=sym, fn=fn, names=names, ref=torch.zeros((16 * BLOCKS * 64,), dtype=torch.float32, device=dev),
                       counters=torch.zeros((32,), dtype=torch.int32, device=dev)))


def probe(have_ref):
    for p in PROBES:
        assert p["fn"](ops._p(p["ref"]), BLOCKS, int(have_ref), ops._p(p["counters"]), ops._stream()) == 0


def show(title):
    torch.cuda.synchronize()
    for p in PROBES:
        c = p["counters"].tolist()
        print("%s — %s: %d launches, %d lanes per class" % (title, p["sym"], c[31], c[30]))
        for i, n in enumerate(p["names"]):
            print("   %-52s in-launch mismatches %8d   vs quiet reference %8d" % (n, c[2 * i], c[2 * i + 1]))
        p["counters"].zero_()
    sys.stdout.flush()
    torch.cuda.synchronize()


probe(0)
torch.cuda.synchronize()
for p in PROBES:
    p["counters"].zero_()
for _ in range(300):
    probe(1)
show("alone (300 launches, nothing else on the chip)")
orig = ops.nms_rotated


def patched(boxes, count, thresh, post_max, workspace=None, out=None):
    r = orig(boxes, count, thresh, post_max, workspace=workspace, out=out)
    probe(1)
    return r


ops.nms_rotated = patched
NF = 8
batches = []
for i in range(NF):
    b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev); batches.append(b)
with torch.no_grad():
    for i in range(16):
        model.forward_async(dict(batches[i % NF])).counts()
    torch.cuda.synchronize()
    for p in PROBES:
        p["counters"].zero_()
    engs = [model._engine] + model._extra_engines
    steps = int(os.environ.get("STEPS", "3000"))
    pend = collections.deque()
    for i in range(steps):
        pend.append(model.forward_async(dict(batches[i % NF])))
        if len(pend) >= len(engs):
            pend.popleft().counts()
    while pend:
        pend.popleft().counts()
show("pipelined %s, %d frames on %d plans (probe launched after every NMS, inside the replayed graphs)" % (wl, steps, len(engs)))
