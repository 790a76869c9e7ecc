import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import torch, bench
from amdq3d import ops
from amdq3d.datasets import load_data_to_gpu
dev = torch.device("cuda:0")
cfgp, npts, nvox, _ = bench.WORKLOADS["voxelnext"]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
b = ds.collate_batch([ds[6]]); load_data_to_gpu(b, dev)
with torch.no_grad():
    model(dict(b))
e = model.engine(); h = e.heads[4]
c = int(h["cand_count"][0].item()); cap = min(h["K"], h["pre"])
boxes = h["cand_boxes"][0][:cap].clone(); count = h["cand_count"][0].clone()
ws = torch.zeros_like(h["nms_ws"]); keep = torch.zeros(cap, dtype=torch.int32, device=dev); kc = torch.zeros(1, dtype=torch.int32, device=dev)
cb = (cap + 63) // 64
def maskcs():
    m = ws[: cap * cb * 8].view(torch.int64).view(cap, cb)[:c]
    return int((m >> 4).sum())
vals = []
for it in range(int(os.environ.get("ITERS", "20"))):
    ops.nms_rotated(boxes, count, h["thresh"], h["post"], workspace=ws, out=(keep, kc))
    torch.cuda.synchronize()
    vals.append(maskcs())
print("cands", c, "distinct mask checksums over runs:", len(set(vals)))
os.environ["AQ3D_NMS_DBG_DUMP"] = "1"
ops.nms_rotated(boxes, count, h["thresh"], h["post"], workspace=ws, out=(keep, kc))
