import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
G.build()
print("after build:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "libaq3d" in l or "hsa-runtime" in l}))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
def peek(tag):
    e = hip.hipPeekAtLastError()
    print("%-28s last error = %d %s" % (tag, e, hip.hipGetErrorString(e).decode()), flush=True)
peek("after build")
import torch
from helpers import build_model, small_cfg
from amdq3d import lib, quant
from amdq3d.datasets import load_data_to_gpu
from oracle import centerpoint as OC
peek("imports")
lib.load(); peek("lib.load")
print(torch.cuda.is_available()); peek("is_available")
dev = torch.device("cuda:0")
cfg = small_cfg("waymo")
ds, model = build_model(cfg, n_points=9000, n_voxels=2500); peek("build_model")
state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
quant.quant(model, 8, 8, sq=True, alpha=0.5, static=False)
model.cuda(); peek("model.cuda")
batch = ds.collate_batch([ds[0]]); load_data_to_gpu(batch, dev); peek("load_data")
eng = model.engine(); peek("engine built")
with torch.no_grad():
    try:
        pred, _ = model(batch)
        print("forward ok", pred[0]["pred_scores"].shape)
    except Exception as e:
        print("FAILED", e)
peek("after forward")
print(sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "libaq3d" in l or "hsa-runtime" in l}))
