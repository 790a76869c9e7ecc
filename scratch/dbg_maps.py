import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
def maps(tag):
    print(tag, sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "libaq3d" in l or "hsa-runtime" in l}), flush=True)
import torch
maps("after import torch")
from amdq3d import lib
lib.load()
maps("after lib.load")
print(torch.cuda.is_available())
maps("after is_available")
x = torch.zeros(4, device="cuda")
maps("after first alloc")
