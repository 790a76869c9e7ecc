"""One 128-channel submanifold layer on the real level-3 index of a W150k frame, 10 launches: the process rocprofv3 --pmc wraps."""
import sys
import numpy as np
sys.path.insert(0, ".")
import amdq3d
import torch
from amdq3d import ops
from amdq3d.synthetic import make_frame
from oracle import sparse as OS
from oracle import voxel as OV
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w16 = "--w16" in sys.argv
RANGE = [-75.2, -75.2, -2, 75.2, 75.2, 4]; VS = [0.1, 0.1, 0.15]
grid = OV.grid_size_of(RANGE, VS)
pts = make_frame(4, 150000, 30000, RANGE, VS, grid)
f, c, n = OV.voxelize_mean(pts, RANGE, VS, 5, 150000)
idx, sh = c.astype(np.int32), [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
nlev = {16: 0, 32: 1, 64: 2, 128: 3}[C]
for k, s, p in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))][:nlev]:
    idx, _, sh = OS.conv_outputs(idx, sh, k, s, p)
    sh = [int(v) for v in sh]
dev = torch.device("cuda:0")
M = len(idx)
lv = ops.IndexLevel(sh, 1, M, dev)
ops.level_from_coords(torch.from_numpy(np.ascontiguousarray(idx)).to(dev), lv)
nbr = ops.subm_neighbors(lv)
count = torch.tensor([M], dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(C)
x = torch.relu(torch.randn(M, C, generator=g)).to(dev)
w = torch.randn(C, 3, 3, 3, C, generator=g) / np.sqrt(13 * C)
wq, codes, wdiv = OS.qconv_weight(w.numpy(), 8)
wk8 = torch.from_numpy(np.ascontiguousarray(codes.reshape(C, 27, C).transpose(1, 0, 2))).to(dev).to(torch.int8)
wdq = torch.from_numpy((np.float32(1) / wdiv).astype(np.float32)).to(dev)
amax = ops.absmax_cols(x, count)
bufs = ops.spconv_prep(x, amax, count, wk8, wdq)
out = torch.empty((M, C), device=dev)
for _ in range(10):
    ops.spconv_coded(bufs[0], bufs[1], bufs[2], nbr, count, relu=True, out=out, w16=w16)
torch.cuda.synchronize()
print("rows", M, "pairs", int((nbr >= 0).sum().item()))
