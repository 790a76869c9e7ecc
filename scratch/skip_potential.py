"""How much MFMA work would 16-row-tile skipping save if voxel rows were Morton-ordered? (CPU, oracle index generation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amdq3d.config import AttrDict, cfg_from_yaml_file
from amdq3d.datasets import SyntheticDataset
from oracle import sparse as S, voxel as V
cfg = cfg_from_yaml_file("cfgs/waymo_models/centerpoint.yaml", AttrDict())
ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=1, n_points=150000, n_voxels=30000, seed0=4)
dc = cfg.DATA_CONFIG
pr = [float(v) for v in dc.POINT_CLOUD_RANGE]
vox = {c.NAME: c for c in dc.DATA_PROCESSOR}["transform_points_to_voxels"]
vs = [float(v) for v in vox.VOXEL_SIZE]
feats, coords, _ = V.voxelize_mean(ds[0]["points"], pr, vs, vox.MAX_POINTS_PER_VOXEL, vox.MAX_NUMBER_OF_VOXELS["test"])
grid = V.grid_size_of(pr, vs)
shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
def morton(idx):
    key = np.zeros(len(idx), np.int64)
    z, y, x = idx[:, 1].astype(np.int64), idx[:, 2].astype(np.int64), idx[:, 3].astype(np.int64)
    for b in range(11):
        key |= ((x >> b) & 1) << (3 * b); key |= ((y >> b) & 1) << (3 * b + 1); key |= ((z >> b) & 1) << (3 * b + 2)
    return key
idx = coords
rng = np.random.default_rng(0)
for lvl, pad in enumerate([None, (1, 1, 1), (1, 1, 1), (0, 1, 1)]):
    if pad is not None:
        idx, _, shape = S.conv_outputs(idx, shape, (3, 3, 3), (2, 2, 2), pad)
    for name, order in (("random", rng.permutation(len(idx))), ("morton", np.argsort(morton(idx), kind="stable"))):
        ii = idx[order]
        nbr = S.subm_neighbors(ii, shape)
        present = nbr >= 0
        m = len(ii)
        pad_rows = (-m) % 16
        pz = np.concatenate([present, np.zeros((pad_rows, 27), bool)])
        t16 = pz.reshape(-1, 16, 27).any(axis=1)
        t32 = pz[: (len(pz) // 32) * 32].reshape(-1, 32, 27).any(axis=1)
        print("level %d M=%6d %-6s pairs/row %.2f | live (16-row tile, k): %.2f  (32-row, k): %.2f  of 27 -> work kept %.0f%% / %.0f%%"
              % (lvl, m, name, present.sum() / m, t16.sum(1).mean(), t32.sum(1).mean(), 100 * t16.mean(), 100 * t32.mean()))
