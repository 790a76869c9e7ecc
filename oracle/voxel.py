"""Hard voxelization + MeanVFE, restated.

Follows the call contract of pcdet/datasets/processor/data_processor.py:45-61,133-180 (spconv
Point2VoxelCPU3d.point_to_voxel: cell = floor((p - range_min) / voxel_size) in fp32, drop out-of-grid points,
voxels numbered in first-seen order up to max_voxels, first P points kept per voxel, coords (z, y, x)) and
MeanVFE pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29 (sum over the P slots / clamp_min(num, 1)).
The in-repo semantic twin for the cell arithmetic is vfe/dynamic_mean_vfe.py:53-57.
"""
import numpy as np


def mask_points_by_range(points, limit_range):
    """pcdet/utils/common_utils.py:78-81 — x, y only, inclusive on both ends."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def grid_size_of(point_cloud_range, voxel_size):
    """data_processor.py:135-136 — float64 divide then np.round."""
    r = np.asarray(point_cloud_range, dtype=np.float64)
    g = (r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float64)
    return np.round(g).astype(np.int64)


def hard_voxelize(points, point_cloud_range, voxel_size, max_points_per_voxel, max_voxels):
    """points f32 [N, C] (xyz first) → voxels f32 [M,P,C], coords i32 [M,3] (z,y,x), num_points i32 [M]."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    N, C = pts.shape
    P = int(max_points_per_voxel)
    rmin = np.asarray(point_cloud_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = grid_size_of(point_cloud_range, voxel_size)
    cell = np.floor((pts[:, :3] - rmin[None, :]) / vs[None, :])            # fp32 sub, div, floor
    ok = np.all((cell >= 0) & (cell < grid[None, :].astype(np.float32)), axis=1)
    idx = np.nonzero(ok)[0]
    c = cell[idx].astype(np.int64)
    key = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                                # first-seen order
    rank_of_uniq = np.empty_like(order)
    rank_of_uniq[order] = np.arange(order.size)
    vid = rank_of_uniq[inv]                                                 # voxel id of each kept point
    keep_v = vid < max_voxels
    idx, vid = idx[keep_v], vid[keep_v]
    M = int(min(order.size, max_voxels))
    # slot of each point inside its voxel = number of earlier points of the same voxel
    s = np.argsort(vid, kind="stable")
    vs_sorted = vid[s]
    start = np.r_[0, np.nonzero(np.diff(vs_sorted))[0] + 1]
    counts = np.diff(np.r_[start, vs_sorted.size])
    slot_sorted = np.arange(vs_sorted.size) - np.repeat(start, counts)
    slot = np.empty_like(slot_sorted)
    slot[s] = slot_sorted
    sel = slot < P
    voxels = np.zeros((M, P, C), dtype=np.float32)
    voxels[vid[sel], slot[sel]] = pts[idx[sel]]
    num = np.minimum(np.bincount(vid, minlength=M), P).astype(np.int32)
    ku = uniq[order][:M]
    coords = np.stack([ku // (grid[0] * grid[1]), (ku // grid[0]) % grid[1], ku % grid[0]], axis=1).astype(np.int32)
    return voxels, coords, num


def mean_vfe(voxels, num_points):
    """Sequential fp32 sum over the P slots (torch CPU reduces a short middle dim in order), then divide."""
    acc = np.zeros((voxels.shape[0], voxels.shape[2]), dtype=np.float32)
    for p in range(voxels.shape[1]):
        acc = acc + voxels[:, p, :]
    nrm = np.maximum(num_points.astype(np.float32), np.float32(1.0))[:, None]
    return (acc / nrm).astype(np.float32)


def voxelize_mean(points, point_cloud_range, voxel_size, max_points_per_voxel, max_voxels, batch_index=0):
    """→ feats f32 [M,C], coords i32 [M,4] (b,z,y,x), num i32 [M] — what aq3d_voxelize produces."""
    voxels, coords, num = hard_voxelize(points, point_cloud_range, voxel_size, max_points_per_voxel, max_voxels)
    feats = mean_vfe(voxels, num)
    b = np.full((coords.shape[0], 1), batch_index, dtype=np.int32)
    return feats, np.concatenate([b, coords], axis=1), num
