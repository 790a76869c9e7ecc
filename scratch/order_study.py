"""CPU study: fraction of (row-block, kernel offset) units that hold at least one neighbour, for different row orders and block
sizes, on the index levels of a W150k frame. Decides the row order the index builder emits and the skip granularity of the
gather-GEMM kernels."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import amdq3d
from amdq3d.synthetic import make_frame
from oracle import sparse as OS
from oracle import voxel as OV

RANGE = [-75.2, -75.2, -2, 75.2, 75.2, 4]
VS = [0.1, 0.1, 0.15]
grid = OV.grid_size_of(RANGE, VS)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pts = make_frame(seed, 150000, 30000, RANGE, VS, grid)
f, c, n = OV.voxelize_mean(pts, RANGE, VS, 5, 150000)
shape = [grid[2] + 1, grid[1], grid[0]]
levels = [(c.astype(np.int32), shape)]
geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
for k, s, p in geoms:
    idx, nbr, osh = OS.conv_outputs(levels[-1][0], levels[-1][1], k, s, p)
    levels.append((idx, osh))


def frac(mask, order, blk):
    m = mask[order]
    pad = (-len(m)) % blk
    if pad:
        m = np.concatenate([m, np.zeros((pad, m.shape[1]), bool)])
    live = m.reshape(-1, blk, m.shape[1]).any(axis=1)
    return live.sum() / (live.shape[0] * live.shape[1]), live.sum()


def patch_order(idx, shape, ps=8):
    b, z, y, x = (idx[:, i].astype(np.int64) for i in range(4))
    PY, PX = (shape[1] + ps - 1) // ps, (shape[2] + ps - 1) // ps
    key = ((((b * shape[0] + z) * PY + y // ps) * PX + x // ps) * ps + y % ps) * ps + x % ps
    return np.argsort(key, kind="stable")


def patch_order_zin(idx, shape, ps=8):
    """patch-major with z INSIDE the patch column: (b, py, px, z, y%8, x%8)"""
    b, z, y, x = (idx[:, i].astype(np.int64) for i in range(4))
    PY, PX = (shape[1] + ps - 1) // ps, (shape[2] + ps - 1) // ps
    key = ((((b * PY + y // ps) * PX + x // ps) * shape[0] + z) * ps + y % ps) * ps + x % ps
    return np.argsort(key, kind="stable")


def morton(idx):
    def spread(v):
        v = v.astype(np.uint64)
        r = np.zeros_like(v)
        for i in range(12):
            r |= ((v >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i)
        return r
    key = spread(idx[:, 3]) | (spread(idx[:, 2]) << np.uint64(1)) | (spread(idx[:, 1]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def tile_mask_sort(mask, order, tile=128):
    """within each tile of `tile` rows (of the given order), sort rows by their 27-bit mask value"""
    w = (mask[order].astype(np.uint64) << np.arange(mask.shape[1], dtype=np.uint64)[None, :]).sum(axis=1)
    out = order.copy()
    for t0 in range(0, len(order), tile):
        s = np.argsort(w[t0:t0 + tile], kind="stable")
        out[t0:t0 + tile] = order[t0:t0 + tile][s]
    return out


rng = np.random.default_rng(0)
for li, (idx, sh) in enumerate(levels):
    nbr = OS.subm_neighbors(idx, sh)
    mask = nbr >= 0
    M = len(idx)
    dens = mask.mean()
    print("level %d: rows %d shape %s, neighbours per row %.2f (%.1f%%)" % (li, M, sh, mask.sum(1).mean(), 100 * dens))
    orders = {"random": rng.permutation(M), "raster(key)": np.argsort(OS._key(idx, sh), kind="stable"),
              "patch8 z-outer": patch_order(idx, sh, 8), "patch8 z-inner": patch_order_zin(idx, sh, 8),
              "patch4 z-inner": patch_order_zin(idx, sh, 4), "morton": morton(idx)}
    gw = (mask.astype(np.uint64) << np.arange(27, dtype=np.uint64)[None, :]).sum(axis=1)
    orders["global mask sort"] = np.argsort(gw, kind="stable")
    for name, o in list(orders.items()):
        if name in ("patch8 z-inner", "morton", "raster(key)"):
            orders[name + " + tile128 mask sort"] = tile_mask_sort(mask, o, 128)
    for name, o in orders.items():
        f16, n16 = frac(mask, o, 16)
        f32_, _ = frac(mask, o, 32)
        f128, n128 = frac(mask, o, 128)
        print("   %-38s live units: blk16 %.3f (x%.2f of useful)  blk32 %.3f  blk128 %.3f" % (name, f16, f16 / dens, f32_, f128))

print("\n=== cheap class keys: stable counting sort by a few mask-derived bits on top of raster order ===")
for li, (idx, sh) in enumerate(levels):
    nbr = OS.subm_neighbors(idx, sh)
    mask = nbr >= 0
    M = len(idx)
    dens = mask.mean()
    base = np.argsort(OS._key(idx, sh), kind="stable")
    mb = mask[base]
    m3 = mb.reshape(M, 3, 3, 3)
    keys = {}
    zpl = m3.any(axis=(2, 3))                 # [M,3] any neighbour in plane kz
    keys["kz planes (3b)"] = (zpl * (1 << np.arange(3))).sum(1)
    ypl = m3.any(axis=(1, 3))
    xpl = m3.any(axis=(1, 2))
    keys["kz+ky planes (5b)"] = keys["kz planes (3b)"] * 4 + (ypl[:, [0, 2]] * (1 << np.arange(2))).sum(1)
    keys["kz,ky,kx planes (7b)"] = keys["kz+ky planes (5b)"] * 4 + (xpl[:, [0, 2]] * (1 << np.arange(2))).sum(1)
    zy = m3.any(axis=3).reshape(M, 9)         # any neighbour in row (kz,ky)
    keys["(kz,ky) rows (9b)"] = (zy * (1 << np.arange(9))).sum(1)
    # top-entropy bits
    p = mb.mean(0)
    ent = -(p * np.log2(np.maximum(p, 1e-9)) + (1 - p) * np.log2(np.maximum(1 - p, 1e-9)))
    for nb in (6, 8, 10, 12):
        sel = np.argsort(-ent)[:nb]
        keys["top-%d entropy bits" % nb] = (mb[:, sel] * (1 << np.arange(nb))).sum(1)
    keys["popcount (5b)"] = mb.sum(1)
    keys["full mask (27b)"] = (mb.astype(np.uint64) << np.arange(27, dtype=np.uint64)[None, :]).sum(1)
    print("level %d (density %.3f):" % (li, dens))
    for name, k in keys.items():
        o = base[np.argsort(k, kind="stable")]
        f16, _ = frac(mask, o, 16)
        o2 = tile_mask_sort(mask, o, 128)
        f16b, _ = frac(mask, o2, 16)
        f128, _ = frac(mask, o, 128)
        print("   %-26s classes %6d  blk16 %.3f (x%.2f)   +tile128 sort %.3f (x%.2f)  blk128 %.3f" % (name, len(np.unique(k)), f16, f16 / dens, f16b, f16b / dens, f128))
