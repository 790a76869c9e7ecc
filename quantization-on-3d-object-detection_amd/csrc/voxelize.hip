// Hard voxelization + mean VFE on device.
//
// Semantics restated from the reference's CPU voxelizer call contract
// (pcdet/datasets/processor/data_processor.py:45-61 → spconv Point2VoxelCPU3d.point_to_voxel) followed by
// MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29):
//   cell_j = floor((p_j - range_min_j) / voxel_size_j) in IEEE fp32, point dropped unless 0 <= cell_j < grid_j;
//   voxels are numbered in first-seen point order, at most max_voxels of them;
//   each voxel keeps its first P points (in point order); feature = sequential fp32 sum of those / count.
//
// Device algorithm (no sort, deterministic output):
//   1. insert   : one thread per point → 64-bit cell key into an open-addressing hash; atomicMin of the point
//                 index per slot (= first-seen point) and a lock-free push onto the slot's point list.
//   2. scan     : flag[i] = (point i is the first of its cell) → exclusive scan = voxel id in first-seen order.
//   3. finalize : one thread per voxel walks its list, keeps the P smallest point indices with a static
//                 compare-swap ladder (registers only) and sums their features in index order.
#include "common.h"
#include "scan.h"

#define VOX_BLOCK 256
#define VOX_PMAX 16

struct VoxWorkspace {
    int64_t *keys;   // [T]  cell key per slot, -1 empty
    int *head;       // [T]  head of the slot's point list, -1 empty
    int *first;      // [T]  smallest point index of the slot (0x7f7f7f7f = none)
    int *cnt;        // [T]  points in the slot
    int *pslot;      // [N]  slot of each point, -1 dropped
    int *next;       // [N]  next point in the slot's list
    int *flags;      // [N]
    int *prefix;     // [N]
    int *chunk_sums; // [chunks]
    int *total;      // [1]
    int *n_dev;      // [1]  n_points as a device scalar (scan length)
    int table_size;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int vox_table_size(int n_points) {
    int t = 1024;
    while (t < 2 * n_points) t <<= 1;
    return t;
}

extern "C" size_t aq3d_voxelize_workspace_bytes(int32_t n_points) {
    if (n_points < 1) n_points = 1;
    size_t T = (size_t)vox_table_size(n_points), N = (size_t)n_points;
    size_t b = 0;
    b += align_up(T * 8, 256);      // keys
    b += align_up(T * 4, 256) * 3;  // head, first, cnt
    b += align_up(N * 4, 256) * 4;  // pslot, next, flags, prefix
    b += align_up((size_t)aq3d_scan_chunks_for(n_points) * 4 + 4, 256);
    b += 256;                       // total + n_dev
    return b;
}

static bool vox_carve(void *ws, size_t ws_bytes, int n_points, VoxWorkspace *w) {
    if (ws_bytes < aq3d_voxelize_workspace_bytes(n_points)) return false;
    char *p = (char *)ws;
    size_t T = (size_t)vox_table_size(n_points), N = (size_t)n_points;
    w->table_size = (int)T;
    w->keys = (int64_t *)p;  p += align_up(T * 8, 256);
    w->head = (int *)p;      p += align_up(T * 4, 256);
    w->first = (int *)p;     p += align_up(T * 4, 256);
    w->cnt = (int *)p;       p += align_up(T * 4, 256);
    w->pslot = (int *)p;     p += align_up(N * 4, 256);
    w->next = (int *)p;      p += align_up(N * 4, 256);
    w->flags = (int *)p;     p += align_up(N * 4, 256);
    w->prefix = (int *)p;    p += align_up(N * 4, 256);
    w->chunk_sums = (int *)p; p += align_up((size_t)aq3d_scan_chunks_for(n_points) * 4 + 4, 256);
    w->total = (int *)p;
    w->n_dev = (int *)p + 1;
    return true;
}

struct VoxParams {
    float min_x, min_y, min_z, vs_x, vs_y, vs_z;
    int nx, ny, nz;
    int C, stride, col0, P, max_voxels, batch_index;
};

__global__ __launch_bounds__(VOX_BLOCK) void vox_insert(const float *__restrict__ points, int n, VoxParams q,
                                                        VoxWorkspace w) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *w.n_dev = n;
    for (int i = blockIdx.x * VOX_BLOCK + threadIdx.x; i < n; i += gridDim.x * VOX_BLOCK) {
        const float *p = points + (size_t)i * q.stride + q.col0;
        // IEEE fp32 subtract, divide, floor: the integer cell must be bit-exact.
        float fx = floorf(__fdiv_rn(__fsub_rn(p[0], q.min_x), q.vs_x));
        float fy = floorf(__fdiv_rn(__fsub_rn(p[1], q.min_y), q.vs_y));
        float fz = floorf(__fdiv_rn(__fsub_rn(p[2], q.min_z), q.vs_z));
        bool ok = (fx >= 0.f) && (fx < (float)q.nx) && (fy >= 0.f) && (fy < (float)q.ny) && (fz >= 0.f) &&
                  (fz < (float)q.nz);
        int slot = -1;
        if (ok) {
            int cx = (int)fx, cy = (int)fy, cz = (int)fz;
            int64_t key = ((int64_t)cz * q.ny + cy) * q.nx + cx;
            slot = aq3d_hash_insert(w.keys, w.table_size, key);
            if (slot >= 0) {
                atomicMin(&w.first[slot], i);
                atomicAdd(&w.cnt[slot], 1);
                int old = atomicExch(&w.head[slot], i);
                w.next[i] = old;
            }
        }
        w.pslot[i] = slot;
    }
}

__global__ __launch_bounds__(VOX_BLOCK) void vox_flags(int n, VoxWorkspace w) {
    for (int i = blockIdx.x * VOX_BLOCK + threadIdx.x; i < n; i += gridDim.x * VOX_BLOCK) {
        int s = w.pslot[i];
        w.flags[i] = (s >= 0 && w.first[s] == i) ? 1 : 0;
    }
}

// one thread per point; the first point of each cell finalises that voxel.
__global__ __launch_bounds__(VOX_BLOCK) void vox_finalize(const float *__restrict__ points, int n, VoxParams q,
                                                          VoxWorkspace w, int *__restrict__ coords_out,
                                                          float *__restrict__ feats_out, int *__restrict__ npts_out,
                                                          int *count_io, int cap, int base_is_count) {
    const int base = base_is_count ? *count_io : 0;
    for (int i = blockIdx.x * VOX_BLOCK + threadIdx.x; i < n; i += gridDim.x * VOX_BLOCK) {
        if (!w.flags[i]) continue;
        const int vid = w.prefix[i];
        if (vid >= q.max_voxels) continue;
        const int row = base + vid;
        if (row >= cap) continue;
        const int s = w.pslot[i];
        // P smallest point indices of the list, ascending, by a static compare-swap ladder.
        int idx[VOX_PMAX];
#pragma unroll
        for (int j = 0; j < VOX_PMAX; ++j) idx[j] = 0x7fffffff;
        for (int p = w.head[s]; p >= 0; p = w.next[p]) {
            int cand = p;
#pragma unroll
            for (int j = 0; j < VOX_PMAX; ++j) {
                int lo = min(idx[j], cand), hi = max(idx[j], cand);
                idx[j] = lo;
                cand = hi;
            }
        }
        int total = w.cnt[s];
        int kept = total < q.P ? total : q.P;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < VOX_PMAX; ++j) {
            if (j < kept) {
                const float *p = points + (size_t)idx[j] * q.stride + q.col0;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < q.C) acc[c] = __fadd_rn(acc[c], p[c]);
            }
        }
        const float denom = (float)(kept < 1 ? 1 : kept);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < q.C) feats_out[(size_t)row * q.C + c] = __fdiv_rn(acc[c], denom);
        int64_t key = w.keys[s];
        int cx = (int)(key % q.nx);
        int cy = (int)((key / q.nx) % q.ny);
        int cz = (int)(key / ((int64_t)q.nx * q.ny));
        coords_out[(size_t)row * 4 + 0] = q.batch_index;
        coords_out[(size_t)row * 4 + 1] = cz;
        coords_out[(size_t)row * 4 + 2] = cy;
        coords_out[(size_t)row * 4 + 3] = cx;
        if (npts_out) npts_out[row] = kept;
    }
}

__global__ void vox_update_count(const int *total, int max_voxels, int cap, int *count_io, int base_is_count) {
    int base = base_is_count ? *count_io : 0;
    int m = *total;
    if (m > max_voxels) m = max_voxels;
    if (base + m > cap) m = cap - base;
    *count_io = base + m;
}

extern "C" int aq3d_voxelize(const float *points, int32_t n_points, const aq3d_voxel_cfg *cfg, int32_t batch_index,
                             int32_t append, int32_t *coords_out, float *feats_out, int32_t *npts_out,
                             int32_t *count_io, int32_t cap, void *workspace, size_t workspace_bytes,
                             void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(cfg && coords_out && feats_out && count_io, "aq3d_voxelize: null argument");
    AQ3D_REQUIRE(n_points >= 0 && cap > 0, "aq3d_voxelize: bad n_points=%d cap=%d", n_points, cap);
    AQ3D_REQUIRE(cfg->num_features >= 3 && cfg->num_features <= 8, "aq3d_voxelize: num_features=%d not in 3..8",
                 cfg->num_features);
    AQ3D_REQUIRE(cfg->max_points_per_voxel >= 1 && cfg->max_points_per_voxel <= VOX_PMAX,
                 "aq3d_voxelize: max_points_per_voxel=%d not in 1..%d", cfg->max_points_per_voxel, VOX_PMAX);
    AQ3D_REQUIRE(cfg->point_stride >= cfg->num_features + (cfg->batch_col ? 1 : 0), "aq3d_voxelize: point_stride too small");
    AQ3D_REQUIRE((int64_t)cfg->grid[0] * cfg->grid[1] * cfg->grid[2] > 0, "aq3d_voxelize: empty grid");
    if (n_points == 0) {
        if (!append) hipMemsetAsync(count_io, 0, sizeof(int), stream);
        return AQ3D_OK;
    }
    AQ3D_REQUIRE(points && workspace, "aq3d_voxelize: null points/workspace");
    VoxWorkspace w;
    if (!vox_carve(workspace, workspace_bytes, n_points, &w)) {
        aq3d_set_error("aq3d_voxelize: workspace %zu < %zu bytes", workspace_bytes,
                       aq3d_voxelize_workspace_bytes(n_points));
        return AQ3D_ERR_WORKSPACE;
    }
    VoxParams q;
    q.min_x = cfg->range_min[0]; q.min_y = cfg->range_min[1]; q.min_z = cfg->range_min[2];
    q.vs_x = cfg->voxel_size[0]; q.vs_y = cfg->voxel_size[1]; q.vs_z = cfg->voxel_size[2];
    q.nx = cfg->grid[0]; q.ny = cfg->grid[1]; q.nz = cfg->grid[2];
    q.C = cfg->num_features; q.stride = cfg->point_stride; q.col0 = cfg->batch_col ? 1 : 0;
    q.P = cfg->max_points_per_voxel; q.max_voxels = cfg->max_voxels; q.batch_index = batch_index;

    const size_t T = (size_t)w.table_size;
    // keys and head are contiguous (both "-1"), first = 0x7f7f7f7f, cnt = 0
    hipMemsetAsync(w.keys, 0xff, (char *)w.first - (char *)w.keys, stream);
    hipMemsetAsync(w.first, 0x7f, T * 4, stream);
    hipMemsetAsync(w.cnt, 0, T * 4, stream);
    const int grid = aq3d_grid(n_points, VOX_BLOCK);
    hipLaunchKernelGGL(vox_insert, dim3(grid), dim3(VOX_BLOCK), 0, stream, points, n_points, q, w);
    hipLaunchKernelGGL(vox_flags, dim3(grid), dim3(VOX_BLOCK), 0, stream, n_points, w);
    int rc = aq3d_scan_flags(w.flags, w.n_dev, 1, n_points, w.prefix, w.total, w.chunk_sums, stream);
    if (rc != AQ3D_OK) return rc;
    hipLaunchKernelGGL(vox_finalize, dim3(grid), dim3(VOX_BLOCK), 0, stream, points, n_points, q, w, coords_out,
                       feats_out, npts_out, count_io, cap, append ? 1 : 0);
    hipLaunchKernelGGL(vox_update_count, dim3(1), dim3(1), 0, stream, w.total, q.max_voxels, cap, count_io,
                       append ? 1 : 0);
    AQ3D_CHECK_LAUNCH("aq3d_voxelize");
    return AQ3D_OK;
}

// ---- MeanVFE for pre-voxelized input (pcdet batch_dict 'voxels' / 'voxel_num_points') ----
__global__ __launch_bounds__(VOX_BLOCK) void mean_vfe_kernel(const float *__restrict__ voxels,
                                                             const float *__restrict__ num_points, int m, int P, int C,
                                                             const int *__restrict__ row_of_voxel, float *__restrict__ out) {
    const int total = m * C;
    for (int t = blockIdx.x * VOX_BLOCK + threadIdx.x; t < total; t += gridDim.x * VOX_BLOCK) {
        int v = t / C, c = t - v * C;
        const int row = row_of_voxel ? row_of_voxel[v] : v;
        if (row < 0) continue;
        const float *src = voxels + (size_t)v * P * C + c;
        float acc = 0.f;
        for (int j = 0; j < P; ++j) acc = __fadd_rn(acc, src[(size_t)j * C]);
        float nrm = fmaxf(num_points[v], 1.0f);
        out[(size_t)row * C + c] = __fdiv_rn(acc, nrm);
    }
}

extern "C" int aq3d_mean_vfe(const float *voxels, const float *num_points, int32_t m, int32_t P, int32_t C,
                             const int32_t *row_of_voxel, float *feats_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(m >= 0 && P >= 1 && C >= 1, "aq3d_mean_vfe: bad sizes m=%d P=%d C=%d", m, P, C);
    if (m == 0) return AQ3D_OK;
    AQ3D_REQUIRE(voxels && num_points && feats_out, "aq3d_mean_vfe: null argument");
    hipLaunchKernelGGL(mean_vfe_kernel, dim3(aq3d_grid((int64_t)m * C, VOX_BLOCK)), dim3(VOX_BLOCK), 0, stream, voxels,
                       num_points, m, P, C, row_of_voxel, feats_out);
    AQ3D_CHECK_LAUNCH("aq3d_mean_vfe");
    return AQ3D_OK;
}
