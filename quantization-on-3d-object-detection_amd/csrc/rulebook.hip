// Sparse-conv index generation: coordinate hash, submanifold gather tables, strided-conv output sets.
//
// The reference gets all of this from spconv (SubMConv3d / SparseConv3d "indice pairs", call sites
// pcdet/models/backbones_3d/spconv_backbone.py:12-15,39-46,194-234). Restated semantics:
//   out[o] = sum_k W[k] * in[o*stride - pad + k]      (k = (kz,ky,kx) row-major, cross-correlation)
//   SubM: active outputs == active inputs (stride 1, pad = k/2).
//   SparseConv: o is active iff at least one active input falls in its window; out_shape = (in+2p-k)/s + 1.
//
// Layout choice for MI355X: instead of spconv's per-offset (in,out) pair lists + scatter-add, we build an
// OUTPUT-STATIONARY gather table nbr[out_row][k] = in_row | -1 (one coalesced i32 row per output site).
// The conv kernel then owns whole output rows: no atomics, deterministic fp32 sums, fused epilogue.
#include "common.h"
#include "scan.h"

#define RB_BLOCK 256

__device__ __forceinline__ int64_t coord_key(int b, int z, int y, int x, int sz, int sy, int sx) {
    return (((int64_t)b * sz + z) * sy + y) * sx + x;
}

__global__ __launch_bounds__(RB_BLOCK) void hash_build_kernel(const int *__restrict__ coords, const int *count, int cap,
                                                              int sz, int sy, int sx, int64_t *keys, int *vals,
                                                              int table_size) {
    int n = *count;
    if (n > cap) n = cap;
    for (int i = blockIdx.x * RB_BLOCK + threadIdx.x; i < n; i += gridDim.x * RB_BLOCK) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        int slot = aq3d_hash_insert(keys, table_size, coord_key(c.x, c.y, c.z, c.w, sz, sy, sx));
        if (slot >= 0) vals[slot] = i;
    }
}

extern "C" int aq3d_hash_build(const int32_t *coords, const int32_t *count, int32_t cap, const int32_t shape[3],
                               int64_t *table_keys, int32_t *table_vals, int32_t table_size, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(coords && count && table_keys && table_vals, "aq3d_hash_build: null argument");
    AQ3D_REQUIRE(aq3d_is_pow2(table_size) && table_size >= 2 * cap, "aq3d_hash_build: table_size=%d must be pow2 >= 2*cap=%d",
                 table_size, 2 * cap);
    hipMemsetAsync(table_keys, 0xff, (size_t)table_size * 8, stream);
    hipLaunchKernelGGL(hash_build_kernel, dim3(aq3d_grid(cap, RB_BLOCK)), dim3(RB_BLOCK), 0, stream, coords, count, cap,
                       shape[0], shape[1], shape[2], table_keys, table_vals, table_size);
    AQ3D_CHECK_LAUNCH("aq3d_hash_build");
    return AQ3D_OK;
}

// one thread per (site, k): 27 consecutive lanes share a site so the nbr row is written coalesced.
__global__ __launch_bounds__(RB_BLOCK) void subm_nbr_kernel(const int *__restrict__ coords, const int *count, int cap,
                                                            int sz, int sy, int sx, int kz, int ky, int kx,
                                                            const int64_t *__restrict__ keys,
                                                            const int *__restrict__ vals, int table_size,
                                                            int *__restrict__ nbr) {
    int n = *count;
    if (n > cap) n = cap;
    const int kvol = kz * ky * kx;
    const long long total = (long long)n * kvol;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        int m = (int)(t / kvol), k = (int)(t - (long long)m * kvol);
        int dz = k / (ky * kx) - kz / 2;
        int dy = (k / kx) % ky - ky / 2;
        int dx = k % kx - kx / 2;
        const int4 c = reinterpret_cast<const int4 *>(coords)[m];
        int z = c.y + dz, y = c.z + dy, x = c.w + dx;
        int r = -1;
        if (z >= 0 && z < sz && y >= 0 && y < sy && x >= 0 && x < sx) {
            int slot = aq3d_hash_find(keys, table_size, coord_key(c.x, z, y, x, sz, sy, sx));
            if (slot >= 0) r = vals[slot];
        }
        nbr[t] = r;
    }
}

extern "C" int aq3d_subm_neighbors(const int32_t *coords, const int32_t *count, int32_t cap, const int32_t shape[3],
                                   const int32_t ksize[3], const int64_t *table_keys, const int32_t *table_vals,
                                   int32_t table_size, int32_t *nbr_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(coords && count && table_keys && table_vals && nbr_out, "aq3d_subm_neighbors: null argument");
    AQ3D_REQUIRE((ksize[0] & 1) && (ksize[1] & 1) && (ksize[2] & 1), "aq3d_subm_neighbors: ksize must be odd");
    const int kvol = ksize[0] * ksize[1] * ksize[2];
    hipLaunchKernelGGL(subm_nbr_kernel, dim3(aq3d_grid((int64_t)cap * kvol, RB_BLOCK, 4096)), dim3(RB_BLOCK), 0, stream,
                       coords, count, cap, shape[0], shape[1], shape[2], ksize[0], ksize[1], ksize[2], table_keys,
                       table_vals, table_size, nbr_out);
    AQ3D_CHECK_LAUNCH("aq3d_subm_neighbors");
    return AQ3D_OK;
}

// ---------------- strided conv: output set + gather table ----------------
struct ConvGeom {
    int isz, isy, isx;  // input shape
    int osz, osy, osx;  // output shape
    int kz, ky, kx, sz, sy, sx, pz, py, px;
};

struct ConvOutWS {
    int *first;       // [out_table] smallest candidate id per output slot
    int *cslot;       // [cap_in * kvol] output slot of candidate, -1 invalid
    int *flags;       // [cap_in * kvol]
    int *prefix;      // [cap_in * kvol]
    int *chunk_sums;
    int *total;
};

static size_t rb_align(size_t v) { return (v + 255) / 256 * 256; }

extern "C" size_t aq3d_conv_outputs_workspace_bytes(int32_t cap_in, int32_t kvol, int32_t out_table_size) {
    size_t nc = (size_t)cap_in * kvol;
    return rb_align((size_t)out_table_size * 4) + 3 * rb_align(nc * 4) +
           rb_align((size_t)aq3d_scan_chunks_for((int)nc) * 4 + 4) + 256;
}

// candidate (m, k): input m contributes to output o = (i + p - k) / s when divisible and in range.
__global__ __launch_bounds__(RB_BLOCK) void conv_candidates_kernel(const int *__restrict__ coords, const int *count,
                                                                   int cap, ConvGeom g, int64_t *out_keys,
                                                                   int out_table_size, ConvOutWS w) {
    int n = *count;
    if (n > cap) n = cap;
    const int kvol = g.kz * g.ky * g.kx;
    const long long total = (long long)n * kvol;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        int m = (int)(t / kvol), k = (int)(t - (long long)m * kvol);
        int kzz = k / (g.ky * g.kx), kyy = (k / g.kx) % g.ky, kxx = k % g.kx;
        const int4 c = reinterpret_cast<const int4 *>(coords)[m];
        int nz = c.y + g.pz - kzz, ny = c.z + g.py - kyy, nx = c.w + g.px - kxx;
        int slot = -1;
        if (nz >= 0 && ny >= 0 && nx >= 0 && nz % g.sz == 0 && ny % g.sy == 0 && nx % g.sx == 0) {
            int oz = nz / g.sz, oy = ny / g.sy, ox = nx / g.sx;
            if (oz < g.osz && oy < g.osy && ox < g.osx) {
                slot = aq3d_hash_insert(out_keys, out_table_size, coord_key(c.x, oz, oy, ox, g.osz, g.osy, g.osx));
                if (slot >= 0) atomicMin(&w.first[slot], (int)t);
            }
        }
        w.cslot[t] = slot;
    }
}

__global__ __launch_bounds__(RB_BLOCK) void conv_flags_kernel(const int *count, int cap, int kvol, ConvOutWS w) {
    int n = *count;
    if (n > cap) n = cap;
    const long long total = (long long)n * kvol;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        int s = w.cslot[t];
        w.flags[t] = (s >= 0 && w.first[s] == (int)t) ? 1 : 0;
    }
}

__global__ __launch_bounds__(RB_BLOCK) void conv_emit_kernel(const int *__restrict__ coords_in, const int *count, int cap,
                                                             ConvGeom g, const int64_t *__restrict__ out_keys,
                                                             int *__restrict__ out_vals, ConvOutWS w,
                                                             int *__restrict__ coords_out, int cap_out, int *count_out) {
    int n = *count;
    if (n > cap) n = cap;
    const int kvol = g.kz * g.ky * g.kx;
    const long long total = (long long)n * kvol;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int m = *w.total;
        *count_out = m < cap_out ? m : cap_out;
    }
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        if (!w.flags[t]) continue;
        int row = w.prefix[t];
        int s = w.cslot[t];
        if (row >= cap_out) { out_vals[s] = -1; continue; }
        out_vals[s] = row;
        int64_t key = out_keys[s];
        int ox = (int)(key % g.osx); key /= g.osx;
        int oy = (int)(key % g.osy); key /= g.osy;
        int oz = (int)(key % g.osz); key /= g.osz;
        reinterpret_cast<int4 *>(coords_out)[row] = make_int4((int)key, oz, oy, ox);
    }
}

// one thread per (out row, k): nbr = input row at o*s - p + k
__global__ __launch_bounds__(RB_BLOCK) void conv_nbr_kernel(const int *__restrict__ coords_out, const int *count_out,
                                                            int cap_out, ConvGeom g, const int64_t *__restrict__ in_keys,
                                                            const int *__restrict__ in_vals, int in_table_size,
                                                            int *__restrict__ nbr) {
    int n = *count_out;
    if (n > cap_out) n = cap_out;
    const int kvol = g.kz * g.ky * g.kx;
    const long long total = (long long)n * kvol;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        int m = (int)(t / kvol), k = (int)(t - (long long)m * kvol);
        int kzz = k / (g.ky * g.kx), kyy = (k / g.kx) % g.ky, kxx = k % g.kx;
        const int4 c = reinterpret_cast<const int4 *>(coords_out)[m];
        int z = c.y * g.sz - g.pz + kzz, y = c.z * g.sy - g.py + kyy, x = c.w * g.sx - g.px + kxx;
        int r = -1;
        if (z >= 0 && z < g.isz && y >= 0 && y < g.isy && x >= 0 && x < g.isx) {
            int slot = aq3d_hash_find(in_keys, in_table_size, coord_key(c.x, z, y, x, g.isz, g.isy, g.isx));
            if (slot >= 0) r = in_vals[slot];
        }
        nbr[t] = r;
    }
}

extern "C" int aq3d_conv_outputs(const int32_t *coords_in, const int32_t *count_in, int32_t cap_in,
                                 const int32_t shape_in[3], const int32_t ksize[3], const int32_t stride[3],
                                 const int32_t pad[3], const int64_t *in_keys, const int32_t *in_vals,
                                 int32_t in_table_size, int32_t *coords_out, int32_t *count_out, int32_t cap_out,
                                 int32_t *nbr_out, int64_t *out_keys, int32_t *out_vals, int32_t out_table_size,
                                 void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(coords_in && count_in && in_keys && in_vals && coords_out && count_out && nbr_out && out_keys &&
                     out_vals && workspace, "aq3d_conv_outputs: null argument");
    AQ3D_REQUIRE(aq3d_is_pow2(out_table_size) && out_table_size >= 2 * cap_out,
                 "aq3d_conv_outputs: out_table_size=%d must be pow2 >= 2*cap_out=%d", out_table_size, 2 * cap_out);
    ConvGeom g;
    g.isz = shape_in[0]; g.isy = shape_in[1]; g.isx = shape_in[2];
    g.kz = ksize[0]; g.ky = ksize[1]; g.kx = ksize[2];
    g.sz = stride[0]; g.sy = stride[1]; g.sx = stride[2];
    g.pz = pad[0]; g.py = pad[1]; g.px = pad[2];
    AQ3D_REQUIRE(g.sz > 0 && g.sy > 0 && g.sx > 0 && g.kz > 0 && g.ky > 0 && g.kx > 0, "aq3d_conv_outputs: bad geometry");
    g.osz = (g.isz + 2 * g.pz - g.kz) / g.sz + 1;
    g.osy = (g.isy + 2 * g.py - g.ky) / g.sy + 1;
    g.osx = (g.isx + 2 * g.px - g.kx) / g.sx + 1;
    const int kvol = g.kz * g.ky * g.kx;
    const size_t nc = (size_t)cap_in * kvol;
    AQ3D_REQUIRE(nc < (size_t)0x7fffffff, "aq3d_conv_outputs: cap_in*kvol overflows int32");
    if (workspace_bytes < aq3d_conv_outputs_workspace_bytes(cap_in, kvol, out_table_size)) {
        aq3d_set_error("aq3d_conv_outputs: workspace %zu < %zu bytes", workspace_bytes,
                       aq3d_conv_outputs_workspace_bytes(cap_in, kvol, out_table_size));
        return AQ3D_ERR_WORKSPACE;
    }
    ConvOutWS w;
    char *p = (char *)workspace;
    w.first = (int *)p;  p += rb_align((size_t)out_table_size * 4);
    w.cslot = (int *)p;  p += rb_align(nc * 4);
    w.flags = (int *)p;  p += rb_align(nc * 4);
    w.prefix = (int *)p; p += rb_align(nc * 4);
    w.chunk_sums = (int *)p; p += rb_align((size_t)aq3d_scan_chunks_for((int)nc) * 4 + 4);
    w.total = (int *)p;

    hipMemsetAsync(out_keys, 0xff, (size_t)out_table_size * 8, stream);
    hipMemsetAsync(w.first, 0x7f, (size_t)out_table_size * 4, stream);
    const int grid_c = aq3d_grid((int64_t)nc, RB_BLOCK, 4096);
    hipLaunchKernelGGL(conv_candidates_kernel, dim3(grid_c), dim3(RB_BLOCK), 0, stream, coords_in, count_in, cap_in, g,
                       out_keys, out_table_size, w);
    hipLaunchKernelGGL(conv_flags_kernel, dim3(grid_c), dim3(RB_BLOCK), 0, stream, count_in, cap_in, kvol, w);
    int rc = aq3d_scan_flags(w.flags, count_in, kvol, (int)nc, w.prefix, w.total, w.chunk_sums, stream);
    if (rc != AQ3D_OK) return rc;
    hipLaunchKernelGGL(conv_emit_kernel, dim3(grid_c), dim3(RB_BLOCK), 0, stream, coords_in, count_in, cap_in, g, out_keys,
                       out_vals, w, coords_out, cap_out, count_out);
    hipLaunchKernelGGL(conv_nbr_kernel, dim3(aq3d_grid((int64_t)cap_out * kvol, RB_BLOCK, 4096)), dim3(RB_BLOCK), 0,
                       stream, coords_out, count_out, cap_out, g, in_keys, in_vals, in_table_size, nbr_out);
    AQ3D_CHECK_LAUNCH("aq3d_conv_outputs");
    return AQ3D_OK;
}

// ---------------- VoxelNeXt bev_out: union of up to 3 index levels projected to 2-D, features summed per site ----------
// Reference: VoxelResBackBone8xVoxelNeXt.forward + bev_out (pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:149-198):
//   x5.indices[:, 1:] *= 2; x6.indices[:, 1:] *= 4; cat(x4, x5, x6); drop z; torch.unique(dim=0) + index_add_.
// Here: candidates (level, row) are inserted into a 2-D hash, the first candidate of a site names its output row
// (deterministic order, no sort), and every output row then GATHERS its contributions through the levels' 3-D hash tables in
// a fixed (level, z) order — no float atomics, bit-reproducible sums.
struct MergeLevels {
    const int *coords[3];
    const int *count[3];
    const float *feats[3];
    const int64_t *keys[3];
    const int *vals[3];
    int table_size[3];
    int sz[3], sy[3], sx[3];   // 3-D shapes of the levels
    int scale[3];
    int n_levels, cap_in, C;
};

__global__ __launch_bounds__(RB_BLOCK) void merge_candidates_kernel(MergeLevels L, int H, int W, int64_t *out_keys,
                                                                   int out_table_size, int *first, int *cslot) {
    const long long total = (long long)L.n_levels * L.cap_in;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        const int l = (int)(t / L.cap_in), m = (int)(t - (long long)l * L.cap_in);
        int n = *L.count[l];
        if (n > L.cap_in) n = L.cap_in;
        int slot = -1;
        if (m < n) {
            const int4 c = reinterpret_cast<const int4 *>(L.coords[l])[m];
            const int y = c.z * L.scale[l], x = c.w * L.scale[l];
            if (y >= 0 && y < H && x >= 0 && x < W) {
                slot = aq3d_hash_insert(out_keys, out_table_size, coord_key(c.x, 0, y, x, 1, H, W));
                if (slot >= 0) atomicMin(&first[slot], (int)t);
            }
        }
        cslot[t] = slot;
    }
}

__global__ __launch_bounds__(RB_BLOCK) void merge_flags_kernel(long long total, const int *first, const int *cslot, int *flags,
                                                              int *n_dev) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_dev = (int)total;   // scan length as a device scalar
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        const int s = cslot[t];
        flags[t] = (s >= 0 && first[s] == (int)t) ? 1 : 0;
    }
}

__global__ __launch_bounds__(RB_BLOCK) void merge_emit_kernel(long long total, int H, int W, const int64_t *out_keys, int *out_vals,
                                                             const int *flags, const int *prefix, const int *cslot,
                                                             const int *total_out, int *coords_out, int cap_out, int *count_out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = *total_out < cap_out ? *total_out : cap_out;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        if (!flags[t]) continue;
        const int row = prefix[t], s = cslot[t];
        if (row >= cap_out) { out_vals[s] = -1; continue; }
        out_vals[s] = row;
        int64_t key = out_keys[s];
        const int x = (int)(key % W); key /= W;
        const int y = (int)(key % H); key /= H;
        reinterpret_cast<int4 *>(coords_out)[row] = make_int4((int)key, 0, y, x);
    }
}

// thread = (output row, 4 channels): sum of the contributing rows in (level, z) order
__global__ __launch_bounds__(RB_BLOCK) void merge_sum_kernel(MergeLevels L, const int *coords_out, const int *count_out, int cap_out,
                                                            float *feats_out) {
    int n = *count_out;
    if (n > cap_out) n = cap_out;
    const int c4n = L.C >> 2;
    const long long total = (long long)n * c4n;
    for (long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * RB_BLOCK) {
        const int row = (int)(t / c4n), c4 = (int)(t - (long long)row * c4n);
        const int4 c = reinterpret_cast<const int4 *>(coords_out)[row];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < L.n_levels; ++l) {
            const int sc = L.scale[l];
            if (c.z % sc != 0 || c.w % sc != 0) continue;
            const int y = c.z / sc, x = c.w / sc;
            if (y >= L.sy[l] || x >= L.sx[l]) continue;
            for (int z = 0; z < L.sz[l]; ++z) {
                const int slot = aq3d_hash_find(L.keys[l], L.table_size[l], coord_key(c.x, z, y, x, L.sz[l], L.sy[l], L.sx[l]));
                if (slot < 0) continue;
                const float4 v = *reinterpret_cast<const float4 *>(L.feats[l] + (size_t)L.vals[l][slot] * L.C + c4 * 4);
                acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y);
                acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
            }
        }
        *reinterpret_cast<float4 *>(feats_out + (size_t)row * L.C + c4 * 4) = acc;
    }
}

extern "C" size_t aq3d_bev_merge_workspace_bytes(int32_t n_levels, int32_t cap_in, int32_t out_table_size) {
    const size_t nc = (size_t)n_levels * cap_in;
    return rb_align((size_t)out_table_size * 4) + 3 * rb_align(nc * 4) + rb_align((size_t)aq3d_scan_chunks_for((int)nc) * 4 + 4) + 512;
}

extern "C" int aq3d_bev_merge(const int32_t *const *coords, const int32_t *const *counts, const float *const *feats,
                              const int64_t *const *keys, const int32_t *const *vals, const int32_t *table_sizes,
                              const int32_t *shapes /* [n_levels][3] */, const int32_t *scales, int32_t n_levels, int32_t cap_in,
                              int32_t C, int32_t H, int32_t W, int32_t *coords_out, int32_t *count_out, float *feats_out,
                              int32_t cap_out, int64_t *out_keys, int32_t *out_vals, int32_t out_table_size, void *workspace,
                              size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(coords && counts && feats && keys && vals && table_sizes && shapes && scales && coords_out && count_out &&
                     feats_out && out_keys && out_vals && workspace, "aq3d_bev_merge: null argument");
    AQ3D_REQUIRE(n_levels >= 1 && n_levels <= 3 && cap_in > 0 && cap_out > 0 && C >= 4 && (C & 3) == 0 && H > 0 && W > 0,
                 "aq3d_bev_merge: bad sizes");
    AQ3D_REQUIRE(aq3d_is_pow2(out_table_size) && out_table_size >= 2 * cap_out, "aq3d_bev_merge: out_table_size=%d must be pow2 >= 2*cap_out", out_table_size);
    if (workspace_bytes < aq3d_bev_merge_workspace_bytes(n_levels, cap_in, out_table_size)) {
        aq3d_set_error("aq3d_bev_merge: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    MergeLevels L;
    for (int l = 0; l < 3; ++l) {
        const int k = l < n_levels ? l : 0;
        L.coords[l] = coords[k]; L.count[l] = counts[k]; L.feats[l] = feats[k]; L.keys[l] = keys[k]; L.vals[l] = vals[k];
        L.table_size[l] = table_sizes[k]; L.sz[l] = shapes[3 * k]; L.sy[l] = shapes[3 * k + 1]; L.sx[l] = shapes[3 * k + 2];
        L.scale[l] = scales[k];
        AQ3D_REQUIRE(L.scale[l] >= 1, "aq3d_bev_merge: scale must be >= 1");
    }
    L.n_levels = n_levels; L.cap_in = cap_in; L.C = C;
    const size_t nc = (size_t)n_levels * cap_in;
    char *p = (char *)workspace;
    int *first = (int *)p;  p += rb_align((size_t)out_table_size * 4);
    int *cslot = (int *)p;  p += rb_align(nc * 4);
    int *flags = (int *)p;  p += rb_align(nc * 4);
    int *prefix = (int *)p; p += rb_align(nc * 4);
    int *chunk_sums = (int *)p; p += rb_align((size_t)aq3d_scan_chunks_for((int)nc) * 4 + 4);
    int *total = (int *)p;
    int *n_dev = total + 1;
    hipMemsetAsync(out_keys, 0xff, (size_t)out_table_size * 8, stream);
    hipMemsetAsync(first, 0x7f, (size_t)out_table_size * 4, stream);
    const int grid = aq3d_grid((int64_t)nc, RB_BLOCK, 4096);
    hipLaunchKernelGGL(merge_candidates_kernel, dim3(grid), dim3(RB_BLOCK), 0, stream, L, H, W, out_keys, out_table_size, first, cslot);
    hipLaunchKernelGGL(merge_flags_kernel, dim3(grid), dim3(RB_BLOCK), 0, stream, (long long)nc, first, cslot, flags, n_dev);
    int rc = aq3d_scan_flags(flags, n_dev, 1, (int)nc, prefix, total, chunk_sums, stream);
    if (rc != AQ3D_OK) return rc;
    hipLaunchKernelGGL(merge_emit_kernel, dim3(grid), dim3(RB_BLOCK), 0, stream, (long long)nc, H, W, out_keys, out_vals, flags, prefix,
                       cslot, total, coords_out, cap_out, count_out);
    hipLaunchKernelGGL(merge_sum_kernel, dim3(aq3d_grid((int64_t)cap_out * (C / 4), RB_BLOCK, 4096)), dim3(RB_BLOCK), 0, stream, L,
                       coords_out, count_out, cap_out, feats_out);
    AQ3D_CHECK_LAUNCH("aq3d_bev_merge");
    return AQ3D_OK;
}
