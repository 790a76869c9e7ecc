/*
 * aq3d.h — C ABI of libaq3d.so: the MI355X (gfx950) hot path of quantized CenterPoint inference.
 *
 * Every entry point is `extern "C"`, takes raw device addresses (tensor.data_ptr()) and explicit
 * sizes, enqueues its work on the given hipStream_t (passed as void*), never allocates, never frees,
 * never synchronises and never calls exit(): it returns 0 on success or a negative aq3d_status.
 * (The reference's native op for this path aborts the process through CHECK_* macros,
 * pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:14-26 — this ABI returns codes instead.)
 *
 * Dynamic sizes (number of voxels, of active sites, of boxes) live in DEVICE memory (`const int *count`)
 * next to a host-known capacity, so that a whole frame can be enqueued — or captured into a hipGraph —
 * without a device→host round trip. Kernels run grid-stride over the capacity and stop at *count.
 *
 * Reference interfaces replaced (file:line relative to the reference tree):
 *   aq3d_voxelize            spconv Point2VoxelCPU3d.point_to_voxel as called from
 *                            pcdet/datasets/processor/data_processor.py:45-61 (+ MeanVFE, vfe/mean_vfe.py:25-29)
 *   aq3d_mean_vfe            pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31
 *   aq3d_level_from_points / aq3d_level_from_coords / aq3d_level_downsample / aq3d_level_neighbors
 *                            spconv.pytorch.SubMConv3d / SparseConv3d index generation, call sites
 *                            pcdet/models/backbones_3d/spconv_backbone.py:12-15,39-46,194-234 (and the voxel generator fused
 *                            with it for raw point input)
 *   aq3d_absmax_cols / aq3d_quantize_rows
 *                            pytorch_quantization TensorQuantizer(axis=1) as used in quant/quant.py:21-26,50-52
 *   aq3d_spconv              spconv gather-GEMM-scatter behind QConvNd.forward, quant/quant.py:36-58,
 *                            with BN1d/ReLU/residual of spconv_backbone.py:8-27,51-67 fused
 *   aq3d_scatter_dense       SparseConvTensor.dense() + view, map_to_bev/height_compression.py:19-23
 *   aq3d_piece_absmax / aq3d_sq_prep / aq3d_conv2d_i8
 *                            SQConv2d.forward, quant/smoothquant.py:38-99
 *   aq3d_conv2d_f32          nn.Conv2d / nn.ConvTranspose2d + BatchNorm2d + ReLU of
 *                            pcdet/models/backbones_2d/base_bev_backbone.py:30-66 and
 *                            pcdet/models/dense_heads/center_head.py:22-29,75-82 (the layers left in fp32)
 *   aq3d_center_decode       centernet_utils.decode_bbox_from_heatmap, model_utils/centernet_utils.py:155-241
 *   aq3d_nms_rotated         iou3d_nms_utils.nms_gpu → iou3d_nms.cpp:137-183 + iou3d_nms_kernel.cu:295-339
 *   aq3d_level_merge(_sum)   VoxelResBackBone8xVoxelNeXt.bev_out, spconv_backbone_voxelnext.py:149-198
 *   aq3d_voxel_decode        centernet_utils.decode_bbox_from_voxels_nuscenes, model_utils/centernet_utils.py:243-354
 *   aq3d_boxes_iou3d / aq3d_recall_counts
 *                            iou3d_nms_utils.boxes_iou3d_gpu (iou3d_nms_utils.py:48-81, kernel iou3d_nms_kernel.cu:236-249) and
 *                            the recall bookkeeping of Detector3DTemplate.generate_recall_record (detector3d_template.py:286-328)
 */
#ifndef AQ3D_H
#define AQ3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    AQ3D_OK = 0,
    AQ3D_ERR_INVALID_ARG = -1,
    AQ3D_ERR_WORKSPACE = -2,
    AQ3D_ERR_LAUNCH = -3,
    AQ3D_ERR_UNSUPPORTED = -4
} aq3d_status;

/* Human-readable description of the last error on the calling thread ("" if none). */
const char *aq3d_last_error_string(void);
/* ABI version, bumped on any signature change. */
int aq3d_abi_version(void);

/* ---------------------------------------------------------------- voxelization ---------------- */

typedef struct {
    float range_min[3];       /* x, y, z lower bound of POINT_CLOUD_RANGE */
    float voxel_size[3];      /* x, y, z VOXEL_SIZE                        */
    int32_t grid[3];          /* nx, ny, nz                                */
    int32_t num_features;     /* C: columns kept per point (xyz + used features), 3..8 */
    int32_t point_stride;     /* floats per input point row (>= C, + 1 if batch column) */
    int32_t batch_col;        /* 1: column 0 of each row is the batch index (pcdet `points [N,1+C]`) and is skipped */
    int32_t max_points_per_voxel; /* P (MAX_POINTS_PER_VOXEL), 1..16 */
    int32_t max_voxels;       /* MAX_NUMBER_OF_VOXELS of one sample; voxels beyond it (first-seen order) are dropped */
} aq3d_voxel_cfg;

/* Bytes of scratch needed by aq3d_voxelize for up to n_points points. */
size_t aq3d_voxelize_workspace_bytes(int32_t n_points);

/*
 * Hard voxelization + mean VFE in one pass.
 * One call voxelizes ONE sample (as the reference does per sample in its dataset worker); a batch is
 * built by calling it once per sample with append = 1, which is what collate_batch's concatenation +
 * batch-index column amounts to (pcdet/datasets/dataset.py:232-244).
 *   points      f32 [n_points, point_stride] (column 0 skipped when cfg->batch_col)
 *   batch_index value written to column 0 of the coords rows
 *   append      0: rows start at 0; 1: rows start at the current *count_io
 *   coords_out  i32 [cap, 4] rows (b, z, y, x), voxels in first-seen point order
 *   feats_out   f32 [cap, C] mean over the first P points (in point order) of each voxel
 *   npts_out    i32 [cap] points kept per voxel (<= P); may be NULL
 *   count_io    i32 [1] device: total number of rows after the call
 */
int aq3d_voxelize(const float *points, int32_t n_points, const aq3d_voxel_cfg *cfg, int32_t batch_index,
                  int32_t append, int32_t *coords_out, float *feats_out, int32_t *npts_out, int32_t *count_io,
                  int32_t cap, void *workspace, size_t workspace_bytes, void *stream);

/* voxels f32 [m, P, C], num_points f32 [m] → feats f32 [*, C]: row row_of_voxel[v] (v itself when NULL; -1 skips the voxel)
 * = sum / max(num,1). */
int aq3d_mean_vfe(const float *voxels, const float *num_points, int32_t m, int32_t P, int32_t C,
                  const int32_t *row_of_voxel, float *feats_out, void *stream);

/* ---------------------------------------------------------------- sparse index levels ---------- */

/* One index level = the set of active sites of one sparse tensor (what spconv keeps as SparseConvTensor.indices + its
 * indice_dict). All arrays are device memory owned by the caller:
 *   bits        u64 [aq3d_level_words(shape, batch)]  occupancy; word w <-> (b, z, y/8, x/8), bit (y%8)*8 + x%8
 *   base        i32 [words]   number of sites in the words before w (raster rank of the word's first site)
 *   row_of_rank i32 [cap]     raster rank -> row
 *   coords      i32 [cap, 4]  rows (b, z, y, x)
 *   count       i32 [1]       live rows = min(total[0], cap)
 *   total       i32 [2]       [0] number of active sites (UNclamped: > cap means the level overflowed its buffers and the
 *                             frame is invalid), [1] the largest per-sample number of sites
 * Row order: bitmap raster order (flags = 0), or — AQ3D_LEVEL_SORTED — class by class (sites grouped by which planes of their
 * 3x3x3 neighbourhood are occupied, heaviest class first, raster order inside a class) so that the 128-row tiles of the
 * gather-GEMM kernels skip most kernel offsets: three more launches per level, worth it for the levels whose convs have >= 128
 * input channels (W150k: 128-channel layers 67 -> 54 us each; 32 / 64 channels gain 2 us per layer, less than the sort costs). */
enum { AQ3D_LEVEL_SORTED = 1 };
typedef struct aq3d_level {
    int32_t shape[3];          /* z, y, x */
    int32_t batch;             /* samples (<= 64) */
    int32_t cap;               /* row capacity */
    int32_t flags;             /* AQ3D_LEVEL_SORTED */
    uint64_t *bits;
    int32_t *base;
    int32_t *row_of_rank;
    int32_t *coords;
    int32_t *count;
    int32_t *total;
} aq3d_level;

int64_t aq3d_level_words(const int32_t shape[3], int32_t batch);

/* Scratch of the index builders below for levels of up to max_cap rows and max_words words and clouds of up to max_points
 * points. The FIRST 256 BYTES must be zero before the first call (they hold self-resetting arrival counters); one
 * workspace serves every call that is ordered on one stream. */
size_t aq3d_index_workspace_bytes(int32_t max_cap, int64_t max_words, int32_t max_points);

/* aq3d_level_from_points / aq3d_level_from_coords REBUILD a level in place. They do not zero its bitmap (an 11.6 MB fill for a Waymo
 * grid that holds 30 000 voxels): they clear the words of the site set the level currently describes — coords[0 .. count). So on
 * entry bits, coords, count and total must be consistent: all zero (a freshly allocated level) or as a builder left them. A level
 * whose previous build overflowed its row capacity (total[0] > cap) is zeroed as a whole. */

/* Hard voxelization + mean VFE straight into an index level (replaces spconv's Point2VoxelCPU3d.point_to_voxel as called
 * from pcdet/datasets/processor/data_processor.py:45-61, + MeanVFE, vfe/mean_vfe.py:25-29, for the whole batch at once):
 *   points f32 [n_points, cfg->point_stride]; with cfg->batch_col the sample index is column 0, else batch_index.
 *   feats_out f32 [cap, C] mean of the first P points (in point order) of every voxel, npts_out i32 [cap] (may be NULL),
 *   rows in the level's row order. cfg->max_voxels is NOT applied: level->total[1] > max_voxels tells the caller that the
 *   reference would have dropped voxels (first-seen order) — aq3d_voxelize + aq3d_level_from_coords reproduce that case. */
int aq3d_level_from_points(const float *points, int32_t n_points, const aq3d_voxel_cfg *cfg, int32_t batch_index,
                           const aq3d_level *level, float *feats_out, int32_t *npts_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Level from given voxel coordinates (pcdet's pre-voxelized batch_dict: voxel_coords [n, 4] rows (b, z, y, x)).
 * row_of_input i32 [n]: the row of input voxel i (-1: outside the grid). */
int aq3d_level_from_coords(const int32_t *coords_in, int32_t n, const aq3d_level *level, int32_t *row_of_input,
                           void *workspace, size_t workspace_bytes, void *stream);

/* Active output set of a strided / dilating sparse conv (spconv SparseConv3d / SparseConv2d index generation):
 * out site o is active iff some active input i = o*stride - pad + k exists (0 <= k < ksize). ksize 1..3, stride 1..2,
 * pad 0..1 per axis; out->shape = (in + 2 pad - ksize) / stride + 1. */
int aq3d_level_downsample(const aq3d_level *in, const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                          const aq3d_level *out, void *workspace, size_t workspace_bytes, void *stream);

/* One gather table: nbr i32 [out->cap, kvol], nbr[m][k] = row of `in` at coords_out[m] * stride - pad + k, or -1;
 * k enumerates (kz, ky, kx) row-major. Submanifold conv: in == out, stride 1, pad = ksize / 2. */
typedef struct aq3d_nbr_job {
    const aq3d_level *in;
    int32_t ksize[3], stride[3], pad[3];
    int32_t *nbr;
} aq3d_nbr_job;
/* Up to 4 gather tables of one output level in ONE launch. */
int aq3d_level_neighbors(const aq3d_level *out, const aq3d_nbr_job *jobs, int32_t n_jobs, void *stream);

/* ---------------------------------------------------------------- sparse conv ----------------- */

/* Per-column abs-max of x f32 [*count, C] → amax f32 [C] (amax must be zeroed by the caller). */
int aq3d_absmax_cols(const float *x, const int32_t *count, int32_t cap, int32_t C, float *amax, void *stream);

/* TensorQuantizer(axis=1) on rows: scale_c = qmax/amax_c; q = clamp(rint(x*scale_c), -qmax-1, qmax).
 *   codes_out  i8  [cap, C] integer codes (may be NULL; only for bits <= 8)
 *   fq_out     f32 [cap, C] fake-quantized values q/scale_c (may be NULL) — the fp32 fallback of layers the coded path does
 *                           not take (act_bits = 16, unusual channel counts)
 *   dequant_out f32 [C]     1/scale_c                      (may be NULL)
 *   per_tensor != 0: one amax = max_c amax_c is used for every column (QuantDescriptor without axis). */
int aq3d_quantize_rows(const float *x, const float *amax, const int32_t *count, int32_t cap, int32_t C,
                       int32_t bits, int32_t per_tensor, int8_t *codes_out, float *fq_out,
                       float *dequant_out, void *stream);

enum {
    AQ3D_ACT_F32 = 0,  /* feats_in is f32 [*, Cin] used as is                       */
    AQ3D_ACT_I8 = 1    /* feats_in is i8 codes [*, Cin]; value = code * act_dequant[c] (generic kernel; production: aq3d_spconv_coded) */
};
enum {
    AQ3D_W_F32 = 0,    /* weight is f32 [kvol, Cout, Cin]                           */
    AQ3D_W_I8 = 1,     /* weight is i8 codes [kvol, Cout, Cin]; value = code * w_dequant[cout] */
    AQ3D_W_F16S = 2    /* fp32 weights already split into fp16 planes, f16 [kvol * Cout, 2, Cin] (aq3d_conv2d_split_weight of the f32
                        * [kvol * Cout, Cin] matrix): f32 activations, Cin = 128, Cout = 128 or 16 only */
};
enum {
    AQ3D_FLAG_RELU = 1,
    AQ3D_FLAG_SKIP_F32 = 2, /* aq3d_spconv_coded with next_codes: do not store the fp32 rows */
    AQ3D_FLAG_W16 = 4       /* aq3d_spconv_coded: 16-bit fixed-point weights (the two upper int8 planes of W'': |error| <= 2^-9 per
                               weight on a +-127 scale, <= 1e-3 of the output scale per layer, measured ~5e-5) — two int8 MFMAs per
                               product instead of three; default: all three planes (24-bit fixed point, fp32-grade sums) */
};

/*
 * Output-stationary gather-GEMM sparse convolution with fused epilogue (fp32 layers — conv_input, unquantised models — and
 * weight-only layers; the w8a8 QConvNd layers run through aq3d_spconv_prep + aq3d_spconv_coded below):
 *   acc[m,o]  = sum_k sum_c A[nbr[m][k], c] * W[k][c][o]                  (fp32 accumulate, MFMA)
 *   y[m,o]    = (acc * w_dequant[o] + bias[o]) * post_scale[o] + post_shift[o]   (bias / post_* may be NULL)
 *   y        += residual[m,o]                                              (may be NULL)
 *   y         = max(y, 0) if flags & AQ3D_FLAG_RELU
 *   absmax_out[o] = max(absmax_out[o], |y[m,o]|) over m                    (may be NULL; caller zeroes it)
 */
int aq3d_spconv(const void *feats_in, int32_t act_mode, const float *act_dequant,
                const void *weight, int32_t w_mode, const float *w_dequant,
                const float *bias, const float *post_scale, const float *post_shift,
                const float *residual, const int32_t *nbr, const int32_t *count_out, int32_t cap_out, int32_t cap_in,
                int32_t cin, int32_t cout, int32_t kvol, int32_t flags,
                float *feats_out, float *absmax_out, const float *next_scale, int8_t *next_codes, int32_t next_bits, void *stream);
/* next_scale / next_codes (both NULL or both set; aq3d_spconv: stem layer only, cin <= 8): static PTQ — the calibrated quantiser
 * of the layer that consumes y runs in this epilogue: next_codes i8 [cap_out, Cout] = clamp(rint(y * next_scale[o]), -2^(b-1),
 * 2^(b-1)-1), row-major, next_scale[o] = (2^(b-1)-1) / amax_o (0 where amax_o <= 2^-24). */

/* One layer of a grouped aq3d_spconv launch (the pointer arguments of aq3d_spconv). */
typedef struct aq3d_spconv_problem {
    const void *feats_in;
    const float *act_dequant;
    const void *weight;
    const float *w_dequant, *bias, *post_scale, *post_shift, *residual;
    const int32_t *nbr, *count_out;
    float *feats_out, *absmax_out;
    int32_t cout_real;                 /* output channels that exist when `cout` is padded (zero weight rows); 0 = all of them */
} aq3d_spconv_problem;

/* Up to 8 sparse convs of one shape (cin, cout, kvol, modes, capacities) as ONE launch — the per-branch SubMConv2d stacks of
 * VoxelNeXtHead's SeparateHead (pcdet/models/dense_heads/voxelnext_head.py:13-47). */
int aq3d_spconv_group(const aq3d_spconv_problem *problems, int32_t n, int32_t act_mode, int32_t w_mode, int32_t cap_out,
                      int32_t cap_in, int32_t cin, int32_t cout, int32_t kvol, int32_t flags, void *stream);

/* ------------- production path of the QConvNd layers: int8-coded activations x fixed-point weight planes on the int8 matrix pipe ----
 * Replaces quant/quant.py:36-58 (QConvNd.forward) around the spconv call sites pcdet/models/backbones_3d/
 * spconv_backbone.py:12-17,39-46:
 *   y[m,o] = sum_{k,c} (q[nbr[m][k],c] * d_c) * (wq[k,o,c] * t_o) = (sum q * W') * (S * t_o),  W' = wq * (d_c / S), S = max_c d_c.
 * aq3d_spconv_prep (ONE launch): TensorQuantizer(axis=1) of x f32 [*count, C] with the given abs-max -> codes_out i8 [cap, C]
 *   (1 byte per element, channel order), AND the per-frame weight planes wprep_out (aq3d_spconv_wprep_bytes: W'' = rint(W' * 2^16)
 *   as three balanced base-256 int8 digits, [kvol][Cout][3][C]) + out_mul_out f32 [Cout] = S * w_dequant / 2^16.
 *   weight_codes i8 [kvol, Cout, C]; bits <= 8; C a power of two.
 *   Either half may be skipped: x == NULL builds the weight planes only (static PTQ: once, the scales are constants),
 *   weight_codes == NULL quantises the rows only.
 * aq3d_spconv_coded: the gather-GEMM (v_mfma_i32_16x16x64_i8, one int32 accumulator per plane) with the epilogue of aq3d_spconv
 *   (incl. next_scale / next_codes / AQ3D_FLAG_SKIP_F32); AQ3D_FLAG_W16 multiplies the two upper planes only (see the flag). */
int aq3d_spconv_coded_supported(int32_t cin, int32_t cout, int32_t kvol);
size_t aq3d_spconv_wprep_bytes(int32_t cin, int32_t cout, int32_t kvol);
int aq3d_spconv_prep(const float *x, const float *amax, const int32_t *count, int32_t cap, int32_t C, int32_t bits,
                     int32_t per_tensor, const int8_t *weight_codes, const float *w_dequant, int32_t kvol, int32_t cout,
                     int8_t *codes_out, void *wprep_out, float *out_mul_out, void *stream);
int aq3d_spconv_coded(const int8_t *codes, int32_t cap_in, const void *wprep, const float *out_mul, const float *bias,
                      const float *post_scale, const float *post_shift, const float *residual, const int32_t *nbr,
                      const int32_t *count_out, int32_t cap_out, int32_t cin, int32_t cout, int32_t kvol,
                      int32_t flags, float *feats_out, float *absmax_out, const float *next_scale, int8_t *next_codes,
                      int32_t next_bits, void *stream);

/* ---------------------------------------------------------------- sparse → dense -------------- */

/* out f32 NHWC [B, H, W, C*D] zero-filled then out[b, y, x, c*D + z] = feats[m, c]. shape = (D, H, W). */
int aq3d_scatter_dense(const float *feats, const int32_t *coords, const int32_t *count, int32_t cap,
                       int32_t C, const int32_t shape[3], int32_t batch_size, float *out, void *stream);

/* ---------------------------------------------------------------- dense BEV convs ------------- */

/* x f32 NHWC [H, W, C] (one frame) → pieces f32 [16, C]: abs-max of x over each of the 16 classes
 * (row class) x (col class), class = 0: first, 1: last, 2: interior even, 3: interior odd index.
 * From these the per-(c,kh,kw) im2col column abs-max of any 3x3 / 1x1 conv with stride 1 or 2 follows exactly.
 * pieces must be zeroed by the caller. */
int aq3d_piece_absmax(const float *x, int32_t H, int32_t W, int32_t C, int32_t ldc, float *pieces, void *stream);

typedef struct {
    int32_t H, W;            /* input height / width                     */
    int32_t Cin, Cout;
    int32_t ksize;           /* 1 or 3 (square)                          */
    int32_t stride;          /* 1 or 2                                   */
    int32_t pad;             /* zero padding on every side               */
    int32_t Ho, Wo;          /* output height / width                    */
    int32_t in_ld;           /* floats between consecutive input pixels  (>= Cin)  */
    int32_t out_ld;          /* floats between consecutive output pixels (>= Cout) */
    int32_t out_coff;        /* channel offset inside the output pixel (concat fusion) */
    int32_t deconv;          /* f32 conv only: ConvTranspose2d with kernel = stride = deconv (1 or 2), ksize must be 1 */
    int32_t relu;
} aq3d_conv2d_geom;

/* Bytes of the per-layer, per-frame parameter block written by aq3d_sq_prep and read by aq3d_conv2d_i8. */
size_t aq3d_sq_param_bytes(int32_t Cin, int32_t Cout, int32_t ksize);

/*
 * Dynamic SmoothQuant + int8 quantisation parameters of one conv for the current frame.
 *   pieces     f32 [16, Cin]   from aq3d_piece_absmax of the layer input
 *   weight     f32 [Cout, K]   K = ksize*ksize*Cin ordered (kh, kw, c)
 *   w_colmax   f32 [K]         max over Cout of |weight| per column (constant, computed once)
 *   alpha      SmoothQuant migration strength (scaling_factor); alpha < 0 disables smoothing (s = 1)
 *   static_amax_x > 0 / static_amax_w != NULL: calibrated amax values instead of dynamic ones
 *   amax_x_acc f32 [1] accumulator for the dynamic per-tensor activation amax; the caller zeroes it once per frame
 *   params     device block of aq3d_sq_param_bytes(): s[K], 1/s[K], out_mul[Cout], amax_w[Cout], scalars, wq i8 [Cout,K]
 */
int aq3d_sq_prep(const float *pieces, const float *weight, const float *w_colmax,
                 const aq3d_conv2d_geom *g, float alpha, int32_t w_bits, int32_t a_bits,
                 float static_amax_x, const float *static_amax_w, float *amax_x_acc, void *params, void *stream);

/* One layer of a grouped aq3d_sq_prep launch (same arguments; the layers must share H, W, Cin, ksize, stride, pad). */
typedef struct aq3d_sq_prep_problem {
    const float *pieces, *weight, *w_colmax;
    const aq3d_conv2d_geom *geom;
    float alpha, static_amax_x;
    const float *static_amax_w;
    float *amax_x_acc;
    void *params;
} aq3d_sq_prep_problem;

/* aq3d_sq_prep for up to 8 layers in two launches instead of 2 n (the branches of one CenterHead SeparateHead). */
int aq3d_sq_prep_group(const aq3d_sq_prep_problem *problems, int32_t n, int32_t w_bits, int32_t a_bits, void *stream);

/* y = epilogue( int32 GEMM of quantised im2col(x) with quantised weights ) — SQConv2d.forward.
 *   epilogue: v = acc * out_mul[o] + bias[o]; y = v * post_scale[o] + post_shift[o]; relu.
 *   pieces_out (may be NULL): f32 [16, pieces_ld] piece abs-max accumulator of the OUTPUT tensor (same definition as
 *   aq3d_piece_absmax over the output image, column out_coff + o), so that the next SmoothQuant conv needs no extra pass;
 *   zeroed by the caller once per frame. pieces_mode 0: the 16 classes exactly; 1: interior pixels are not separated
 *   by parity — each workgroup's interior abs-max goes to one of the four interior classes (their maximum is the
 *   abs-max over the whole interior): exact for every consumer with stride 1, whose tap windows always take the max over
 *   all four, and much cheaper to produce. */
int aq3d_conv2d_i8(const float *x, const void *params, const aq3d_conv2d_geom *g,
                   const float *bias, const float *post_scale, const float *post_shift,
                   float *y, float *pieces_out, int32_t pieces_ld, int32_t pieces_mode, void *stream);

/* One convolution of a grouped launch (the arguments of aq3d_conv2d_i8 / aq3d_conv2d_f32). */
typedef struct aq3d_conv2d_problem {
    const float *x;
    const void *weight_or_params;      /* i8: the aq3d_sq_prep parameter block; f32: weight f32 [Cout_total, K] */
    const aq3d_conv2d_geom *geom;
    const float *bias, *post_scale, *post_shift;
    float *y;
    float *pieces_out;
    int32_t pieces_ld, pieces_mode;
    const void *weight_split;          /* f32, optional: aq3d_conv2d_split_weight of the same weights (used by 1x1 / deconv GEMMs) */
} aq3d_conv2d_problem;

/* Up to 8 independent convolutions of one launch class (same H, W, Cin, ksize, stride, pad; Cout on the same side of 128
 * for i8, same row class for f32) as ONE kernel launch — the per-branch convs of CenterHead's SeparateHead
 * (pcdet/models/dense_heads/center_head.py:12-46), each too small to fill the chip on its own. */
int aq3d_conv2d_i8_group(const aq3d_conv2d_problem *problems, int32_t n, void *stream);
int aq3d_conv2d_f32_group(const aq3d_conv2d_problem *problems, int32_t n, void *stream);

/* weight f32 [rows, K] → out f16 [rows, 2, K]: hi = f16(w) | lo' = f16((w - hi) * 2^11), the operand form the fp32 convs use on the
 * fp16 matrix pipe (weight_split above). Done once per weight: every pixel tile of a launch otherwise repeats the split. */
int aq3d_conv2d_split_weight(const float *weight, int64_t rows, int32_t K, void *out, void *stream);

/* fp32 conv / transposed conv (MFMA f32), weight f32 [Cout_total, K] with K ordered (kh, kw, c);
 * for deconv = d the weight rows are ordered (i, j, o): row = (i*d + j)*Cout + o. */
int aq3d_conv2d_f32(const float *x, const float *weight, const aq3d_conv2d_geom *g,
                    const float *bias, const float *post_scale, const float *post_shift,
                    float *y, float *pieces_out, int32_t pieces_ld, void *stream);

/* SQConv2d.forward (quant/smoothquant.py:38-99) with MORE than 8 activation bits — the reference's "16-bit activation
 * preservation" (README.md:10,21; act_bits of quant/quant_centerpoint.py:74-83, 257-258). params: the block aq3d_sq_prep wrote
 * with the same a_bits (smoothing scales, activation scale, int8 weight codes). The codes do not fit the int8 matrix pipe, so
 * the layer multiplies the fake-quantised VALUES like the reference does: x' = clamp(rint((x / s) * sx)) / sx per (pixel,
 * tap, channel) while staging, weights = the int8 codes, the per-channel weight step in the epilogue; fp32 accumulation on the
 * fp16 pipe with split operands. workspace: aq3d_conv2d_sq16_workspace_bytes. */
size_t aq3d_conv2d_sq16_workspace_bytes(int32_t Cin, int32_t Cout, int32_t ksize);
int aq3d_conv2d_sq16(const float *x, const void *params, const aq3d_conv2d_geom *g, int32_t w_bits, int32_t a_bits,
                     const float *bias, const float *post_scale, const float *post_shift, float *y, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- head decode + NMS ----------- */

typedef struct {
    int32_t H, W;              /* feature map size                           */
    int32_t num_classes;       /* classes of this head (<= 16)                */
    int32_t K;                 /* MAX_OBJ_PER_SAMPLE (<= 512)                 */
    int32_t has_vel;
    float feature_map_stride;
    float voxel_size[2];       /* x, y */
    float pc_range_min[2];     /* x, y */
    float post_center_range[6];
    float score_thresh;        /* strict >; negative disables                */
    int32_t hm_ld, reg_ld;     /* pixel strides (floats) of the heat map and of the regression maps */
} aq3d_decode_cfg;

/* Workspace of the decode calls. The returned size lets the top-K run on many workgroups (score pass + histogram select) when the map
 * (H*W pixels) or the sparse head's row capacity is at most 36 096 entries; num_classes * K * 8 + 256 bytes is the minimum either call
 * accepts (one workgroup per class then). Contents are scratch: nothing has to be initialised or preserved between calls. */
size_t aq3d_center_decode_workspace_bytes(const aq3d_decode_cfg *cfg);
size_t aq3d_voxel_decode_workspace_bytes(const aq3d_decode_cfg *cfg, int32_t cap);

/* Raw head outputs (NHWC f32, pre-sigmoid heat map; dim pre-exp) → candidate boxes sorted by score (desc,
 * ties by class-major flat index asc) with the range / score mask applied:
 *   boxes_out f32 [K, 7 or 9], scores_out f32 [K], labels_out i32 [K] (head-local class id), count_out i32 [1]. */
int aq3d_center_decode(const float *hm, const float *center, const float *center_z, const float *dim,
                       const float *rot, const float *vel, const aq3d_decode_cfg *cfg,
                       float *boxes_out, float *scores_out, int32_t *labels_out, int32_t *count_out,
                       void *workspace, size_t workspace_bytes, void *stream);

size_t aq3d_nms_workspace_bytes(int32_t cap);

/* Rotated BEV-IoU NMS over boxes already sorted by descending score.
 *   boxes f32 [*count, box_ld] (x,y,z,dx,dy,dz,heading,...); keep_out i32 [cap] indices into boxes in keep order,
 *   keep_count i32 [1] = min(#kept, post_max). cap <= 4096. */
int aq3d_nms_rotated(const float *boxes, int32_t box_ld, const int32_t *count, int32_t cap, float thresh,
                     int32_t post_max, int32_t *keep_out, int32_t *keep_count,
                     void *workspace, size_t workspace_bytes, void *stream);

/* out rows = in rows gathered by keep (boxes, scores, labels + label_offset) — final pred_dict tensors. */
int aq3d_gather_dets(const float *boxes, int32_t box_ld, const float *scores, const int32_t *labels,
                     const int32_t *class_map, const int32_t *keep, const int32_t *keep_count, int32_t cap,
                     float *boxes_out, float *scores_out, int64_t *labels_out, void *stream);

/* The whole post-processing chain of up to 8 heads in FIVE launches (top-K, decode, mask, sweep, gather) instead of five per head:
 * center_head.py:297-365 (generate_predicted_boxes: every head in turn) and voxelnext_head.py:418-560. Per head the arguments of
 * aq3d_center_decode (coords == NULL) or aq3d_voxel_decode (coords / count / cap given), of aq3d_nms_rotated on the first nms_cap
 * (<= cfg->K) candidates, and of aq3d_gather_dets; results identical to the three calls head by head. */
typedef struct aq3d_head_problem {
    const float *hm, *center, *center_z, *dim, *rot, *vel;     /* maps (dense head) or rows (sparse head) */
    const int32_t *coords, *count;                             /* sparse head only                         */
    int32_t cap;                                               /* sparse head: capacity of the rows        */
    int32_t sample;                                            /* sparse head: -1 = decode every live row; b >= 0 = only the rows of sample b
                                                                * of a batch (voxelnext_head.py:418-488 loops over the samples); the rows must
                                                                * be ordered by sample, as a raster-ordered index level's are */
    const aq3d_decode_cfg *cfg;
    float *cand_boxes, *cand_scores;                           /* [K, 7 or 9], [K]                          */
    int32_t *cand_labels, *cand_count;                         /* [K], [1]                                  */
    void *decode_workspace;
    size_t decode_workspace_bytes;                             /* aq3d_center_ / aq3d_voxel_decode_workspace_bytes */
    float nms_thresh;
    int32_t nms_cap, post_max;
    int32_t *keep, *keep_count;                                /* [nms_cap], [1]                            */
    void *nms_workspace;
    size_t nms_workspace_bytes;                                /* aq3d_nms_workspace_bytes(nms_cap)         */
    const int32_t *class_map;                                  /* head-local class -> global class, or NULL */
    float *boxes_out, *scores_out;                             /* [nms_cap, 7 or 9], [nms_cap]              */
    int64_t *labels_out;                                       /* [nms_cap]                                 */
} aq3d_head_problem;
int aq3d_decode_nms_batch(const aq3d_head_problem *heads, int32_t n, void *stream);

/* ---------------------------------------------------------------- VoxelNeXt (fully sparse) extras ---------------- */

/* VoxelResBackBone8xVoxelNeXt.bev_out (pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:149-198): union of up to 3
 * index levels projected onto the (y, x) plane — level l coordinates multiplied by scales[l] — as a 2-D level (shape (1, H, W)),
 * and the features of coinciding sites summed in a fixed (level, z) order (no float atomics).
 *   aq3d_level_merge      builds the merged level `out`;
 *   aq3d_level_merge_sum  feats_out f32 [out->cap, C] from feats[l] f32 [levels[l]->cap, C]. */
int aq3d_level_merge(const aq3d_level *const *levels, const int32_t *scales, int32_t n, const aq3d_level *out,
                     void *workspace, size_t workspace_bytes, void *stream);
int aq3d_level_merge_sum(const aq3d_level *const *levels, const float *const *feats, const int32_t *scales, int32_t n,
                         int32_t C, const aq3d_level *out, float *feats_out, void *stream);

/* centernet_utils.decode_bbox_from_voxels_nuscenes (model_utils/centernet_utils.py:243-354) for ONE sample: like
 * aq3d_center_decode, but the candidates are the *count live rows of a sparse head output (row-major maps with pixel
 * strides cfg->hm_ld / cfg->reg_ld) and their grid position comes from coords rows (b, z, y, x). cfg->H / W are ignored. */
int aq3d_voxel_decode(const float *hm, const float *center, const float *center_z, const float *dim, const float *rot,
                      const float *vel, const int32_t *coords, const int32_t *count, int32_t cap, const aq3d_decode_cfg *cfg,
                      float *boxes_out, float *scores_out, int32_t *labels_out, int32_t *count_out, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- evaluation helpers (recall vs ground truth) ---- */

/* iou_out f32 [n, m] = 3-D IoU (rotated BEV overlap x height overlap) of boxes_a [n, lda] vs boxes_b [m, ldb],
 * first 7 columns (x, y, z, dx, dy, dz, heading). */
int aq3d_boxes_iou3d(const float *boxes_a, int32_t n, int32_t lda, const float *boxes_b, int32_t m, int32_t ldb,
                     float *iou_out, void *stream);

/* counts_out i32 [n_thresh] += number of columns j of iou [n, m] whose maximum over the rows exceeds thresholds[t]. */
int aq3d_recall_counts(const float *iou, int32_t n, int32_t m, const float *thresholds, int32_t n_thresh,
                       int32_t *counts_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AQ3D_H */
