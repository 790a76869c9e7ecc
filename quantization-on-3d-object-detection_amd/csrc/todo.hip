// Entry points declared in include/aq3d.h whose kernels are not written yet: they fail loudly.
#include "common.h"
#define AQ3D_TODO(name) aq3d_set_error(name ": not implemented yet"); return AQ3D_ERR_UNSUPPORTED
extern "C" {
int aq3d_scatter_dense(const float *, const int32_t *, const int32_t *, int32_t, int32_t, const int32_t[3], int32_t, float *, void *) { AQ3D_TODO("aq3d_scatter_dense"); }
int aq3d_piece_absmax(const float *, int32_t, int32_t, int32_t, int32_t, float *, void *) { AQ3D_TODO("aq3d_piece_absmax"); }
size_t aq3d_sq_param_bytes(int32_t, int32_t, int32_t) { return 0; }
int aq3d_sq_prep(const float *, const float *, const float *, const aq3d_conv2d_geom *, float, int32_t, int32_t, float, const float *, void *, void *) { AQ3D_TODO("aq3d_sq_prep"); }
int aq3d_conv2d_i8(const float *, const void *, const aq3d_conv2d_geom *, const float *, const float *, const float *, float *, void *) { AQ3D_TODO("aq3d_conv2d_i8"); }
int aq3d_conv2d_f32(const float *, const float *, const aq3d_conv2d_geom *, const float *, const float *, const float *, float *, void *) { AQ3D_TODO("aq3d_conv2d_f32"); }
size_t aq3d_center_decode_workspace_bytes(const aq3d_decode_cfg *) { return 0; }
int aq3d_center_decode(const float *, const float *, const float *, const float *, const float *, const float *, const aq3d_decode_cfg *, float *, float *, int32_t *, int32_t *, void *, size_t, void *) { AQ3D_TODO("aq3d_center_decode"); }
size_t aq3d_nms_workspace_bytes(int32_t) { return 0; }
int aq3d_nms_rotated(const float *, int32_t, const int32_t *, int32_t, float, int32_t, int32_t *, int32_t *, void *, size_t, void *) { AQ3D_TODO("aq3d_nms_rotated"); }
int aq3d_gather_dets(const float *, int32_t, const float *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, int32_t, float *, float *, int64_t *, void *) { AQ3D_TODO("aq3d_gather_dets"); }
}
