"""ctypes binding of libaq3d.so (the C ABI declared in include/aq3d.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libaq3d.so")

ABI_VERSION = 19


class Aq3dError(RuntimeError):
    pass


class VoxelCfg(Structure):
    _fields_ = [("range_min", c_float * 3), ("voxel_size", c_float * 3), ("grid", c_int32 * 3),
                ("num_features", c_int32), ("point_stride", c_int32), ("batch_col", c_int32),
                ("max_points_per_voxel", c_int32), ("max_voxels", c_int32)]


class Conv2dGeom(Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32), ("ksize", c_int32),
                ("stride", c_int32), ("pad", c_int32), ("Ho", c_int32), ("Wo", c_int32), ("in_ld", c_int32),
                ("out_ld", c_int32), ("out_coff", c_int32), ("deconv", c_int32), ("relu", c_int32)]


class DecodeCfg(Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("num_classes", c_int32), ("K", c_int32), ("has_vel", c_int32),
                ("feature_map_stride", c_float), ("voxel_size", c_float * 2), ("pc_range_min", c_float * 2),
                ("post_center_range", c_float * 6), ("score_thresh", c_float), ("hm_ld", c_int32),
                ("reg_ld", c_int32)]


class HeadProblem(Structure):
    _fields_ = [("hm", c_void_p), ("center", c_void_p), ("center_z", c_void_p), ("dim", c_void_p), ("rot", c_void_p), ("vel", c_void_p),
                ("coords", c_void_p), ("count", c_void_p), ("cap", c_int32), ("sample", c_int32), ("cfg", POINTER(DecodeCfg)),
                ("cand_boxes", c_void_p), ("cand_scores", c_void_p), ("cand_labels", c_void_p), ("cand_count", c_void_p),
                ("decode_workspace", c_void_p), ("decode_workspace_bytes", c_size_t), ("nms_thresh", c_float), ("nms_cap", c_int32),
                ("post_max", c_int32), ("keep", c_void_p), ("keep_count", c_void_p), ("nms_workspace", c_void_p),
                ("nms_workspace_bytes", c_size_t), ("class_map", c_void_p), ("boxes_out", c_void_p), ("scores_out", c_void_p),
                ("labels_out", c_void_p)]


class Conv2dProblem(Structure):
    _fields_ = [("x", c_void_p), ("weight_or_params", c_void_p), ("geom", POINTER(Conv2dGeom)), ("bias", c_void_p),
                ("post_scale", c_void_p), ("post_shift", c_void_p), ("y", c_void_p), ("pieces_out", c_void_p),
                ("pieces_ld", c_int32), ("pieces_mode", c_int32), ("weight_split", c_void_p)]


class SpconvProblem(Structure):
    _fields_ = [("feats_in", c_void_p), ("act_dequant", c_void_p), ("weight", c_void_p), ("w_dequant", c_void_p),
                ("bias", c_void_p), ("post_scale", c_void_p), ("post_shift", c_void_p), ("residual", c_void_p),
                ("nbr", c_void_p), ("count_out", c_void_p), ("feats_out", c_void_p), ("absmax_out", c_void_p), ("cout_real", c_int32)]


class Level(Structure):
    _fields_ = [("shape", c_int32 * 3), ("batch", c_int32), ("cap", c_int32), ("flags", c_int32), ("bits", c_void_p),
                ("base", c_void_p), ("row_of_rank", c_void_p), ("coords", c_void_p), ("count", c_void_p), ("total", c_void_p)]


class NbrJob(Structure):
    _fields_ = [("level_in", POINTER(Level)), ("ksize", c_int32 * 3), ("stride", c_int32 * 3), ("pad", c_int32 * 3),
                ("nbr", c_void_p)]


class SqPrepProblem(Structure):
    _fields_ = [("pieces", c_void_p), ("weight", c_void_p), ("w_colmax", c_void_p), ("geom", POINTER(Conv2dGeom)),
                ("alpha", c_float), ("static_amax_x", c_float), ("static_amax_w", c_void_p), ("amax_x_acc", c_void_p),
                ("params", c_void_p)]


I3 = c_int32 * 3
P = c_void_p

# name -> (restype, argtypes); mirrors include/aq3d.h one to one.
SIGNATURES = {
    "aq3d_last_error_string": (c_char_p, []),
    "aq3d_abi_version": (c_int32, []),
    "aq3d_voxelize_workspace_bytes": (c_size_t, [c_int32]),
    "aq3d_voxelize": (c_int32, [P, c_int32, POINTER(VoxelCfg), c_int32, c_int32, P, P, P, P, c_int32, P, c_size_t, P]),
    "aq3d_mean_vfe": (c_int32, [P, P, c_int32, c_int32, c_int32, P, P, P]),
    "aq3d_level_words": (c_int64, [I3, c_int32]),
    "aq3d_index_workspace_bytes": (c_size_t, [c_int32, c_int64, c_int32]),
    "aq3d_level_from_points": (c_int32, [P, c_int32, POINTER(VoxelCfg), c_int32, POINTER(Level), P, P, P, c_size_t, P]),
    "aq3d_level_from_coords": (c_int32, [P, c_int32, POINTER(Level), P, P, c_size_t, P]),
    "aq3d_level_downsample": (c_int32, [POINTER(Level), I3, I3, I3, POINTER(Level), P, c_size_t, P]),
    "aq3d_level_neighbors": (c_int32, [POINTER(Level), POINTER(NbrJob), c_int32, P]),
    "aq3d_level_merge": (c_int32, [P, P, c_int32, POINTER(Level), P, c_size_t, P]),
    "aq3d_level_merge_sum": (c_int32, [P, P, P, c_int32, c_int32, POINTER(Level), P, P]),
    "aq3d_absmax_cols": (c_int32, [P, P, c_int32, c_int32, P, P]),
    "aq3d_quantize_rows": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_int32, P, P, P, P]),
    "aq3d_spconv": (c_int32, [P, c_int32, P, P, c_int32, P, P, P, P, P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32,
                              c_int32, P, P, P, P, c_int32, P]),
    "aq3d_spconv_coded_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "aq3d_spconv_wprep_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "aq3d_spconv_prep": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_int32, P, P, c_int32, c_int32, P, P, P, P]),
    "aq3d_spconv_coded": (c_int32, [P, c_int32, P, P, P, P, P, P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32, P, P, P, P,
                                    c_int32, P]),
    "aq3d_scatter_dense": (c_int32, [P, P, P, c_int32, c_int32, I3, c_int32, P, P]),
    "aq3d_piece_absmax": (c_int32, [P, c_int32, c_int32, c_int32, c_int32, P, P]),
    "aq3d_sq_param_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "aq3d_sq_prep": (c_int32, [P, P, P, POINTER(Conv2dGeom), c_float, c_int32, c_int32, c_float, P, P, P, P]),
    "aq3d_spconv_group": (c_int32, [POINTER(SpconvProblem), c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, P]),
    "aq3d_sq_prep_group": (c_int32, [POINTER(SqPrepProblem), c_int32, c_int32, c_int32, P]),
    "aq3d_conv2d_i8_group": (c_int32, [POINTER(Conv2dProblem), c_int32, P]),
    "aq3d_conv2d_f32_group": (c_int32, [POINTER(Conv2dProblem), c_int32, P]),
    "aq3d_conv2d_split_weight": (c_int32, [P, ctypes.c_int64, c_int32, P, P]),
    "aq3d_conv2d_i8": (c_int32, [P, P, POINTER(Conv2dGeom), P, P, P, P, P, c_int32, c_int32, P]),
    "aq3d_conv2d_f32": (c_int32, [P, P, POINTER(Conv2dGeom), P, P, P, P, P, c_int32, P]),
    "aq3d_conv2d_sq16_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "aq3d_conv2d_sq16": (c_int32, [P, P, POINTER(Conv2dGeom), c_int32, c_int32, P, P, P, P, P, c_size_t, P]),
    "aq3d_center_decode_workspace_bytes": (c_size_t, [POINTER(DecodeCfg)]),
    "aq3d_voxel_decode_workspace_bytes": (c_size_t, [POINTER(DecodeCfg), c_int32]),
    "aq3d_center_decode": (c_int32, [P, P, P, P, P, P, POINTER(DecodeCfg), P, P, P, P, P, c_size_t, P]),
    "aq3d_nms_workspace_bytes": (c_size_t, [c_int32]),
    "aq3d_nms_rotated": (c_int32, [P, c_int32, P, c_int32, c_float, c_int32, P, P, P, c_size_t, P]),
    "aq3d_gather_dets": (c_int32, [P, c_int32, P, P, P, P, P, c_int32, P, P, P, P]),
    "aq3d_decode_nms_batch": (c_int32, [POINTER(HeadProblem), c_int32, P]),
    "aq3d_voxel_decode": (c_int32, [P, P, P, P, P, P, P, P, c_int32, POINTER(DecodeCfg), P, P, P, P, P, c_size_t, P]),
    "aq3d_boxes_iou3d": (c_int32, [P, c_int32, c_int32, P, c_int32, c_int32, P, P]),
    "aq3d_recall_counts": (c_int32, [P, c_int32, c_int32, P, c_int32, P, P]),
}

_lib = None


def load(path=None):
    """Load libaq3d.so and bind every symbol of include/aq3d.h. Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.isfile(path):
        raise Aq3dError("libaq3d.so not found at %s — build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    # libaq3d.so needs libamdhip64.so.7. PyTorch-ROCm ships its own copy of that runtime; the device pointers and stream
    # handles this host passes in belong to THAT instance, so it has to be in the process first — loading libaq3d.so
    # before torch pulls in /opt/rocm's runtime as a second instance, and launches on torch's streams then fail with
    # "no ROCm-capable device" (seen when build() ran before the first `import torch`).
    try:
        import torch  # noqa: F401
    except ImportError:     # a host without torch drives the C ABI with its own HIP runtime
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise Aq3dError("libaq3d.so at %s does not export %s" % (path, name))
        fn.restype = res
        fn.argtypes = args
    if lib.aq3d_abi_version() != ABI_VERSION:
        raise Aq3dError("libaq3d.so ABI %d != binding ABI %d" % (lib.aq3d_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().aq3d_last_error_string()
        raise Aq3dError("%s failed (status %d): %s" % (what or "aq3d call", rc, msg.decode() if msg else ""))
