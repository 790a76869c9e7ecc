"""Parameter containers of the CenterPoint detector, named exactly like the reference's modules so that
pcdet checkpoints load key-for-key and quantization exclusion lists address the same paths:

    vfe                   MeanVFE                  pcdet/models/backbones_3d/vfe/mean_vfe.py:6-31
    backbone_3d           VoxelResBackBone8x       pcdet/models/backbones_3d/spconv_backbone.py:184-295
    map_to_bev_module     HeightCompression        pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-26
    backbone_2d           BaseBEVBackbone          pcdet/models/backbones_2d/base_bev_backbone.py:6-112
    dense_head            CenterHead               pcdet/models/dense_heads/center_head.py:12-100

These classes hold parameters, hyper-parameters and topology only. They have no torch compute path: the
detector compiles them into a device plan (engine.py) that runs the HIP kernels. Calling one directly raises.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _ntuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class SparseModule(nn.Module):
    """Marker base class (spconv.pytorch.modules.SparseModule)."""

    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container; run the detector (HIP engine), there is no torch path"
                           % type(self).__name__)


class SparseConvolution(SparseModule):
    """Sparse conv parameters, KRSC weight layout [Cout, kz, ky, kx, Cin] as spconv 2.x stores it."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None,
                 subm=False, ndim=3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.ndim = ndim
        self.kernel_size, self.stride, self.padding = _ntuple(kernel_size, ndim), _ntuple(stride, ndim), _ntuple(padding, ndim)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        fan_in = in_channels * int(np.prod(self.kernel_size))
        nn.init.normal_(self.weight, 0.0, float(np.sqrt(2.0 / fan_in)))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=True)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=False)


class SubMConv2d(SparseConvolution):
    """2-D submanifold conv (VoxelNeXt tail / head), weight [Cout, ky, kx, Cin]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=True, ndim=2)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=False, ndim=2)


class SparseSequential(nn.Sequential, SparseModule):
    def forward(self, *a, **k):
        return SparseModule.forward(self)


class SparseBasicBlock(SparseModule):
    """spconv_backbone.py:30-67: conv1-bn1-relu-conv2-bn2-(+identity)-relu, both convs SubM 3x3x3 with bias."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, bias=None, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        if bias is None:
            bias = norm_fn is not None
        self.conv1 = SubMConv3d(inplanes, planes, 3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                   norm_fn=None):
    if conv_type == "subm":
        conv = SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                            indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, *a, **k):
        raise RuntimeError("MeanVFE is fused into the voxelizer kernel; run the detector")


class VoxelResBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        use_bias = self.model_cfg.get("USE_BIAS", None)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + np.array([1, 0, 0]))]
        self.conv_input = SparseSequential(
            SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key="subm1"), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = SparseSequential(
            SparseBasicBlock(16, 16, bias=use_bias, norm_fn=norm_fn, indice_key="res1"),
            SparseBasicBlock(16, 16, bias=use_bias, norm_fn=norm_fn, indice_key="res1"))
        self.conv2 = SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            SparseBasicBlock(32, 32, bias=use_bias, norm_fn=norm_fn, indice_key="res2"),
            SparseBasicBlock(32, 32, bias=use_bias, norm_fn=norm_fn, indice_key="res2"))
        self.conv3 = SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            SparseBasicBlock(64, 64, bias=use_bias, norm_fn=norm_fn, indice_key="res3"),
            SparseBasicBlock(64, 64, bias=use_bias, norm_fn=norm_fn, indice_key="res3"))
        self.conv4 = SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            SparseBasicBlock(128, 128, bias=use_bias, norm_fn=norm_fn, indice_key="res4"),
            SparseBasicBlock(128, 128, bias=use_bias, norm_fn=norm_fn, indice_key="res4"))
        last_pad = self.model_cfg.get("last_pad", 0)
        self.conv_out = SparseSequential(
            SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                         indice_key="spconv_down2"), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}

    def forward(self, *a, **k):
        raise RuntimeError("VoxelResBackBone8x is a parameter container; run the detector")


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    def forward(self, *a, **k):
        raise RuntimeError("HeightCompression is a parameter container; run the detector")


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = self.model_cfg.get("LAYER_NUMS", None) or []
        layer_strides = self.model_cfg.get("LAYER_STRIDES", None) or []
        num_filters = self.model_cfg.get("NUM_FILTERS", None) or []
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        upsample_strides = self.model_cfg.get("UPSAMPLE_STRIDES", None) or []
        num_upsample_filters = self.model_cfg.get("NUM_UPSAMPLE_FILTERS", None) or []
        assert len(upsample_strides) == len(num_upsample_filters)
        num_levels = len(layer_nums)
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(num_levels):
            cur_layers = [
                nn.ZeroPad2d(1),
                nn.Conv2d(c_in_list[idx], num_filters[idx], kernel_size=3, stride=layer_strides[idx], padding=0, bias=False),
                nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01),
                nn.ReLU()]
            for _ in range(layer_nums[idx]):
                cur_layers.extend([
                    nn.Conv2d(num_filters[idx], num_filters[idx], kernel_size=3, padding=1, bias=False),
                    nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01),
                    nn.ReLU()])
            self.blocks.append(nn.Sequential(*cur_layers))
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride > 1 or (stride == 1 and not self.model_cfg.get("USE_CONV_FOR_NO_STRIDE", False)):
                    self.deblocks.append(nn.Sequential(
                        nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], upsample_strides[idx],
                                           stride=upsample_strides[idx], bias=False),
                        nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01),
                        nn.ReLU()))
                else:
                    stride = int(np.round(1 / stride))
                    self.deblocks.append(nn.Sequential(
                        nn.Conv2d(num_filters[idx], num_upsample_filters[idx], stride, stride=stride, bias=False),
                        nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01),
                        nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > num_levels:
            raise NotImplementedError("extra trailing deblock is not part of the CenterPoint configs")
        self.num_bev_features = c_in

    def forward(self, *a, **k):
        raise RuntimeError("BaseBEVBackbone is a parameter container; run the detector")


class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False, norm_func=None):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in self.sep_head_dict:
            output_channels = self.sep_head_dict[cur_name]["out_channels"]
            num_conv = self.sep_head_dict[cur_name]["num_conv"]
            fc_list = []
            for _ in range(num_conv - 1):
                fc_list.append(nn.Sequential(
                    nn.Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                    nn.BatchNorm2d(input_channels) if norm_func is None else norm_func(input_channels),
                    nn.ReLU()))
            fc_list.append(nn.Conv2d(input_channels, output_channels, kernel_size=3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*fc_list)
            if "hm" in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight.data)
                        if hasattr(m, "bias") and m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(cur_name, fc)

    def forward(self, *a, **k):
        raise RuntimeError("SeparateHead is a parameter container; run the detector")


class CenterHead(nn.Module):
    """Inference half of pcdet/models/dense_heads/center_head.py (no target assignment / losses)."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get("FEATURE_MAP_STRIDE", None)
        self.class_names = class_names
        self.class_names_each_head = []
        self.class_id_mapping_each_head = []
        for cur_class_names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur_class_names if x in class_names])
            self.class_id_mapping_each_head.append(
                torch.tensor([self.class_names.index(x) for x in cur_class_names if x in class_names], dtype=torch.int32))
        total_classes = sum(len(x) for x in self.class_names_each_head)
        assert total_classes == len(self.class_names), "class_names_each_head=%s" % self.class_names_each_head
        norm_func = partial(nn.BatchNorm2d, eps=self.model_cfg.get("BN_EPS", 1e-5), momentum=self.model_cfg.get("BN_MOM", 0.1))
        self.shared_conv = nn.Sequential(
            nn.Conv2d(input_channels, self.model_cfg.SHARED_CONV_CHANNEL, 3, stride=1, padding=1,
                      bias=self.model_cfg.get("USE_BIAS_BEFORE_NORM", False)),
            norm_func(self.model_cfg.SHARED_CONV_CHANNEL), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for cur_class_names in self.class_names_each_head:
            cur_head_dict = {k: dict(v) for k, v in self.separate_head_cfg.HEAD_DICT.items()}
            cur_head_dict["hm"] = dict(out_channels=len(cur_class_names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(
                input_channels=self.model_cfg.SHARED_CONV_CHANNEL, sep_head_dict=cur_head_dict, init_bias=-2.19,
                use_bias=self.model_cfg.get("USE_BIAS_BEFORE_NORM", False), norm_func=norm_func))
        self.predict_boxes_when_training = predict_boxes_when_training

    def forward(self, *a, **k):
        raise RuntimeError("CenterHead is a parameter container; run the detector")


class VoxelResBackBone8xVoxelNeXt(nn.Module):
    """pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:69-147 — six 3-D stages, 2-D tail (conv_out, shared_conv)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        ks = model_cfg.get("SPCONV_KERNEL_SIZES", [3, 3, 3, 3])
        ch = model_cfg.get("CHANNELS", [16, 32, 64, 128, 128])
        out_channel = model_cfg.get("OUT_CHANNEL", 128)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + np.array([1, 0, 0]))]
        self.conv_input = SparseSequential(
            SubMConv3d(input_channels, ch[0], 3, padding=1, bias=False, indice_key="subm1"), norm_fn(ch[0]), nn.ReLU())
        block = post_act_block
        sbb = lambda c, key: SparseBasicBlock(c, c, bias=True, norm_fn=norm_fn, indice_key=key)
        self.conv1 = SparseSequential(sbb(ch[0], "res1"), sbb(ch[0], "res1"))
        self.conv2 = SparseSequential(
            block(ch[0], ch[1], ks[0], norm_fn=norm_fn, stride=2, padding=int(ks[0] // 2), indice_key="spconv2", conv_type="spconv"),
            sbb(ch[1], "res2"), sbb(ch[1], "res2"))
        self.conv3 = SparseSequential(
            block(ch[1], ch[2], ks[1], norm_fn=norm_fn, stride=2, padding=int(ks[1] // 2), indice_key="spconv3", conv_type="spconv"),
            sbb(ch[2], "res3"), sbb(ch[2], "res3"))
        self.conv4 = SparseSequential(
            block(ch[2], ch[3], ks[2], norm_fn=norm_fn, stride=2, padding=int(ks[2] // 2), indice_key="spconv4", conv_type="spconv"),
            sbb(ch[3], "res4"), sbb(ch[3], "res4"))
        self.conv5 = SparseSequential(
            block(ch[3], ch[4], ks[3], norm_fn=norm_fn, stride=2, padding=int(ks[3] // 2), indice_key="spconv5", conv_type="spconv"),
            sbb(ch[4], "res5"), sbb(ch[4], "res5"))
        self.conv6 = SparseSequential(
            block(ch[4], ch[4], ks[3], norm_fn=norm_fn, stride=2, padding=int(ks[3] // 2), indice_key="spconv6", conv_type="spconv"),
            sbb(ch[4], "res6"), sbb(ch[4], "res6"))
        self.conv_out = SparseSequential(
            SparseConv2d(ch[3], out_channel, 3, stride=1, padding=1, bias=False, indice_key="spconv_down2"),
            norm_fn(out_channel), nn.ReLU())
        self.shared_conv = SparseSequential(
            SubMConv2d(out_channel, out_channel, 3, stride=1, padding=1, bias=True), nn.BatchNorm1d(out_channel), nn.ReLU(True))
        self.num_point_features = out_channel
        self.backbone_channels = {"x_conv1": ch[0], "x_conv2": ch[1], "x_conv3": ch[2], "x_conv4": ch[3]}

    def forward(self, *a, **k):
        raise RuntimeError("VoxelResBackBone8xVoxelNeXt is a parameter container; run the detector")


class SparseSeparateHead(nn.Module):
    """pcdet/models/dense_heads/voxelnext_head.py:13-47 — per-branch SubMConv2d stacks on the sparse BEV tensor."""

    def __init__(self, input_channels, sep_head_dict, kernel_size, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in self.sep_head_dict:
            output_channels = self.sep_head_dict[cur_name]["out_channels"]
            num_conv = self.sep_head_dict[cur_name]["num_conv"]
            fc_list = []
            for _ in range(num_conv - 1):
                fc_list.append(SparseSequential(
                    SubMConv2d(input_channels, input_channels, kernel_size, padding=int(kernel_size // 2), bias=use_bias,
                               indice_key=cur_name),
                    nn.BatchNorm1d(input_channels), nn.ReLU()))
            fc_list.append(SubMConv2d(input_channels, output_channels, 1, bias=True, indice_key=cur_name + "out"))
            fc = nn.Sequential(*fc_list)
            if "hm" in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            self.__setattr__(cur_name, fc)

    def forward(self, *a, **k):
        raise RuntimeError("SparseSeparateHead is a parameter container; run the detector")


class VoxelNeXtHead(nn.Module):
    """Inference half of pcdet/models/dense_heads/voxelnext_head.py:50-110,418-559 (no IoU branch / double flip)."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=False):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get("FEATURE_MAP_STRIDE", None)
        self.class_names = class_names
        assert not self.model_cfg.get("IOU_BRANCH", False), "IoU branch (Waymo VoxelNeXt) is outside the nuScenes config"
        assert not self.model_cfg.get("DOUBLE_FLIP", False), "double-flip test-time augmentation is not on the path"
        self.class_names_each_head, self.class_id_mapping_each_head = [], []
        for cur_class_names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur_class_names if x in class_names])
            self.class_id_mapping_each_head.append(
                torch.tensor([self.class_names.index(x) for x in cur_class_names if x in class_names], dtype=torch.int32))
        assert sum(len(x) for x in self.class_names_each_head) == len(self.class_names)
        kernel_size_head = self.model_cfg.get("KERNEL_SIZE_HEAD", 3)
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for cur_class_names in self.class_names_each_head:
            cur_head_dict = {k: dict(v) for k, v in self.separate_head_cfg.HEAD_DICT.items()}
            cur_head_dict["hm"] = dict(out_channels=len(cur_class_names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SparseSeparateHead(
                input_channels=self.model_cfg.get("SHARED_CONV_CHANNEL", 128), sep_head_dict=cur_head_dict,
                kernel_size=kernel_size_head, init_bias=-2.19, use_bias=self.model_cfg.get("USE_BIAS_BEFORE_NORM", False)))
        self.predict_boxes_when_training = predict_boxes_when_training

    def forward(self, *a, **k):
        raise RuntimeError("VoxelNeXtHead is a parameter container; run the detector")


def find_all_spconv_keys(model, prefix=""):
    """pcdet/utils/spconv_utils.py:15-29 — state-dict keys of sparse-conv weights (layout adaptor on load)."""
    found = set()
    for name, child in model.named_children():
        new_prefix = "%s.%s" % (prefix, name) if prefix != "" else name
        if isinstance(child, SparseConvolution):
            found.add(new_prefix + ".weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found
