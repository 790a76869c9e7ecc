"""Post-training-quantization surgery with the reference's API surface:

    quant(model, w_bits, act_bits, sq, alpha, static, test_loader, n_batches)   quant/quant_centerpoint.py:74-131
    q_conv3d / smoothquant / pytorch_quant / collect_stats / compute_amax       quant/quantize.py:13-207
    QConvNd                                                                      quant/quant.py:6-58
    SQConv2d                                                                     quant/smoothquant.py:6-99
    backbone_no_list / no_list                                                   quant/quant_centerpoint.py:24-71

In the reference these wrappers ARE the compute path (fake-quant in fp32 around spconv / matmul). Here they are
descriptors: they record bit widths, smoothing strength, calibration state and exclusion decisions module by
module, and the detector's engine compiles them into int8 device kernels (weights are quantised once at plan
time instead of on every forward, cf. quant/quant.py:42-48,56).
"""
import numpy as np
import torch
import torch.nn as nn

from .modules import SparseConv3d, SparseConvolution, SparseModule, SubMConv3d

_EPOCH = [0]   # bumped by every surgery / calibration-state change; engines rebuild when it moves


def surgery_epoch():
    return _EPOCH[0]


def _touch():
    _EPOCH[0] += 1


class TensorQuantizer(nn.Module):
    """Descriptor + calibration state of one fake-quantizer (pytorch_quantization.nn.TensorQuantizer as used by the
    reference: signed, max calibrator, `axis` = kept dimension or None for per-tensor)."""

    def __init__(self, num_bits=8, axis=None):
        super().__init__()
        self.num_bits = int(num_bits)
        self.axis = axis
        self._amax = None          # torch tensor once calibrated (static mode)
        self._calib_amax = None    # running max while calibrating
        self._if_calib = False
        self._if_quant = True

    # -- pytorch_quantization-style switches (quant/quantize.py:177-194)
    def enable_calib(self):
        self._if_calib = True
        _touch()

    def disable_calib(self):
        self._if_calib = False
        _touch()

    def enable_quant(self):
        self._if_quant = True
        _touch()

    def disable_quant(self):
        self._if_quant = False
        _touch()

    def collect(self, amax):
        """MaxCalibrator.collect: running element-wise max."""
        amax = amax.detach().float().cpu().reshape(-1)
        self._calib_amax = amax if self._calib_amax is None else torch.maximum(self._calib_amax, amax)

    def load_calib_amax(self, *args, **kwargs):
        if self._calib_amax is not None:
            self._amax = self._calib_amax.clone()
        _touch()

    @property
    def amax(self):
        return self._amax

    def extra_repr(self):
        return "%d bit, axis=%s, %s" % (self.num_bits, self.axis, "calibrated" if self._amax is not None else "dynamic")

    def forward(self, *a, **k):
        raise RuntimeError("TensorQuantizer is a descriptor; quantisation runs inside the HIP kernels")


class QConvNd(SparseModule):
    """Channel-wise quant descriptor for SubMConv3d / SparseConv3d (quant/quant.py:6-58):
    weight per out-channel (axis 0 of the [oc, ic*k^3] view), activations per in-channel (cw) or per tensor."""

    def __init__(self, module, w_bits, act_bits, cw):
        super().__init__()
        self.module = module
        self.w_bits, self.act_bits, self.cw = int(w_bits), int(act_bits), bool(cw)
        self.w_quant = TensorQuantizer(w_bits, axis=0)
        self.act_quant = TensorQuantizer(act_bits, axis=1 if cw else None)


class QConv2d(SparseModule):
    """Descriptor of quant/quant_voxelnext.py:138-169 for SubMConv2d / SparseConv2d. As committed, the reference computes
    the quantised activations but never writes them back (quant_voxelnext.py:165-167), so the layer is effectively
    WEIGHT-ONLY int8 (per out-channel) on fp32 activations — that is what is reproduced."""

    def __init__(self, module, w_bits, act_bits):
        super().__init__()
        self.module = module
        self.w_bits, self.act_bits = int(w_bits), int(act_bits)
        self.w_quant = TensorQuantizer(w_bits, axis=0)
        self.act_quant = TensorQuantizer(act_bits, axis=None)     # present (calibration addresses it), unused at inference


class _QuantConv2dBase(nn.Module):
    def __init__(self, conv, w_bits, act_bits):
        super().__init__()
        # share the Parameters of the wrapped nn.Conv2d (the reference copies vars(), quant/quantize.py:59-63)
        self.in_channels, self.out_channels = conv.in_channels, conv.out_channels
        self.kernel_size = conv.kernel_size[0] if isinstance(conv.kernel_size, tuple) else conv.kernel_size
        self.stride = conv.stride[0] if isinstance(conv.stride, tuple) else conv.stride
        self.padding = conv.padding[0] if isinstance(conv.padding, tuple) else conv.padding
        self.dilation = conv.dilation[0] if isinstance(conv.dilation, tuple) else conv.dilation
        self.weight = conv.weight
        if conv.bias is not None:
            self.bias = conv.bias
        else:
            self.register_parameter("bias", None)
        self._weight_quantizer = TensorQuantizer(w_bits, axis=0)
        self._input_quantizer = TensorQuantizer(act_bits, axis=None)

    def forward(self, *a, **k):
        raise RuntimeError("%s is a descriptor; run the detector" % type(self).__name__)


class SQConv2d(_QuantConv2dBase):
    """SmoothQuant conv descriptor (quant/smoothquant.py:6-99): s = act^alpha / w^(1-alpha) per im2col column."""

    def __init__(self, conv, scaling_factor, w_bits, act_bits):
        super().__init__(conv, w_bits, act_bits)
        self.scaling_factor = scaling_factor


class QuantConv2d(_QuantConv2dBase):
    """pytorch_quantization.nn.Conv2d equivalent (quant/quantize.py:120-170): per-tensor input, per-oc weight."""
    scaling_factor = None


# ================================ surgery (names and arguments as in quant/quantize.py) ================
def q_conv3d(model, module_dict, curr_path, w_bits, act_bits, cw, src, no_list):
    for name, module in model.named_children():
        path = "%s.%s" % (curr_path, name) if curr_path else name
        q_conv3d(module, module_dict, path, w_bits, act_bits, cw, src, no_list)
        if isinstance(module, src) and path not in no_list:
            model._modules[name] = QConvNd(module=module, w_bits=w_bits, act_bits=act_bits, cw=cw)
    _touch()


def smoothquant_layer(nn_instance, quant_module, scaling_factor, w_bits, act_bits):
    if scaling_factor is None:
        raise ValueError("Please specify the scaling_factor parameter!")
    if act_bits is None or w_bits is None:
        raise ValueError("Please specify the num_bits parameter!")
    return quant_module(nn_instance, scaling_factor, w_bits, act_bits)


def smoothquant(model, module_dict, curr_path, alpha, w_bits, act_bits, src, tgt, no_list):
    for name, module in model.named_children():
        path = "%s.%s" % (curr_path, name) if curr_path else name
        smoothquant(module, module_dict, path, alpha, w_bits, act_bits, src, tgt, no_list)
        if isinstance(module, src) and not isinstance(module, _QuantConv2dBase) and path not in no_list:
            model._modules[name] = smoothquant_layer(module, tgt, alpha, w_bits, act_bits)
    _touch()


def pytorch_quant_layer(nn_instance, quant_module, w_bits, act_bits):
    return quant_module(nn_instance, w_bits, act_bits)


def pytorch_quant(model, module_dict, curr_path, w_bits, act_bits, src, tgt):
    for name, module in model.named_children():
        path = "%s.%s" % (curr_path, name) if curr_path else name
        pytorch_quant(module, module_dict, path, w_bits, act_bits, src, tgt)
        if isinstance(module, src) and not isinstance(module, _QuantConv2dBase):
            model._modules[name] = pytorch_quant_layer(module, tgt, w_bits, act_bits)
    _touch()


def collect_stats(model, data_loader, n_batches=200):
    """quant/quantize.py:175-195 — quantizers to calibration mode, n_batches+2 forwards, back to quant mode."""
    from .datasets import load_data_to_gpu
    model.eval()
    for name, module in model.named_modules():
        if name.endswith("_quantizer") or name.endswith("_quant"):
            module.enable_calib()
            module.disable_quant()
    with torch.no_grad():
        for i, batch_dict in enumerate(data_loader):
            load_data_to_gpu(batch_dict)
            model(batch_dict)
            if i > n_batches:
                break
    for name, module in model.named_modules():
        if name.endswith("_quantizer") or name.endswith("_quant"):
            module.disable_calib()
            module.enable_quant()


def compute_amax(model, device=None, **kwargs):
    """quant/quantize.py:198-207 — every quantizer has a max calibrator here, so `method` is ignored."""
    for _, module in model.named_modules():
        if isinstance(module, TensorQuantizer):
            module.load_calib_amax(**kwargs)


# ================================ the north-star recipe (quant/quant_centerpoint.py:24-131) ============
backbone_no_list = ["backbone_3d.conv_input.0"]

no_list = ["dense_head.heads_list.%d.%s" % (h, b) for h in range(6)
           for b in ("center.1", "center_z.1", "dim.1", "rot.1", "vel.1", "hm.0.0", "hm.1")]


def quant(model, w_bits, act_bits, sq, alpha, static, test_loader=None, n_batches=200):
    if sq:
        q_conv3d(model, module_dict={}, curr_path="", w_bits=w_bits, act_bits=act_bits, cw=sq,
                 src=(SubMConv3d, SparseConv3d), no_list=backbone_no_list)
        smoothquant(model, module_dict={}, curr_path="", alpha=alpha, w_bits=w_bits, act_bits=act_bits,
                    src=(nn.Conv2d,), tgt=SQConv2d, no_list=no_list)
    else:
        q_conv3d(model, module_dict={}, curr_path="", w_bits=w_bits, act_bits=act_bits, cw=sq,
                 src=(SubMConv3d, SparseConv3d), no_list=[])
        pytorch_quant(model, module_dict={}, curr_path="", w_bits=w_bits, act_bits=act_bits, src=(nn.Conv2d,),
                      tgt=QuantConv2d)
    if static:
        collect_stats(model, test_loader, n_batches)
        compute_amax(model, None, method="entropy")


# ================================ the VoxelNeXt recipe (quant/quant_voxelnext.py:170-353) ==============
def q_conv3d_voxelnext(model, module_dict, curr_path, w_bits, act_bits, cw):
    """quant_voxelnext.py:170-176: every 3-D sparse conv except backbone_3d.conv_input.0."""
    q_conv3d(model, module_dict, curr_path, w_bits, act_bits, cw, (SubMConv3d, SparseConv3d), ["backbone_3d.conv_input.0"])


def q_conv2d(model, module_dict, curr_path="", w_bits=None, act_bits=None):
    """quant_voxelnext.py:281-292: every 2-D sparse conv whose path is not in the head exclusion list."""
    from .modules import SparseConv2d, SubMConv2d
    if act_bits is None or w_bits is None:
        raise ValueError("Please specify the num_bits parameter!")
    for name, module in model.named_children():
        path = "%s.%s" % (curr_path, name) if curr_path else name
        q_conv2d(module, module_dict, path, w_bits, act_bits)
        if isinstance(module, (SubMConv2d, SparseConv2d)) and path not in no_list:
            model._modules[name] = QConv2d(module=module, w_bits=w_bits, act_bits=act_bits)
    _touch()


def dynamic_quant_voxelnext(model, w_bits, act_bits, sq=False, alpha=None):
    """quant_voxelnext.py:330-337. The committed main passes sq=True, whose 2-D branch raises NameError in the reference
    (SQSubM2d is never imported); the runnable semantics are QConv3d(cw) + QConv2d, selected here for either value of sq."""
    q_conv3d_voxelnext(model, {}, "", w_bits, act_bits, cw=True if sq else False)
    q_conv2d(model, {}, "", w_bits, act_bits)


# ================================ plan-time weight quantisation =======================================
def quantize_weight_rows(w2d, bits):
    """TensorQuantizer(axis=0) on a [oc, K] fp32 matrix with IEEE numpy ops → (codes int8/int16 array, dequant f32 [oc])."""
    w = np.ascontiguousarray(w2d, dtype=np.float32)
    qmax = np.float32(2.0 ** (bits - 1) - 1.0)
    amax = np.abs(w).max(axis=1, keepdims=True).astype(np.float32)
    eps = np.float32(1.0 / (1 << 24))
    with np.errstate(divide="ignore"):
        scale = (qmax / amax).astype(np.float32)
    zero = amax <= eps
    mul = np.where(zero, np.float32(0), scale).astype(np.float32)
    div = np.where(zero, np.float32(1), scale).astype(np.float32)
    codes = np.clip(np.rint(w * mul), -qmax - 1, qmax)
    return codes, (np.float32(1.0) / div).reshape(-1).astype(np.float32), amax.reshape(-1)


# ================================ persisted calibration state (the reference recomputes PTQ every run) ===========
def quant_state_dict(model):
    """{quantizer module path: calibrated amax tensor} for every calibrated TensorQuantizer."""
    return {name: m.amax.clone() for name, m in model.named_modules() if isinstance(m, TensorQuantizer) and m.amax is not None}


def save_quant_state(model, path):
    torch.save({"quant_state": quant_state_dict(model), "version": 1}, path)


def load_quant_state(model, path_or_state, strict=True):
    """Restore calibrated amax values onto a model that went through the same surgery. Returns the loaded names."""
    state = path_or_state
    if not isinstance(state, dict):
        state = torch.load(path_or_state, map_location="cpu", weights_only=False)
    state = state.get("quant_state", state)
    mods = {name: m for name, m in model.named_modules() if isinstance(m, TensorQuantizer)}
    missing = [k for k in state if k not in mods]
    if strict and missing:
        raise KeyError("quant state has no matching quantizer for: %s" % ", ".join(missing[:5]))
    loaded = []
    for k, v in state.items():
        if k in mods:
            mods[k]._amax = v.detach().float().cpu().reshape(-1).clone()
            loaded.append(k)
    _touch()
    return loaded
