"""ResNet-50 input producer with frozen BatchNorm, plus the (backbone, position-encoding) pair the
model indexes as ``backbone[0]`` / ``backbone[1]``.

Reference: models/DDETR_backbone.py (FrozenBatchNorm2d :27-59, BackboneBase :62-97 -- only layer2-4
train, features C3/C4/C5 at strides 8/16/32 with nearest-interpolated masks, Joiner :135-160).  The
reference builds the trunk from torchvision (absent here); this is a plain restatement of the
ResNet-50 v1.5 topology with torchvision's parameter names (``body.layerK.B.convJ`` ...), so
ImageNet / reference checkpoints load.  3x3 / strided convolutions are MIOpen's; the 1x1 stride-1
convolutions of the bf16 channels-last path run as token-major GEMMs with BatchNorm and ReLU folded in
(`pointwise_conv_bn`).
"""
from __future__ import annotations


import torch
import torch.nn.functional as F
from torch import nn

from .blocks import NestedTensor, PositionEmbeddingSine
from .linear import token_linear

# 1x1 convolutions of channels-last bf16 feature maps as token-major GEMMs (tests switch it off to compare)
pointwise_as_gemm = True
# relu(a + b) of the bottleneck tails as one HIP pass (attribute; `bench.py --set backbone.fused_add_relu=0` for an A/B)
fused_add_relu = True


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (buffers), eps inside the rsqrt."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, *args)

    def _folded(self, dtype):
        """(scale, bias) of the frozen affine map, computed in float32 once and cached per dtype
        (the statistics are buffers that only change when a checkpoint is loaded)."""
        key = (dtype, self.weight._version, self.bias._version, self.running_mean._version,
               self.running_var._version, self.weight.device)
        if getattr(self, "_fold_key", None) != key:
            w, b = self.weight.float(), self.bias.float()
            scale = w * (self.running_var.float() + self.eps).rsqrt()
            bias = b - self.running_mean.float() * scale
            self._fold = (scale.view(1, -1, 1, 1).to(dtype), bias.view(1, -1, 1, 1).to(dtype))
            self._fold_key = key
        return self._fold

    def forward(self, x):
        scale, bias = self._folded(x.dtype)
        return torch.addcmul(bias, x, scale)

    def folded_vectors(self, dtype):
        """(scale [C], bias [C]) of the frozen affine map"""
        scale, bias = self._folded(dtype)
        return scale.view(-1), bias.view(-1)


class AddReLUFunction(torch.autograd.Function):
    """relu(a + b) as one HIP pass (csrc/elementwise.hip; bit-identical to `add` then `relu` in bfloat16); the gradient
    of both addends is dy * (y > 0)."""

    @staticmethod
    def forward(ctx, a, b):
        from . import _lib, roofline
        y = torch.empty_like(a)
        st = _lib.lib().add_relu_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(),
                                      torch.cuda.current_stream(a.device).cuda_stream)
        if st:
            raise RuntimeError("add_relu: " + _lib.strerror(st))
        roofline.add(roofline.tensor_bytes(a, b, y))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(dy, y, 0)
        return g, g


class AffineReLUFunction(torch.autograd.Function):
    """relu(x * scale[c] + bias[c]) of a channels-last bf16 tensor (frozen BatchNorm + ReLU) as one HIP pass each way
    (csrc/elementwise.hip); scale / bias are constants (buffers of FrozenBatchNorm2d)."""

    @staticmethod
    def forward(ctx, x, scale, bias):
        from . import _lib, roofline
        y = torch.empty_like(x)
        C = x.shape[1]
        st = _lib.lib().affine_relu_bf16(x.data_ptr(), scale.data_ptr(), bias.data_ptr(), y.data_ptr(), x.numel(), C,
                                         torch.cuda.current_stream(x.device).cuda_stream)
        if st:
            raise RuntimeError("affine_relu: " + _lib.strerror(st))
        roofline.add(roofline.tensor_bytes(x, y))
        ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        from . import _lib, roofline
        y, scale = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(y)
        st = _lib.lib().affine_relu_backward_bf16(dy.data_ptr(), y.data_ptr(), scale.data_ptr(), dx.data_ptr(), y.numel(),
                                                  y.shape[1], torch.cuda.current_stream(y.device).cuda_stream)
        if st:
            raise RuntimeError("affine_relu_backward: " + _lib.strerror(st))
        roofline.add(roofline.tensor_bytes(dy, y, dx))
        return dx, None, None


def bn_relu(x, bn):
    """relu(bn(x)) for a FrozenBatchNorm2d; one fused pass for channels-last bf16 GPU tensors"""
    if (fused_add_relu and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
            and not torch.is_autocast_enabled() and x.data_ptr() % 16 == 0):
        scale, bias = bn.folded_vectors(x.dtype)
        return AffineReLUFunction.apply(x, scale.contiguous(), bias.contiguous())
    return F.relu(bn(x))


def add_relu(a, b):
    """relu(a + b); the fused kernel when both are bf16 GPU tensors of one shape and memory layout"""
    if (fused_add_relu and a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape == b.shape
            and a.stride() == b.stride() and a.numel() % 8 == 0 and not torch.is_autocast_enabled()
            and (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last))
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        return AddReLUFunction.apply(a, b)
    return F.relu(a + b)


def pointwise_conv_bn(x, conv, bn, relu):
    """1x1 stride-1 convolution + frozen BatchNorm (+ ReLU) of a channels-last bf16 tensor as ONE GEMM over
    the N*H*W pixels: the BN scale is folded into the weight rows (a [Cout, Cin] multiply, differentiable
    w.r.t. the weight), the BN shift is the GEMM bias and the ReLU its epilogue, so the separate BN and
    activation passes over the feature map disappear and the weight gradient runs on the token-major MFMA
    kernel (linear.py) instead of MIOpen's split-K path with its float32 workspace zero/cast kernels."""
    if conv.stride != (1, 1):
        # a strided 1x1 convolution reads every stride-th pixel: subsample, then the same GEMM (the stages' downsample
        # branches; their MIOpen weight-gradient solvers were two of the three remaining non-repeatable kernels of the
        # step, profiles/r03_nondeterminism.txt)
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
    N, C, H, W = x.shape
    scale, bias = bn.folded_vectors(x.dtype)
    w = conv.weight.reshape(conv.out_channels, C) * scale[:, None]
    y = token_linear(x.permute(0, 2, 3, 1).reshape(N * H * W, C), w, bias, relu=relu)
    return y.view(N, H, W, conv.out_channels).permute(0, 3, 1, 2)


class RepeatableConv3x3(torch.autograd.Function):
    """A stride-1, pad-1 3x3 convolution whose input gradient is computed as a FORWARD convolution of the output
    gradient with the flipped, transposed weight instead of MIOpen's backward-data solver.

    tools/nondet_modules.py (profiles/r03_nondeterminism.txt): with the extra-level convolution replaced, the only kernel
    of the full-size train step whose result is not repeatable bit for bit is the backward-data solver MIOpen picks for
    the trunk's 3x3 convolutions (`layer4.x.conv2`); its forward solvers and its weight-gradient solvers are repeatable.
    Same arithmetic (a correlation with the 180-degree rotated kernel), a repeatable kernel."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.conv2d(x, weight, None, 1, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = weight.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)
            dx = F.conv2d(dy, wt, None, 1, 1)
        if ctx.needs_input_grad[1]:
            dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        return dx, dw


repeatable_conv_backward = True      # (tools/grad_repeat.py flips the attribute for its A/B; not read from the environment)


def conv3x3(x, conv):
    if (repeatable_conv_backward and x.is_cuda and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and conv.kernel_size == (3, 3) and torch.is_grad_enabled()
            and (x.requires_grad or conv.weight.requires_grad) and not torch.is_autocast_enabled()):
        return RepeatableConv3x3.apply(x, conv.weight)
    return conv(x)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)      # v1.5: stride on the 3x3
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        fast = (pointwise_as_gemm and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_autocast_enabled()
                and x.is_contiguous(memory_format=torch.channels_last))
        if self.downsample is None:
            idt = x
        elif fast and self.downsample[0].kernel_size == (1, 1) and self.downsample[0].padding == (0, 0):
            idt = pointwise_conv_bn(x, self.downsample[0], self.downsample[1], relu=False)   # (BN folded into the GEMM)
        else:
            idt = self.downsample(x)
        if fast:
            out = pointwise_conv_bn(x, self.conv1, self.bn1, relu=True)
            out = bn_relu(conv3x3(out, self.conv2), self.bn2)
            if out.is_contiguous(memory_format=torch.channels_last):
                return add_relu(pointwise_conv_bn(out, self.conv3, self.bn3, relu=False), idt)
            return F.relu(self.bn3(self.conv3(out)) + idt)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        return F.relu(self.bn3(self.conv3(out)) + idt)


class ResNet50Body(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._stage(64, 3, 1)
        self.layer2 = self._stage(128, 4, 2)
        self.layer3 = self._stage(256, 6, 2)
        self.layer4 = self._stage(512, 3, 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = F.max_pool2d(bn_relu(self.conv1(x), self.bn1), 3, stride=2, padding=1)
        c2 = self.layer1(x)
        c3 = self.layer2(c2)
        c4 = self.layer3(c3)
        c5 = self.layer4(c4)
        return c3, c4, c5


class Backbone(nn.Module):
    def __init__(self, train_backbone=True):
        super().__init__()
        self.body = ResNet50Body()
        for name, p in self.body.named_parameters():
            if not train_backbone or ('layer2' not in name and 'layer3' not in name and 'layer4' not in name):
                p.requires_grad_(False)
        self.strides = [8, 16, 32]
        self.num_channels = [512, 1024, 2048]

    def forward(self, tensor_list: NestedTensor):
        m = tensor_list.mask
        assert m is not None
        out = []
        for x in self.body(tensor_list.tensors):
            mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
            out.append(NestedTensor(x, mask, getattr(tensor_list, "no_padding", False)))
        return out


class Joiner(nn.Sequential):
    """[0] = backbone, [1] = position encoding; returns (features, positions)."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        feats = self[0](tensor_list)
        return feats, [self[1](x, out_dtype=x.tensors.dtype) for x in feats]


def build_r50_backbone(hidden_dim=256, train_backbone=True):
    return Joiner(Backbone(train_backbone), PositionEmbeddingSine(hidden_dim // 2, normalize=True))
