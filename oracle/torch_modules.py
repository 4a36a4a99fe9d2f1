"""ORACLE (test infrastructure, not product code): plain-torch restatements of the reference's encoder / head / VGG modules.

Independent of the product package: nothing here imports `hifihr_amd`.  The classes carry the reference's module structure and
state-dict names, so a product module's `state_dict()` loads into its restatement (and the other way round) and both can be run
on the same weights.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

  ResEncoderRef / Resnet4CRef      reference network/res_encoder.py:10-50, 345-373 on the vendored
                                   utils/Freihand_GNN_mano/network/resnet.py layout (BasicBlock :40-72, Bottleneck :75-122),
                                   layer4 strides forced to 1 (:360-362); pinned by tests/golden/resnet18_small.npz
  MMPoolRef                        network/res_encoder.py:247-265
  HandEncoderRef                   network/res_encoder.py:53-167
  LightEstimatorRef                network/res_encoder.py:169-209
  EffiEncoderRef / EfficientNetB3Ref  network/effnet_encoder.py:6-18, network/efficientnet_pt/model.py:17-215, utils.py:36-145;
                                   pinned by tests/golden/effnet_b3_small.npz
  PerceptualLossRef                utils/perceptual_loss.py:9-45 (torchvision VGG19 features[0:15]; weights seeded: unpinned)
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init


def weights_init(m):
    """network/res_encoder.py:225-237."""
    name = m.__class__.__name__
    if name.find("Block") == -1 and name.find("Conv") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    elif name.find("Linear") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    elif name.find("BatchNorm") != -1:
        init.normal_(m.weight.data, 1.0, 0.02)
    if hasattr(m, "bias") and m.bias is not None:
        init.constant_(m.bias.data, 0.0)


def normalize_batch_3C(batch):
    """network/res_encoder.py:212-216."""
    mean = batch.new_tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)
    std = batch.new_tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
    return (batch - mean) / std


# ------------------------------------------------------------------------------------------------
# ResNet trunk
# ------------------------------------------------------------------------------------------------
class BasicBlockRef(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + identity)


class BottleneckRef(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + identity)


_ARCH = {"res18": (BasicBlockRef, (2, 2, 2, 2)), "res50": (BottleneckRef, (3, 4, 6, 3)), "res101": (BottleneckRef, (3, 4, 23, 3))}


class ResNetTrunkRef(nn.Module):
    def __init__(self, block, layers, in_ch=3, layer4_stride=1):
        super().__init__()
        self.block = block
        self.conv1 = nn.Conv2d(in_ch, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], layer4_stride)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, planes, blocks, stride):
        out = planes * self.block.expansion
        down = None
        if stride != 1 or self.inplanes != out:
            down = nn.Sequential(nn.Conv2d(self.inplanes, out, 1, stride, 0, bias=False), nn.BatchNorm2d(out))
        mods = [self.block(self.inplanes, planes, stride, down)]
        self.inplanes = out
        mods += [self.block(out, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


class Resnet4CRef(nn.Module):
    def __init__(self, pretrain="res18"):
        super().__init__()
        block, layers = _ARCH[pretrain]
        self.model = ResNetTrunkRef(block, layers, in_ch=3, layer4_stride=1)

    def forward(self, x):
        m = self.model
        x = F.max_pool2d(F.relu(m.bn1(m.conv1(x))), 3, 2, 1)
        x = m.layer1(x)
        x_low = m.layer2(x)
        x = m.layer4(m.layer3(x_low))
        return x_low, x


class MMPoolRef(nn.Module):
    def __init__(self, shape=(1, 1), dim=1, p=0.0):
        super().__init__()
        self.p = nn.Parameter(torch.ones(dim) * p, requires_grad=True)
        self.shape = shape

    def forward(self, x):
        w = torch.sigmoid(self.p)
        return F.adaptive_max_pool2d(x, self.shape) * w + F.adaptive_avg_pool2d(x, self.shape) * (1 - w)


class ResEncoderRef(nn.Module):
    def __init__(self, pretrain="res18"):
        super().__init__()
        self.mmpool = MMPoolRef((1, 1))
        self.encoder1 = Resnet4CRef(pretrain)

    def forward(self, x):
        low, features = self.encoder1(normalize_batch_3C(x))
        return low, self.mmpool(features).reshape(features.shape[0], -1)


# ------------------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------------------
def _mlp(dims):
    layers = [nn.Linear(dims[0], dims[1]), nn.ReLU(inplace=True)]
    for a, b in zip(dims[1:-1], dims[2:]):
        layers.append(nn.Linear(a, b))
    seq = nn.Sequential(*layers)
    seq.apply(weights_init)
    return seq


class HandEncoderRef(nn.Module):
    def __init__(self, hand_model, ncomps, in_dim=1024, use_mean_shape=False, ifRender=True):
        super().__init__()
        self.use_mean_shape, self.ifRender, self.hand_model = use_mean_shape, ifRender, hand_model
        self.shape_ncomp, self.pose_ncomp, self.tex_ncomp = ncomps
        self.base_layers = nn.Sequential(nn.Linear(in_dim, 1024), nn.BatchNorm1d(1024), nn.ReLU(inplace=True),
                                         nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True))
        self.base_layers.apply(weights_init)
        self.pose_reg = _mlp([512, 128, self.pose_ncomp])
        self.shape_reg = _mlp([512, 128, self.shape_ncomp])
        if hand_model == "nimble" or self.tex_ncomp:
            self.tex_reg = _mlp([512, 128, self.tex_ncomp])
        self.trans_reg = _mlp([512, 128, 32, 3])
        if hand_model == "mano":
            self.rot_reg = _mlp([512, 128, 32, 3])
        self.scale_reg = _mlp([512, 128, 32, 1])

    def forward(self, features):
        bs, device = features.shape[0], features.device
        base = self.base_layers(features)
        has_tex = self.hand_model == "nimble" or bool(self.tex_ncomp)
        pose_params = self.pose_reg(base)
        scale = self.scale_reg(base)
        trans = self.trans_reg(base)
        rot = self.rot_reg(base) if self.hand_model == "mano" else None
        texture_params = self.tex_reg(base) if (self.ifRender and has_tex) else None
        shape_params = None if self.use_mean_shape else self.shape_reg(base)
        if texture_params is None and self.hand_model == "nimble":
            texture_params = torch.zeros(bs, self.tex_ncomp, device=device)
        if shape_params is None:
            shape_params = torch.zeros(bs, self.shape_ncomp, device=device)
        return {"pose_params": pose_params, "shape_params": shape_params, "texture_params": texture_params,
                "scale": scale, "trans": trans, "rot": rot}


class LightEstimatorRef(nn.Module):
    def __init__(self, in_dim=512):
        super().__init__()
        conv1 = nn.Conv2d(32, 48, 1, 4) if in_dim == 32 else nn.Conv2d(in_dim, 48, 1, 2)
        self.base_layers = nn.Sequential(conv1, nn.ReLU(inplace=True), nn.Conv2d(48, 48, 3, 1), nn.ReLU(inplace=True),
                                         nn.MaxPool2d(3, 1, 1), nn.Conv2d(48, 64, 3, 2), nn.ReLU(inplace=True), nn.MaxPool2d(2, 2))
        self.light_reg = nn.Sequential(nn.Linear(256, 64), nn.ReLU(inplace=True), nn.Linear(64, 6))
        self.light_reg.apply(weights_init)

    def forward(self, low_features):
        base = self.base_layers(low_features)
        lights = self.light_reg(base.reshape(base.shape[0], -1))
        return {"colors": F.hardtanh(lights[:, :3]), "directions": lights[:, 3:]}


# ------------------------------------------------------------------------------------------------
# EfficientNet-b3 (width 1.2, depth 1.4, static "same" padding computed for 300-pixel inputs)
# ------------------------------------------------------------------------------------------------
_B0 = [(3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2), (3, 2, 6, 40, 80, 3), (5, 1, 6, 80, 112, 3),
       (5, 2, 6, 112, 192, 4), (3, 1, 6, 192, 320, 1)]


def _round_filters(f, width=1.2, divisor=8):
    f *= width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    return int(nf + divisor if nf < 0.9 * f else nf)


def _same_pad(k, s, size=300):
    o = math.ceil(size / s)
    p = max((o - 1) * s + (k - 1) + 1 - size, 0)
    return (p // 2, p - p // 2, p // 2, p - p // 2)


def _b3_table():
    out = []
    for k, s, e, i, o, r in _B0:
        i, o, r = _round_filters(i), _round_filters(o), int(math.ceil(1.4 * r))
        out.append((k, s, e, i, o))
        out += [(k, 1, e, o, o)] * (r - 1)
    return out


class _SwishRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, i):
        ctx.save_for_backward(i)
        return i * torch.sigmoid(i)

    @staticmethod
    def backward(ctx, g):
        i, = ctx.saved_tensors
        s = torch.sigmoid(i)
        return g * (s * (1 + i * (1 - s)))


class SamePadConv2dRef(nn.Conv2d):
    def __init__(self, cin, cout, k, stride=1, groups=1, bias=True):
        super().__init__(cin, cout, k, stride, 0, 1, groups, bias)
        self.pad4 = _same_pad(k, stride)

    def forward(self, x):
        if any(self.pad4):
            x = F.pad(x, self.pad4)
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, 1, self.groups)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=0.01, eps=1e-3)


class MBConvBlockRef(nn.Module):
    def __init__(self, k, stride, expand, cin, cout):
        super().__init__()
        self.stride, self.cin, self.cout, self.expand = stride, cin, cout, expand
        mid = cin * expand
        if expand != 1:
            self._expand_conv = SamePadConv2dRef(cin, mid, 1, bias=False)
            self._bn0 = _bn(mid)
        self._depthwise_conv = SamePadConv2dRef(mid, mid, k, stride, groups=mid, bias=False)
        self._bn1 = _bn(mid)
        sq = max(1, int(cin * 0.25))
        self._se_reduce = SamePadConv2dRef(mid, sq, 1)
        self._se_expand = SamePadConv2dRef(sq, mid, 1)
        self._project_conv = SamePadConv2dRef(mid, cout, 1, bias=False)
        self._bn2 = _bn(cout)

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand != 1:
            x = _SwishRef.apply(self._bn0(self._expand_conv(x)))
        x = _SwishRef.apply(self._bn1(self._depthwise_conv(x)))
        s = self._se_expand(_SwishRef.apply(self._se_reduce(F.adaptive_avg_pool2d(x, 1))))
        x = torch.sigmoid(s) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            if drop_connect_rate and self.training:
                keep = 1 - drop_connect_rate
                x = x / keep * torch.floor(keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device))
            x = x + inputs
        return x


class EfficientNetB3Ref(nn.Module):
    def __init__(self):
        super().__init__()
        stem = _round_filters(32)
        self._conv_stem = SamePadConv2dRef(3, stem, 3, 2, bias=False)
        self._bn0 = _bn(stem)
        table = _b3_table()
        self._blocks = nn.ModuleList([MBConvBlockRef(*row) for row in table])
        head = _round_filters(1280)
        self._conv_head = SamePadConv2dRef(table[-1][4], head, 1, bias=False)
        self._bn1 = _bn(head)

    def extract_features(self, x):
        x = _SwishRef.apply(self._bn0(self._conv_stem(x)))
        low, n = None, len(self._blocks)
        for idx, blk in enumerate(self._blocks):
            x = blk(x, drop_connect_rate=0.2 * float(idx) / n)
            if idx == 4:
                low = x
        return _SwishRef.apply(self._bn1(self._conv_head(x))), low


class EffiEncoderRef(nn.Module):
    def __init__(self, pretrain="effb3"):
        super().__init__()
        self.encoder = EfficientNetB3Ref()

    def forward(self, x):
        features, low = self.encoder.extract_features(x)
        features = F.avg_pool2d(features, 7, 1)
        return low, features.reshape(features.shape[0], -1)


# ------------------------------------------------------------------------------------------------
# VGG19 perceptual loss
# ------------------------------------------------------------------------------------------------
_VGG19 = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


class PerceptualLossRef(nn.Module):
    """utils/perceptual_loss.py:9-45: torchvision vgg19().features[0 : final_layer + 1] on ImageNet-normalised images, the real
    branch detached; `model.<i>.weight` at torchvision's indices.  Seeded torchvision initialisation (no downloaded weights)."""

    def __init__(self, type="l2", reduction="mean", final_layer=14, seed=0):
        super().__init__()
        self.type, self.reduction = type, reduction
        gen = torch.Generator().manual_seed(seed)
        layers, cin = [], 3
        for v in _VGG19:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                conv = nn.Conv2d(cin, v, 3, 1, 1)
                with torch.no_grad():
                    w = torch.empty(v, cin, 3, 3)
                    init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu", generator=gen)
                    conv.weight.copy_(w)
                    conv.bias.zero_()
                layers += [conv, nn.ReLU(inplace=False)]
                cin = v
        self.model = nn.Sequential(*layers[: final_layer + 1])
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad_(False)

    def features(self, images):
        mean = images.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = images.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        return self.model((images - mean) / std)

    def forward(self, fakeIm, realIm):
        f_fake = self.features(fakeIm)
        with torch.no_grad():
            f_real = self.features(realIm)
        if self.type == "l1":
            return F.l1_loss(f_fake, f_real, reduction=self.reduction)
        if self.type == "l2":
            return F.mse_loss(f_fake, f_real, reduction=self.reduction)
        return F.l1_loss(f_fake, f_real, reduction=self.reduction) + F.mse_loss(f_fake, f_real, reduction=self.reduction)
