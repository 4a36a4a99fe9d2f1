"""VGG19 perceptual loss (reference utils/perceptual_loss.py:9-45; call site losses.py:392-396).

The reference takes torchvision's VGG19 `features[0 .. final_layer]` (final_layer = 14: conv1_1 ReLU conv1_2 ReLU pool
conv2_1 ReLU conv2_2 ReLU pool conv3_1 ReLU conv3_2 ReLU conv3_3 -- the last convolution WITHOUT its ReLU), feeds it the
ImageNet-normalised fake and real images and returns the MSE (or L1, or both) of the two feature maps, the real branch
detached.  VGG parameters never reach the optimizer, so they are frozen here (the reference computes their gradients
and throws them away).

ONE path: seven 3x3 convolutions with the bias (+ ReLU) epilogue on the f32-MFMA kernels (the >= 128-channel ones through
Winograd + csrc/gemm.hip) and the 2x2 max-pools of csrc/pool.hip, channels_last activations, GPU tensors only (a CPU tensor
raises).  The torch restatement the tests compare against is oracle/torch_modules.PerceptualLossRef.  `model.<i>.weight / .bias`
sit at torchvision's indices, so `load_vgg19_features(state_dict)` accepts torchvision's `vgg19().features.state_dict()` (or the
full model's) as is.

Pretrained weights cannot be downloaded here (no network): without a state dict the layers carry torchvision's own
initialisation (kaiming_normal fan_out / zero bias) from a seeded generator -- SURVEY.md section 8(d), config 3.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

_VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vgg19_feature_layout(final_layer=14):
    """[(index, kind, cin, cout)] of torchvision vgg19().features up to and including `final_layer`."""
    out, cin, i = [], 3, 0
    for v in _VGG19_CFG:
        if v == "M":
            out.append((i, "pool", cin, cin)); i += 1
        else:
            out.append((i, "conv", cin, v)); out.append((i + 1, "relu", v, v)); cin = v; i += 2
    return [t for t in out if t[0] <= final_layer]


class _NormalizeToNHWC4(torch.autograd.Function):
    """transforms.Normalize(mean, std) + repack to the kernels' 4-channel channels_last layout in one launch
    (csrc/conv.hip image_to_nhwc4_kernel); the backward hands dL/dimage back in NCHW."""

    _inv_std = {}          # device -> 1/std; filled by forward so that a hipGraph capture of the backward copies nothing

    @staticmethod
    def forward(ctx, images):
        from . import ops
        if images.device not in _NormalizeToNHWC4._inv_std:
            _NormalizeToNHWC4._inv_std[images.device] = torch.tensor([1.0 / s for s in _STD], device=images.device).view(1, 3, 1, 1)
        return ops.image_to_nhwc4(images, None, True)

    @staticmethod
    def backward(ctx, gy):
        return (gy[:, :3] * _NormalizeToNHWC4._inv_std[gy.device]).contiguous()


class _FeatureMSE(torch.autograd.Function):
    """F.mse_loss(fake, real) with `real` constant, on channels_last feature maps.  ATen's mse backward wrote its gradient in a different
    memory format from its inputs: a strided 0.66 ms kernel plus a 0.19 ms re-layout per step on the [48, 256, 56, 56] maps of
    BASELINE configs[2].  Here the difference is kept (same layout as the inputs) and the backward is one dense scaling of it."""

    @staticmethod
    def forward(ctx, fake, real):
        d = fake - real
        ctx.save_for_backward(d)
        nrm = torch.linalg.vector_norm(d)
        return nrm * nrm / d.numel()

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return d * (g * (2.0 / d.numel())), None


class PerceptualLoss(nn.Module):
    def __init__(self, type="l2", reduction="mean", final_layer=14, seed=0):
        super().__init__()
        if type not in ("l1", "l2", "both"):
            raise NotImplementedError(type)
        self.type, self.reduction = type, reduction
        self.layout = vgg19_feature_layout(final_layer)
        gen = torch.Generator().manual_seed(seed)
        layers = []
        for idx, (i, kind, cin, cout) in enumerate(self.layout):
            if kind == "conv":
                fused_relu = idx + 1 < len(self.layout) and self.layout[idx + 1][1] == "relu"
                from .network import Conv2dMFMA
                m = Conv2dMFMA(cin, cout, 3, 1, 1, bias=True, relu=fused_relu)
                with torch.no_grad():          # torchvision's VGG initialisation, drawn in the standard NCHW order
                    w = torch.empty(cout, cin, 3, 3)
                    nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu", generator=gen)
                    m.weight.copy_(w)
                    m.bias.zero_()
                layers.append(m)
            elif kind == "relu":
                layers.append(nn.ReLU(inplace=False))
            else:
                layers.append(nn.MaxPool2d(2, 2))
        self.model = nn.Sequential(*layers)
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad_(False)

    def load_vgg19_features(self, state_dict):
        """torchvision vgg19 weights: keys `features.<i>.weight` (whole model) or `<i>.weight` (features only)."""
        own = {}
        for k, v in state_dict.items():
            k = k[len("features."):] if k.startswith("features.") else k
            head = k.split(".")[0]
            if head.isdigit() and int(head) < len(self.model) and hasattr(self.model[int(head)], "weight"):
                own[k] = v
        missing = [f"{i}.{n}" for i, m in enumerate(self.model) if hasattr(m, "weight") for n in ("weight", "bias")
                   if f"{i}.{n}" not in own]
        if missing:
            raise KeyError(f"VGG19 state dict lacks {missing}")
        with torch.no_grad():
            for k, v in own.items():
                i, n = k.split(".")
                getattr(self.model[int(i)], n).copy_(v)

    def features(self, images):
        from . import ops
        x = _NormalizeToNHWC4.apply(images)
        skip = False
        kinds = [k for (_, k, _, _) in self.layout]
        # Where a conv + ReLU is followed by another convolution or by a pool, the ReLU's backward runs inside THAT layer's backward
        # (the F(4x4, 3x3) output transform / the pool's scatter mask the gradient they produce), and the convolution in front is told
        # that its gradient arrives masked: four of the six bias_relu_bwd passes over (dy, y) of 150-620 MB each are gone.
        relu_before = False                    # x is the output of a fused conv + ReLU
        for j, (m, kind) in enumerate(zip(self.model, kinds)):
            if kind == "conv":
                nxt = kinds[j + 2] if (m.relu and j + 2 < len(kinds)) else None        # what consumes relu(conv(x))
                x = m(x, grad_premasked=nxt in ("conv", "pool"), mask_input_grad=relu_before)
                skip = m.relu                  # bias + ReLU inside the convolution's epilogue when a ReLU follows
                relu_before = m.relu
            elif kind == "relu":
                assert skip
                skip = False
            else:
                x = ops.maxpool2d(x, 2, 2, 0, relu_input=relu_before)
                relu_before = False
        return x

    def forward(self, fakeIm, realIm):
        f_fake = self.features(fakeIm)
        with torch.no_grad():
            f_real = self.features(realIm)
        if self.type == "l1":
            return F.l1_loss(f_fake, f_real, reduction=self.reduction)
        if self.type == "l2":
            if self.reduction == "mean" and f_fake.requires_grad and f_fake.is_cuda and f_fake.stride() == f_real.stride():
                return _FeatureMSE.apply(f_fake, f_real)
            return F.mse_loss(f_fake, f_real, reduction=self.reduction)
        return F.l1_loss(f_fake, f_real, reduction=self.reduction) + F.mse_loss(f_fake, f_real, reduction=self.reduction)
