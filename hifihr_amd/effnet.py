"""EfficientNet-b3 feature extractor with the reference's structure and state-dict names.

Mirrors reference network/efficientnet_pt/model.py (MBConvBlock :17-94, EfficientNet.__init__ :97-188,
extract_features :195-215), network/efficientnet_pt/utils.py (round_filters :57-69, round_repeats :72-77,
drop_connect :82-91, Conv2dStaticSamePadding :122-145, MemoryEfficientSwish :36-52) and
network/effnet_encoder.py (EffiEncoder :6-18).  Built table-driven from the block table of SURVEY.md Appendix A:
width 1.2 / depth 1.4 / "image_size" 300 (the STATIC same-padding is computed for a 300-pixel input at EVERY layer
and applied unchanged to the 224-pixel input: k3 s1 -> (1,1), k3 s2 -> (0,1), k5 s1 -> (2,2), k5 s2 -> (1,2)),
BN eps 1e-3 / torch-momentum 0.01, SE ratio 1/4 of the block INPUT filters, drop-connect 0.2 * idx / 26 in training.

ONE path, on the hand-written kernels (GPU tensors only, a CPU tensor raises): the stem and the 1x1 expand / project / head
convolutions (96 % of the FLOPs) on the MFMA kernel, the depthwise convolutions (csrc/dwconv.hip), every BatchNorm (+ swish)
(csrc/bn.hip) and squeeze-excite (csrc/se.hip + mlp.hip).  The torch restatement the tests compare against is
oracle/torch_modules.EfficientNetB3Ref (pinned by the reference's own network/efficientnet_pt, tests/golden/effnet_b3_small.npz).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# (kernel, stride, expand, in, out, repeats) of efficientnet-b0; b3 scales width x1.2 and depth x1.4
_B0_BLOCKS = [(3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2), (3, 2, 6, 40, 80, 3), (5, 1, 6, 80, 112, 3),
              (5, 2, 6, 112, 192, 4), (3, 1, 6, 192, 320, 1)]
_WIDTH, _DEPTH, _STATIC_SIZE = 1.2, 1.4, 300
_BN_EPS, _BN_MOM = 1e-3, 0.01
_DROP_CONNECT = 0.2


def _drop_connect_uniform(x):
    """The per-sample uniform draw of utils.py:87 (`torch.rand([batch_size, 1, 1, 1], dtype, device)`), one per skip block in block
    order.  A module-level hook so that a parity test can hand in the draws the reference made (its CPU generator under
    torch.manual_seed(5)): tests/test_gpu_conv.py::test_efficientnet_b3_hip_vs_reference_golden."""
    return torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device)


def round_filters(f, width=_WIDTH, divisor=8):
    f *= width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if nf < 0.9 * f:
        nf += divisor
    return int(nf)


def round_repeats(r, depth=_DEPTH):
    return int(math.ceil(depth * r))


def static_same_pad(k, s, size=_STATIC_SIZE):
    """(left, right, top, bottom) exactly as Conv2dStaticSamePadding computes it for a `size`-pixel input."""
    o = math.ceil(size / s)
    p = max((o - 1) * s + (k - 1) + 1 - size, 0)
    return (p // 2, p - p // 2, p // 2, p - p // 2)


def b3_block_table():
    """[(k, stride, expand, in, out)] for the 26 blocks of efficientnet-b3."""
    out = []
    for k, s, e, i, o, r in _B0_BLOCKS:
        i, o, r = round_filters(i), round_filters(o), round_repeats(r)
        out.append((k, s, e, i, o))
        out += [(k, 1, e, o, o)] * (r - 1)
    return out


# bn0 + swish inside the depthwise kernels' loads (ops._BNSwishDwConv; built and parity-tested in round 5) is OFF by default: MEASURED
# slower (profiles/r05_time_dw_bnswish.txt, batch 48: forward 1 542 -> 1 623 us per step, backward-weight 1 033 -> 3 089 us; config 3
# 34.45 -> 36.80 ms/step).  The streaming depthwise kernels load every input element 1.7-10 x (the window overlap) and each load then
# pays a v_exp + v_rcp: they turn from L1-bound into VALU-bound, and the weight gradient -- which needs the ACTIVATED tensor again --
# pays it a second time.  What would work is a depthwise kernel that stages an input tile in LDS (each element activated once) for
# forward and weight gradient both; not built.  HIFIHR_EFFNET_FUSE_BN0=1 turns the fused path on.
_FUSE_BN0 = os.environ.get("HIFIHR_EFFNET_FUSE_BN0", "0") != "0"


class SqueezeExciteConv(nn.Conv2d):
    """The 1x1 `_se_reduce` / `_se_expand` convolutions: parameter holders with nn.Conv2d's state-dict names (weight, bias);
    the arithmetic runs inside ops.squeeze_excite."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1, 1, 0, 1, 1, True)

    def forward(self, x):
        raise RuntimeError("squeeze-excite convolutions run fused (ops.squeeze_excite), not on their own")


class PointwiseConvMFMA(nn.Module):
    """Bias-free 1x1 convolution on the hand-written MFMA implicit-GEMM kernel (weight: logical [K,C,1,1])."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1).contiguous(memory_format=torch.channels_last))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))          # nn.Conv2d's default initialisation

    def forward(self, x, want_stats=False, fork=False):
        from . import ops
        return ops.conv2d(x, self.weight, 1, 0, want_stats, fork)


class DepthwiseConvHIP(nn.Module):
    """Depthwise k x k convolution with the static same padding on the hand-written kernels (weight [C,1,k,k])."""

    def __init__(self, c, k, stride):
        super().__init__()
        self.stride, self.pad4 = stride, static_same_pad(k, stride)
        self.weight = nn.Parameter(torch.empty(c, 1, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x, want_stats=False):
        from . import ops
        return ops.dwconv2d(x, self.weight, self.stride, self.pad4, want_stats)


def _pointwise(cin, cout):
    return PointwiseConvMFMA(cin, cout)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=_BN_MOM, eps=_BN_EPS)


def _conv_bn_swish(conv, bn, x, act=True, fork=False):
    """conv -> BN -> swish: batch statistics from the convolution's epilogue in training mode (none requested in evaluation mode:
    only a training batch-norm consumes and cleans the slot buffer), BN + swish one fused launch.  fork: -> (result, alias of x for x's
    other consumer -- the block's skip connection; its gradient is added inside the convolution's backward-data launch, ops._Conv2dMFMA)."""
    from . import ops
    if fork:
        if bn.training:
            y, st, xa = conv(x, want_stats=True, fork=True)
        else:
            (y, xa), st = conv(x, fork=True), None
        return ops.bn_act(y, st, bn, None, "swish" if act else None), xa
    if bn.training:
        y, st = conv(x, want_stats=True)
    else:
        y, st = conv(x), None
    return ops.bn_act(y, st, bn, None, "swish" if act else None)


class MBConvBlock(nn.Module):
    def __init__(self, k, stride, expand, cin, cout):
        super().__init__()
        self.stride, self.cin, self.cout, self.expand = stride, cin, cout, expand
        mid = cin * expand
        if expand != 1:
            self._expand_conv = _pointwise(cin, mid)
            self._bn0 = _bn(mid)
        self._depthwise_conv = DepthwiseConvHIP(mid, k, stride)
        self._bn1 = _bn(mid)
        sq = max(1, int(cin * 0.25))
        self._se_reduce = SqueezeExciteConv(mid, sq)
        self._se_expand = SqueezeExciteConv(sq, mid)
        self._project_conv = _pointwise(mid, cout)
        self._bn2 = _bn(cout)

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        from . import ops
        fused_expand = None
        if self.expand != 1 and _FUSE_BN0 and self._bn0.training and x.is_cuda:
            # (HIFIHR_EFFNET_FUSE_BN0=0: bn0 + swish as a launch of its own, the activated tensor materialised)
            skip = self.stride == 1 and self.cin == self.cout
            if skip and inputs.requires_grad and torch.is_grad_enabled() and os.environ.get("HIFIHR_EFFNET_FORK", "1") != "0":
                e, st0, inputs = self._expand_conv(x, want_stats=True, fork=True)
            else:
                e, st0 = self._expand_conv(x, want_stats=True)
            fused_expand = (e, st0)
        elif self.expand != 1:
            # (the skip connection's gradient joins the expand convolution's backward-data launch -- which then runs on conv_igemm_kernel,
            #  the only epilogue that adds one, instead of the row-share GEMM: config 3 35.09 / 35.14 -> 35.05 / 35.06 ms/step and 19 launches
            #  fewer; HIFIHR_EFFNET_FORK=0 leaves the sum to autograd)
            skip = self.stride == 1 and self.cin == self.cout
            if skip and inputs.requires_grad and torch.is_grad_enabled() and os.environ.get("HIFIHR_EFFNET_FORK", "1") != "0":
                x, inputs = _conv_bn_swish(self._expand_conv, self._bn0, x, fork=True)
            else:
                x = _conv_bn_swish(self._expand_conv, self._bn0, x)
        if fused_expand is not None:
            # expand convolution -> [bn0 + swish applied inside the depthwise kernel's loads] -> depthwise convolution (ops._BNSwishDwConv)
            e, st0 = fused_expand
            dwc = self._depthwise_conv
            y, st = ops.bn_swish_dwconv(e, st0, self._bn0, dwc.weight, dwc.stride, dwc.pad4, want_stats=True)
            x = ops.bn_act(y, st, self._bn1, None, "swish")
        else:
            x = _conv_bn_swish(self._depthwise_conv, self._bn1, x)         # statistics from the depthwise kernel's epilogue
        x = ops.squeeze_excite(x, self._se_reduce, self._se_expand)        # pool + 2 small linears + scale, fused
        x = _conv_bn_swish(self._project_conv, self._bn2, x, act=False)
        if self.stride == 1 and self.cin == self.cout:
            if drop_connect_rate and self.training:                       # utils.py:82-91: x / keep * floor(keep + u) + inputs, one launch
                return ops.drop_connect_add(x, inputs, _drop_connect_uniform(x), 1 - drop_connect_rate)
            x = x + inputs
        return x


class EfficientNetB3(nn.Module):
    """`extract_features` of EfficientNet.from_name('efficientnet-b3') (the unused classifier `_fc` is omitted)."""

    def __init__(self):
        super().__init__()
        stem = round_filters(32)
        from .network import Conv2dMFMA
        self._conv_stem = Conv2dMFMA(3, stem, 3, stride=2, pad=0)          # the static same padding goes into the NHWC4 repack
        self._bn0 = _bn(stem)
        table = b3_block_table()
        assert len(table) == 26 and table[0][3] == stem
        self._blocks = nn.ModuleList([MBConvBlock(k, s, e, i, o) for (k, s, e, i, o) in table])
        head = round_filters(1280)
        self._conv_head = _pointwise(table[-1][4], head)
        self._bn1 = _bn(head)
        self.out_channels, self.low_channels = head, table[4][4]

    def extract_features(self, x):
        from . import ops
        x4 = ops.image_to_nhwc4(x, pad4=static_same_pad(3, 2), normalize=False)          # model.py:197-199 (no normalisation)
        x = _conv_bn_swish(self._conv_stem, self._bn0, x4)
        low = None
        n = len(self._blocks)
        for idx, blk in enumerate(self._blocks):
            x = blk(x, drop_connect_rate=_DROP_CONNECT * float(idx) / n)
            if idx == 4:
                low = x                                                   # model.py:209-210
        x = _conv_bn_swish(self._conv_head, self._bn1, x)
        return x, low


class EffiEncoder(nn.Module):
    """reference network/effnet_encoder.py:6-18 (no input normalisation; AvgPool2d(7) on the 7x7 head features)."""

    def __init__(self, pretrain="effb3"):
        super().__init__()
        assert pretrain == "effb3"
        self.encoder = EfficientNetB3()
        self.pool = nn.AvgPool2d(7, stride=1)

    def forward(self, x):
        features, low = self.encoder.extract_features(x)
        features = self.pool(features)
        return low, features.reshape(features.shape[0], -1)
