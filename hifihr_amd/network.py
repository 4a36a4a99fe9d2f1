"""Encoder and regression heads with the reference's module structure and state-dict names, on the hand-written kernels.

  ResEncoder / Resnet_4C(res18)   reference network/res_encoder.py:10-50, 345-373 (layer4 strides forced to 1)
  MMPool                          reference network/res_encoder.py:247-265
  HandEncoder                     reference network/res_encoder.py:53-167
  LightEstimator                  reference network/res_encoder.py:169-209
  normalize_batch_3C              reference network/res_encoder.py:212-216 (fused with the NCHW -> NHWC4 repack)

ONE path: every forward runs the kernels of hifihr_amd/csrc through hifihr_amd/ops.py on GPU tensors; a CPU tensor raises
(no fallback).  The plain-torch restatements the tests compare against live in oracle/torch_modules.py (test infrastructure).
The reference's res18 dimensions are broken; this build uses feat 512 / low 128.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init


def weights_init(m):
    """reference network/res_encoder.py:225-237."""
    name = m.__class__.__name__
    if name.find("Block") == -1 and name.find("Conv") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    elif name.find("Linear") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    elif name.find("BatchNorm") != -1:
        init.normal_(m.weight.data, 1.0, 0.02)
    if hasattr(m, "bias") and m.bias is not None:
        init.constant_(m.bias.data, 0.0)


class Conv2dMFMA(nn.Module):
    """Bias-free nn.Conv2d whose forward / backward-data / backward-weight run as hand-written f32-MFMA implicit
    GEMM kernels (hifihr_amd/csrc/conv.hip).  The weight keeps torch's logical [K,C,R,S] shape (state-dict
    compatible with the reference) in channels_last memory format = the kernel's physical [K][R][S][C]."""

    def __init__(self, cin, cout, k, stride=1, pad=0, bias=False, relu=True):
        super().__init__()
        self.stride, self.pad, self.relu = stride, pad, relu
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last))
        init.kaiming_uniform_(self.weight, a=5 ** 0.5)                 # nn.Conv2d's default initialisation
        self.bias = None
        if bias:                                       # LightEstimator / VGG convolutions: bias (+ ReLU unless relu=False) epilogue
            bound = 1.0 / (cin * k * k) ** 0.5
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))

    def forward(self, x, want_stats=False, grad_premasked=False, mask_input_grad=False, fork=False):
        from . import ops
        w = self.weight          # the 3-channel stem: the input arrives as NHWC4; ops.conv2d pads the filter (and un-pads its gradient)
        if self.bias is not None:
            return ops.conv2d_bias_act(x, w, self.bias, self.stride, self.pad, self.relu,   # conv + bias (+ ReLU), one launch
                                       grad_premasked=grad_premasked and not self.bias.requires_grad, mask_input_grad=mask_input_grad)
        return ops.conv2d(x, w, self.stride, self.pad, want_stats, fork)


def _conv(cin, cout, k, stride, pad):
    return Conv2dMFMA(cin, cout, k, stride, pad)


def _conv_bn(conv, x, bn):
    """-> (conv(x), batch statistics for bn or None).  The statistics come out of the convolution's epilogue, but only a
    TRAINING batch-norm consumes (and cleans) the slot buffer they are added into: in evaluation mode none is requested."""
    if bn.training:
        return conv(x, want_stats=True)
    return conv(x), None


def _conv_bn_fork(conv, x, bn):
    """_conv_bn for an x that has a second consumer: -> (conv(x), statistics or None, the alias of x that consumer must read) -- its
    gradient is then added inside the convolution's backward-data launch (ops._Conv2dMFMA, `fork`)."""
    from . import ops
    if not (x.requires_grad and torch.is_grad_enabled() and ops.conv_fork_enabled()):
        out, st = _conv_bn(conv, x, bn)
        return out, st, x
    if bn.training:
        return conv(x, want_stats=True, fork=True)
    out, xa = conv(x, fork=True)
    return out, None, xa


class _PendingBN:
    """A block output whose last batch-norm (+ identity + ReLU) has not been applied yet: (raw convolution output, its batch
    statistics, the BatchNorm2d, the identity branch).  The next BasicBlock applies it inside the input transform of its first
    convolution (ops.bn_act_wino_conv) when that convolution runs on the Winograd path, or materialises it (one bn_act launch)."""

    def __init__(self, y, stats, bn, identity):
        self.y, self.stats, self.bn, self.identity = y, stats, bn, identity

    def materialize(self):
        from . import ops
        return ops.bn_act(self.y, self.stats, self.bn, self.identity, True)


def _resolve(x):
    return x.materialize() if isinstance(x, _PendingBN) else x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.lazy_out = False          # the trunk sets it where the next module takes a _PendingBN (Resnet_4C)

    def forward(self, x):
        # conv (MFMA, BN statistics from its epilogue) -> BN (+ identity) + ReLU, fused INTO the next convolution's Winograd input
        # transform where there is one (csrc/wino4_bn.hip), else one fused launch of its own (csrc/bn.hip)
        from . import ops
        idt_raw = st_ds = None
        fused = False
        if isinstance(x, _PendingBN):
            p = x
            if p.bn.training and ops.bn_wino_fusable(p.y, self.conv1.weight, p.bn, self.conv1.stride, self.conv1.pad):
                out, st, x = ops.bn_act_wino_conv(p.y, p.stats, p.bn, p.identity, self.conv1.weight, self.bn1.training)
                fused = True
            else:
                x = p.materialize()
        if not fused:
            ds = self.downsample
            if (ds is not None and self.bn1.training and ds[1].training and self.conv1.stride == 2 and torch.is_grad_enabled()
                    and ops.conv2d_pair_ok(x, self.conv1.weight, ds[0].weight, 2)):
                # conv1 and the downsample branch's 1x1 read the same x: ONE launch (ops.conv2d_pair)
                out, st, idt_raw, st_ds = ops.conv2d_pair(x, self.conv1.weight, ds[0].weight, 2)
            else:
                out, st, x = _conv_bn_fork(self.conv1, x, self.bn1)       # (x: from here on the alias the identity branch reads)
        if st is not None and ops.bn_wino_fusable(out, self.conv2.weight, self.bn1, 1, 1):
            out, st, _ = ops.bn_act_wino_conv(out, st, self.bn1, None, self.conv2.weight, self.bn2.training)
        else:
            out = ops.bn_act(out, st, self.bn1, None, True)
            out, st = _conv_bn(self.conv2, out, self.bn2)
        idt = x
        if idt_raw is not None:
            idt = ops.bn_act(idt_raw, st_ds, self.downsample[1], None, False)
        elif self.downsample is not None:
            idt, st2 = _conv_bn(self.downsample[0], x, self.downsample[1])
            idt = ops.bn_act(idt, st2, self.downsample[1], None, False)
        if self.lazy_out and self.bn2.training:
            return _PendingBN(out, st, self.bn2, idt)
        return ops.bn_act(out, st, self.bn2, idt, True)


class Bottleneck(nn.Module):
    """torchvision's ResNet v1.5 bottleneck (1x1 -> 3x3 carrying the stride -> 1x1, expansion 4), the block of the reference's
    res50 / res101 encoders (reference network/res_encoder.py:349-362 through torchvision.models.resnet50 / resnet101; the same
    class as the vendored utils/Freihand_GNN_mano/network/resnet.py:75-122)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1, 1, 0)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1, 1, 0)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        from . import ops
        x = _resolve(x)
        out, st = _conv_bn(self.conv1, x, self.bn1)
        out = ops.bn_act(out, st, self.bn1, None, True)
        out, st = _conv_bn(self.conv2, out, self.bn2)
        out = ops.bn_act(out, st, self.bn2, None, True)
        out, st = _conv_bn(self.conv3, out, self.bn3)
        idt = x
        if self.downsample is not None:
            idt, st2 = _conv_bn(self.downsample[0], x, self.downsample[1])
            idt = ops.bn_act(idt, st2, self.downsample[1], None, False)
        return ops.bn_act(out, st, self.bn3, idt, True)


class ResNet18Trunk(nn.Module):
    """torchvision-layout ResNet without avgpool/fc (ResNet-18 by default; block=Bottleneck, layers=(3,4,6,3) / (3,4,23,3) give
    ResNet-50 / -101); `layer4_stride=1` = the three stride edits of reference network/res_encoder.py:360-362."""
    block, layers = BasicBlock, (2, 2, 2, 2)

    def __init__(self, in_ch=3, layer4_stride=1, block=None, layers=None):
        super().__init__()
        if block is not None:
            self.block, self.layers = block, tuple(layers)
        self.conv1 = _conv(in_ch, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make(64, self.layers[0], 1)
        self.layer2 = self._make(128, self.layers[1], 2)
        self.layer3 = self._make(256, self.layers[2], 2)
        self.layer4 = self._make(512, self.layers[3], layer4_stride)
        for m in self.modules():
            if isinstance(m, Conv2dMFMA):
                init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, planes, blocks, stride):
        blk, out = self.block, planes * self.block.expansion
        down = None
        if stride != 1 or self.inplanes != out:
            down = nn.Sequential(_conv(self.inplanes, out, 1, stride, 0), nn.BatchNorm2d(out))
        layers = [blk(self.inplanes, planes, stride, down)]
        self.inplanes = out
        layers += [blk(out, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class Resnet_4C(nn.Module):
    def __init__(self, pretrain="res18", if_4c=False):
        super().__init__()
        arch = {"res18": (BasicBlock, (2, 2, 2, 2)), "res50": (Bottleneck, (3, 4, 6, 3)), "res101": (Bottleneck, (3, 4, 23, 3))}
        if pretrain not in arch:
            raise NotImplementedError(f"encoder '{pretrain}' is not built (res18 / res50 / res101; the HRNet variants need timm)")
        self.model = ResNet18Trunk(in_ch=4 if if_4c else 3, layer4_stride=1, block=arch[pretrain][0], layers=arch[pretrain][1])
        # BasicBlocks hand their output over un-normalised (a _PendingBN) wherever the next module is another BasicBlock: it applies the
        # batch-norm inside its first convolution's input transform.  The two block outputs this module returns are materialised.
        m = self.model
        blocks = [b for layer in (m.layer1, m.layer2, m.layer3, m.layer4) for b in layer]
        for b, nxt in zip(blocks, blocks[1:]):
            if isinstance(b, BasicBlock) and isinstance(nxt, BasicBlock) and b is not m.layer2[-1]:
                b.lazy_out = True

    def forward(self, x):
        from . import ops
        m = self.model
        h, st = _conv_bn(m.conv1, x, m.bn1)
        x = ops.bn_relu_maxpool(h, st, m.bn1)
        x = m.layer1(x)
        x_low = _resolve(m.layer2(x))
        cut = getattr(self, "segment_cut", None)
        if cut is None:
            x = _resolve(m.layer4(m.layer3(x_low)))
            return x_low, x
        # segmented backward (traineval.SegmentedGraphedTrainStep): the autograd graph is cut at the layer boundaries, so that the
        # backward of each segment is a graph launch of its own and its gradient bucket can be exchanged while the next one runs
        x_low = cut("layer2", x_low)
        x3 = cut("layer3", _resolve(m.layer3(x_low)))
        x = cut("layer4", _resolve(m.layer4(x3)))
        return x_low, x


class MMPool(nn.Module):
    """sigma(p) * maxpool + (1 - sigma(p)) * avgpool to 1 x 1, one fused kernel per direction (csrc/pool.hip)."""

    def __init__(self, shape=(1, 1), dim=1, p=0.0, eps=1e-6):
        super().__init__()
        if tuple(shape) != (1, 1):
            raise NotImplementedError("MMPool: only the (1, 1) output the reference uses is built")
        self.p = nn.Parameter(torch.ones(dim) * p, requires_grad=True)
        self.shape = shape

    def forward(self, x):
        from . import ops
        return ops.mmpool(x, self.p).reshape(x.shape[0], x.shape[1], 1, 1)


class ResEncoder(nn.Module):
    """The trunk's convolutions run as hand-written MFMA kernels on channels_last (NHWC) activations; the input normalisation
    (normalize_batch_3C) is fused with the NCHW -> NHWC4 repack."""

    def __init__(self, pretrain="res18", if_4c=False):
        super().__init__()
        self.mmpool = MMPool((1, 1))
        self.encoder1 = Resnet_4C(pretrain, if_4c=if_4c)
        if if_4c:
            raise NotImplementedError("four_channel input is not used by any new_model config")

    def forward(self, x):
        from . import ops
        x = ops.image_to_nhwc4(x)
        low, features = self.encoder1(x)
        features = self.mmpool(features).reshape(features.shape[0], -1)
        return low, features


def _mlp(dims, relu_after_first=True):
    layers = [nn.Linear(dims[0], dims[1])]
    if relu_after_first:
        layers.append(nn.ReLU(inplace=True))
    for a, b in zip(dims[1:-1], dims[2:]):
        layers.append(nn.Linear(a, b))
    seq = nn.Sequential(*layers)
    seq.apply(weights_init)
    return seq


class HandEncoder(nn.Module):
    def __init__(self, hand_model, ncomps, in_dim=1024, use_mean_shape=False, ifRender=True):
        """Every Linear (+ BatchNorm1d + ReLU) is one fused HIP launch (csrc/mlp.hip); the heads run level by level as grouped
        launches."""
        super().__init__()
        self.use_mean_shape, self.ifRender, self.hand_model = use_mean_shape, ifRender, hand_model
        self.shape_ncomp, self.pose_ncomp, self.tex_ncomp = ncomps
        self.base_layers = nn.Sequential(nn.Linear(in_dim, 1024), nn.BatchNorm1d(1024), nn.ReLU(inplace=True),
                                         nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True))
        self.base_layers.apply(weights_init)
        self.pose_reg = _mlp([512, 128, self.pose_ncomp])
        self.shape_reg = _mlp([512, 128, self.shape_ncomp])
        if hand_model == "nimble" or self.tex_ncomp:       # nimble, or MANO with the vertex-colour texture stand-in (models.py)
            self.tex_reg = _mlp([512, 128, self.tex_ncomp])
        self.trans_reg = _mlp([512, 128, 32, 3])
        if hand_model == "mano":
            self.rot_reg = _mlp([512, 128, 32, 3])
        self.scale_reg = _mlp([512, 128, 32, 1])

    def _heads_grouped(self, base, has_tex):
        """The heads level by level: all first layers in one launch, all second layers in one, the third layers of the
        three-layer heads in one (ops.linear_group) -- 3 launches instead of 13-15.  Same layers, same arithmetic."""
        from . import ops
        heads = [("pose", self.pose_reg)]
        if not self.use_mean_shape:
            heads.append(("shape", self.shape_reg))
        if self.ifRender and has_tex:
            heads.append(("tex", self.tex_reg))
        heads.append(("trans", self.trans_reg))
        if self.hand_model == "mano":
            heads.append(("rot", self.rot_reg))
        heads.append(("scale", self.scale_reg))
        cur = {name: base for name, _ in heads}
        out = {}
        depth = 0
        while cur:
            members, names = [], []
            for name, seq in heads:
                if name not in cur:
                    continue
                mods = list(seq)
                relu = depth + 1 < len(mods) and isinstance(mods[depth + 1], nn.ReLU)
                members.append((cur[name], mods[depth], relu))
                names.append((name, len(mods), relu))
            ys = ops.linear_group(members)
            nxt = {}
            for (name, nmods, relu), y in zip(names, ys):
                step = depth + (2 if relu else 1)
                if step >= nmods:
                    out[name] = y
                else:
                    nxt[name] = (y, step)
            # the heads share one layout (Linear, ReLU, Linear[, Linear]): all survivors sit at the same module index
            assert len({v[1] for v in nxt.values()}) <= 1
            cur = {k: v[0] for k, v in nxt.items()}
            depth = next(iter(nxt.values()))[1] if nxt else depth
        return out["pose"], out.get("shape"), out.get("tex"), out["scale"], out["trans"], out.get("rot")

    def forward(self, features):
        from . import ops
        bs, device = features.shape[0], features.device
        bl = self.base_layers
        base = ops.linear(ops.linear(features, bl[0], act=True, bn=bl[1]), bl[3], act=True, bn=bl[4])
        has_tex = self.hand_model == "nimble" or bool(self.tex_ncomp)
        pose_params, shape_params, texture_params, scale, trans, rot = self._heads_grouped(base, has_tex)
        if texture_params is None and self.hand_model == "nimble":
            texture_params = torch.zeros(bs, self.tex_ncomp, device=device)
        if shape_params is None:
            shape_params = torch.zeros(bs, self.shape_ncomp, device=device)
        return {"pose_params": pose_params, "shape_params": shape_params, "texture_params": texture_params,
                "scale": scale, "trans": trans, "rot": rot}


class LightEstimator(nn.Module):
    """The three convolutions (+ bias + ReLU fused) and the two max-pools run on the hand-written kernels on channels_last
    activations; same module indices / state-dict names as the reference's nn.Sequential."""

    def __init__(self, in_dim=512):
        super().__init__()
        mk = lambda ci, co, k, s: Conv2dMFMA(ci, co, k, s, 0, bias=True)
        if in_dim == 32:                       # efficientnet-b3 low features [b,32,56,56]
            conv1 = mk(32, 48, 1, 4)
        else:                                  # [b,in_dim,28,28] (512 in the reference; 128 for res18)
            conv1 = mk(in_dim, 48, 1, 2)
        self.base_layers = nn.Sequential(conv1, nn.ReLU(inplace=True), mk(48, 48, 3, 1), nn.ReLU(inplace=True),
                                         nn.MaxPool2d(3, 1, 1), mk(48, 64, 3, 2), nn.ReLU(inplace=True),
                                         nn.MaxPool2d(2, 2))
        self.light_reg = nn.Sequential(nn.Linear(256, 64), nn.ReLU(inplace=True), nn.Linear(64, 6))
        self.light_reg.apply(weights_init)
        self.hardtanh = nn.Hardtanh()

    def forward(self, low_features):
        from . import ops
        bl = self.base_layers
        x = bl[2](bl[0](low_features))                      # conv + bias + ReLU each (the nn.ReLU entries only keep the indices)
        x = bl[5](ops.maxpool2d(x, 3, 1, 1))
        if x.is_cuda and x.shape[1] % 4 == 0:
            flat = ops.maxpool2d_flatten(x, 2, 2, 0)        # pool + the reference's `.view(B, -1)` (NCHW order) in one launch each way
        else:
            base = ops.maxpool2d(x, 2, 2, 0)
            flat = base.reshape(base.shape[0], -1)
        lights = ops.linear(ops.linear(flat, self.light_reg[0], act=True), self.light_reg[2])
        # the reference checks `torch.any(colors.isnan())` here with a host sync every step (:205); omitted on purpose
        if lights.is_cuda and lights.requires_grad:
            colors, directions = ops.light_split(lights)           # hardtanh + the two slices as one autograd node (ops._LightSplit)
            return {"colors": colors, "directions": directions}
        return {"colors": self.hardtanh(lights[:, :3]), "directions": lights[:, 3:]}
