"""GPU parity: MFMA implicit-GEMM convolution through the C ABI vs plain PyTorch fp32 conv2d (CPU), on every
distinct (Cin,Cout,k,stride,HW) shape of the modified ResNet-18 plus edge cases."""
import os

import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu

# (H_in, C, K, R, stride, pad) of ResNet-18 with layer4 strides forced to 1, at 224x224 input
RESNET18_SHAPES = [(224, 4, 64, 7, 2, 3), (56, 64, 64, 3, 1, 1), (56, 64, 128, 3, 2, 1), (28, 128, 128, 3, 1, 1),
                   (56, 64, 128, 1, 2, 0), (28, 128, 256, 3, 2, 1), (14, 256, 256, 3, 1, 1), (28, 128, 256, 1, 2, 0),
                   (14, 256, 512, 3, 1, 1), (14, 512, 512, 3, 1, 1), (14, 256, 512, 1, 1, 0)]


@pytest.fixture(scope="module")
def lib():
    from hifihr_amd._lib import get_lib
    assert torch.cuda.is_available()
    return get_lib()


@pytest.mark.parametrize("shape", RESNET18_SHAPES)
def test_resnet18_conv_shapes(lib, shape):
    H, C, K, R, s, p = shape
    kc.conv_case(lib, "cuda", 4, H, H, C, K, R, s, p, seed=H + C, rtol=3e-5)


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", [(3, 9, 7, 16, 64, 3, 1, 1), (1, 6, 6, 48, 48, 3, 1, 0), (32, 14, 14, 512, 512, 3, 1, 1),
                                                    (4, 224, 224, 4, 64, 3, 1, 1),      # VGG19 conv1_1: backward-data onto 4 channels (conv3x3_oc4_kernel)
                                                    (2, 60, 500, 64, 64, 3, 1, 1)])     # conv_halo_kernel with a ragged last column tile
def test_conv_edge_and_full_batch(lib, N, H, W, C, K, R, stride, pad):
    kc.conv_case(lib, "cuda", N, H, W, C, K, R, stride, pad, seed=1, bias=(K == 48), rtol=5e-5)


def test_image_to_nhwc4(lib):
    kc.image_to_nhwc4_case(lib, "cuda")


def test_resnet18_trunk_mfma_vs_reference_golden(golden_dir):
    """The trunk running on the hand-written MFMA convolutions reproduces the reference's vendored ResNet-18 (with the
    layer4 stride edits) forward and backward from name-seeded weights -- same fixture as the CPU test."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    g = np.load(os.path.join(golden_dir, "resnet18_small.npz"))
    enc = Resnet_4C("res18")                            # stem + blocks on the MFMA convs and the fused BN/add/ReLU kernels
    enc.model.load_state_dict(seeded_state_dict(enc.model))
    enc = enc.cuda().train()
    net = enc.model
    x = ops.image_to_nhwc4(torch.tensor(g["x"]).cuda())
    low, feat = enc(x)
    np.testing.assert_allclose(low.detach().cpu().numpy(), g["low"], atol=5e-5, rtol=1e-4)
    # features: 2e-4 -- layers 2-4 run Winograd F(4x4, 3x3) (each convolution within 1e-5 of max |y| of the direct result, csrc/wino4.hip)
    # and a batch of 2 through 20 train-mode batch-norms amplifies that rounding: 9e-5 observed on values of 0.4-2.6; with
    # HIFIHR_WINO_M=2 (F(2x2, 3x3)) the same comparison holds at 5e-5.  The per-convolution parity is pinned shape by shape above.
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat"], atol=2e-4, rtol=1e-4)
    ((low * torch.tensor(g["wl"]).cuda()).sum() + (feat * torch.tensor(g["wf"]).cuda()).sum()).backward()
    for key, grad in (("g_conv1", net.conv1.weight.grad), ("g_bn1", net.bn1.weight.grad),
                      ("g_l4c2", net.layer4[1].conv2.weight.grad[:8]), ("g_l2ds", net.layer2[0].downsample[0].weight.grad)):
        ref = g[key]
        err = np.abs(grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        # batch of 2 through 20 train-mode BatchNorms amplifies rounding (18 samples per channel in layer 4): the tight gradient
        # bound is held on the batch-of-8 fixture below
        assert err < 1.5e-2, (key, err)


def test_resnet18_trunk_mfma_vs_reference_golden_batch8(golden_dir):
    """Batch of 8 (tests/golden/resnet18_b8.npz: the reference's vendored ResNet executed by tools/make_golden.py:gen_resnet18_b8):
    features and low features 2e-4 absolute (1e-5 of their range observed); all fourteen stored gradients (stem, every stage, both
    kinds of shortcut, batch-norm scale and shift) within 1e-2 relative L2 / 2e-2 of their maximum.  Why not tighter: this network is
    perfectly conditioned (PyTorch fp32 against float64: gradients 2e-6), but its gradient is DISCONTINUOUS in the ReLU signs, and a
    forward that rounds differently (Winograd: 1e-5 of max |y|) moves a handful of the ~1e6 pre-activations across zero; one flipped
    unit of a 4x4 map changes that channel's gradient by ~1/128 and the batch-norm backward spreads it (measured: 1.7e-3 .. 7e-3
    of max depending on which units flip).  The arithmetic of the backward kernels themselves is held to 2e-4 by the next test,
    which imposes one ReLU pattern on both sides."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    g = np.load(os.path.join(golden_dir, "resnet18_b8.npz"))
    x, wl, wf = kc.resnet18_b8_inputs(g)
    enc = Resnet_4C("res18")
    enc.model.load_state_dict(seeded_state_dict(enc.model))
    enc = enc.cuda().train()
    low, feat = enc(ops.image_to_nhwc4(x.cuda()))
    ((low * wl.cuda()).sum() + (feat * wf.cuda()).sum()).backward()
    worst, l2 = kc.resnet18_b8_check(g, enc.model, low, feat, out_atol=2e-4, grad_rtol=2e-2, grad_l2=1e-2)
    print("resnet18 b8 gradient errors (max / max|ref|):", {k: f"{v:.1e}" for k, v in worst.items()})
    print("resnet18 b8 gradient errors (relative L2):", {k: f"{v:.1e}" for k, v in l2.items()})


def _b8_errors(golden_dir, precision="fast"):
    """The batch-of-8 trunk fixture under the CURRENT process's dispatch switches -> (worst max error, worst relative L2, feature error)."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    g = np.load(os.path.join(golden_dir, "resnet18_b8.npz"))
    x, wl, wf = kc.resnet18_b8_inputs(g)
    enc = Resnet_4C("res18")
    enc.model.load_state_dict(seeded_state_dict(enc.model))
    enc = enc.cuda().train()
    with ops.conv_precision(precision):
        low, feat = enc(ops.image_to_nhwc4(x.cuda()))
    ((low * wl.cuda()).sum() + (feat * wf.cuda()).sum()).backward()             # (outside the scope: every convolution follows its forward)
    worst, l2 = kc.resnet18_b8_check(g, enc.model, low, feat, out_atol=2e-4, grad_rtol=1.0, grad_l2=1.0)
    ferr = float(np.abs(feat.detach().cpu().numpy() - g["feat"]).max() / np.abs(g["feat"]).max())
    return max(worst.values()), max(l2.values()), ferr


def test_conv_precision_knob_feature_and_gradient_errors_of_both_dispatches(golden_dir):
    """`conv_precision("reference")` (what `Model(conv_precision="reference")` wraps its encoder in) sends the stride-1 3x3 layers to the direct
    kernels IN THIS PROCESS, per call site.  MEASURED on the batch-of-8 fixture (profiles/r04_precision_by_dispatch.txt): the features
    then agree with the reference's to 5e-6 of their maximum instead of 1.3e-5 -- but the fourteen stored gradients do NOT get closer:
    worst max error 1.1e-2 / relative L2 8.9e-3 on the direct kernels against 8.1e-3 / 7.9e-3 on the default dispatch (and 6.3e-3 with
    F(2x2) everywhere).  The gradient differences are ReLU sign flips of pre-activations within rounding distance of zero, and ANY fp32
    summation order other than the reference's own flips a handful of them -- the direct MFMA kernels' order as much as Winograd's.  What
    pins the backward arithmetic is test_resnet18_trunk_backward_given_the_same_relu_pattern (2e-4).  Both dispatches are measured here so
    that the numbers are on file with every run."""
    m_ref, l2_ref, f_ref = _b8_errors(golden_dir, "reference")
    m_fast, l2_fast, f_fast = _b8_errors(golden_dir, "fast")
    print(f"batch-of-8 trunk fixture, worst of 14 gradients vs the reference (max / max|ref|, relative L2), feature error / max: "
          f"conv_precision='reference': {m_ref:.1e}, {l2_ref:.1e}, {f_ref:.1e};  'fast' (default): {m_fast:.1e}, {l2_fast:.1e}, {f_fast:.1e}")
    # (bounds: twice the measured values -- these errors are ReLU sign flips, a handful more on another box must not turn the suite red)
    assert m_ref < 2.5e-2 and l2_ref < 2e-2 and f_ref < 2e-5, (m_ref, l2_ref, f_ref)
    assert m_fast < 2.5e-2 and l2_fast < 2e-2, (m_fast, l2_fast)


@pytest.mark.parametrize("env,max_tol,l2_tol", [({"HIFIHR_WINO_M": "2"}, 2.5e-2, 2e-2), ({"HIFIHR_WINOGRAD": "0"}, 2.5e-2, 2e-2)])
def test_trunk_gradient_error_by_dispatch_switch(golden_dir, env, max_tol, l2_tol):
    """The process-wide switches on the same fixture (a subprocess per setting: HIFIHR_WINO_M is read once by the library): F(2x2, 3x3)
    everywhere (HIFIHR_WINO_M=2, +0.6 ms/step) and no Winograd at all (HIFIHR_WINOGRAD=0, +2.9 ms/step).  Measured: 6.3e-3 / 6.5e-3 and
    1.1e-2 / 8.9e-3 (max / relative L2) -- no dispatch brings the gradients closer than ~6e-3 on this fixture; README "Precision of the
    default dispatch" holds the table."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_conv as t; "
            "print('B8ERR ' + json.dumps(t._b8_errors(%r)))" % (root, os.path.join(root, "tests"), golden_dir))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("B8ERR ")]
    assert r.returncode == 0 and line, r.stderr[-2000:]
    m, l2, f = json.loads(line[-1][6:])
    print(f"{env}: worst of 14 trunk gradients vs the reference: max / max|ref| {m:.1e}, relative L2 {l2:.1e}; features {f:.1e}")
    assert m < max_tol and l2 < l2_tol, (env, m, l2)


def test_resnet18_trunk_backward_given_the_same_relu_pattern(golden_dir):
    """Backward ARITHMETIC of the whole MFMA trunk, isolated from ReLU sign flips.  Against reference-generated gradients the trunk
    differs by up to ~5e-3 of a gradient's maximum on the batch-of-8 fixture although its activations agree to 1e-5: with ~1e6 ReLU
    inputs a forward rounding error of 1e-5 (Winograd F(4x4, 3x3); 1e-6 for the direct kernels) lands a handful of pre-activations on
    the other side of zero, and one flipped unit in a 4x4 map changes that channel's gradients by ~1 / 128 (tools/_probe/b8_bisect.py
    counts them against a float64 run).  Here the oracle trunk (float64, CPU) is run with OUR ReLU patterns imposed -- every F.relu
    becomes a multiplication by the mask the product's forward produced -- so that what is compared is the chain of backward-data /
    backward-weight / batch-norm-backward kernels itself: every parameter gradient within 2e-4 of its maximum."""
    import os, sys
    import numpy as np
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    import oracle.torch_modules as tm
    g = np.load(os.path.join(golden_dir, "resnet18_b8.npz"))
    x, wl, wf = kc.resnet18_b8_inputs(g)
    enc = Resnet_4C("res18")
    enc.model.load_state_dict(seeded_state_dict(enc.model))
    enc = enc.cuda().train()
    masks = []
    real_bn_act, real_stem = ops.bn_act, ops.bn_relu_maxpool

    def rec_bn_act(xx, stats, bn, residual=None, relu=True):
        y = real_bn_act(xx, stats, bn, residual, relu)
        if relu is True:
            masks.append((y.detach() > 0).cpu())
        return y

    def rec_stem(xx, stats, bn):
        z = F.batch_norm(xx.detach().double(), None, None, bn.weight.detach().double(), bn.bias.detach().double(), True, 0.0, bn.eps)
        masks.append((z > 0).cpu())
        return real_stem(xx, stats, bn)
    real_fused = ops.bn_act_wino_conv

    def rec_fused(xx, stats, bn, residual, w, want_stats):
        # batch-norm (+ identity) + ReLU applied inside the next convolution's input transform: the activation is not stored
        z = F.batch_norm(xx.detach().double(), None, None, bn.weight.detach().double(), bn.bias.detach().double(), True, 0.0, bn.eps)
        if residual is not None:
            z = z + residual.detach().double()
        masks.append((z > 0).cpu())
        return real_fused(xx, stats, bn, residual, w, want_stats)
    ops.bn_act, ops.bn_relu_maxpool, ops.bn_act_wino_conv = rec_bn_act, rec_stem, rec_fused
    try:
        low, feat = enc(ops.image_to_nhwc4(x.cuda()))
    finally:
        ops.bn_act, ops.bn_relu_maxpool, ops.bn_act_wino_conv = real_bn_act, real_stem, real_fused
    ((low * wl.cuda()).sum() + (feat * wf.cuda()).sum()).backward()
    assert len(masks) == 17                                    # stem + 2 per BasicBlock
    ref = tm.Resnet4CRef("res18")
    ref.model.load_state_dict(seeded_state_dict(ref.model))
    ref = ref.double().train()
    it = iter(masks)
    real_relu = tm.F.relu

    def masked_relu(t, inplace=False):
        m = next(it)
        assert m.shape == t.shape
        return t * m.to(t.dtype)
    tm.F.relu = masked_relu
    try:
        low0, feat0 = ref(tm.normalize_batch_3C(x).double())
    finally:
        tm.F.relu = real_relu
    ((low0 * wl.double()).sum() + (feat0 * wf.double()).sum()).backward()
    assert float((feat.detach().cpu().double() - feat0.detach()).abs().max()) <= 1e-4 * float(feat0.detach().abs().max())
    worst = {}
    mine = dict(enc.model.named_parameters())
    for n, p in ref.model.named_parameters():
        worst[n] = float((mine[n].grad.detach().cpu().double() - p.grad).abs().max() / p.grad.abs().max())
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    print("largest gradient errors with the ReLU pattern imposed:", sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    assert not bad, bad


def test_trunk_gradients_identical_in_every_backward_form(golden_dir):
    """The three forms the trunk's backward can take give the same gradients: (a) the unfused batch-norm backward
    (HIFIHR_BN_WINO_BWD=0), (b) reduction in the output-transform epilogue + one apply launch (plain parameters), (c) additionally the
    apply inside the producer's dual transform (parameters in a FlatParams: gradients accumulated straight into the flat buffer, which
    is what the training step runs).  Same forward bits in all three; gradients within 5e-5 of their maximum (reduction order)."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    from hifihr_amd.optim import FlatParams
    g = np.load(os.path.join(golden_dir, "resnet18_b8.npz"))
    x, wl, wf = kc.resnet18_b8_inputs(g)

    def run(form):
        enc = Resnet_4C("res18")
        enc.model.load_state_dict(seeded_state_dict(enc.model))
        enc = enc.cuda().train()
        if form == "c":
            flat = FlatParams(enc)
            flat.zero_grad()
        os.environ["HIFIHR_BN_WINO_BWD"] = "0" if form == "a" else "1"
        try:
            low, feat = enc(ops.image_to_nhwc4(x.cuda()))
            ((low * wl.cuda()).sum() + (feat * wf.cuda()).sum()).backward()
        finally:
            os.environ.pop("HIFIHR_BN_WINO_BWD", None)
        return feat.detach().clone(), {n: p.grad.detach().clone() for n, p in enc.model.named_parameters()}
    fa, ga = run("a")
    for form in ("b", "c"):
        f, gr = run(form)
        assert torch.equal(f, fa)
        worst = max((float((gr[n] - ga[n]).abs().max() / (ga[n].abs().max() + 1e-30)), n) for n in ga)
        print("form", form, "largest gradient difference from the unfused backward:", worst)
        assert worst[0] <= 5e-5, (form, worst)


@pytest.mark.parametrize("C,relu,residual,N,H", [(64, True, False, 32, 56), (128, True, True, 8, 28), (512, False, False, 32, 14), (256, True, True, 4, 14)])
def test_bn_act_kernels(lib, C, relu, residual, N, H):
    kc.bn_act_case(lib, "cuda", N, H, H, C, relu, residual, seed=C + N)


@pytest.mark.parametrize("N,H,W,C", [(32, 112, 112, 64), (3, 37, 21, 64), (2, 16, 16, 256), (5, 9, 8, 12)])
def test_bn_relu_maxpool_stem(lib, N, H, W, C):
    kc.bn_relu_maxpool_case(lib, "cuda", N, H, W, C, seed=N + C)


@pytest.mark.parametrize("producer", ["stats", "conv3x3", "conv1x1", "halo", "wino4", "wino2", "dw"])
def test_bn_statistics_with_mean_much_larger_than_std(lib, producer):
    """Round-2 review: channels with mean 50 / std 0.1.  Shifted sums per lane, fp64 slots, fp64 fold (csrc/hifihr_internal.h "FORWARD
    statistics") against float64 statistics of the same tensor, for every kernel that produces batch statistics."""
    kc.bn_large_mean_case(lib, "cuda", producer)


def test_conv_epilogue_bn_statistics(lib):
    kc.conv_bnstats_case(lib, "cuda", 8, 56, 56, 64, 64, 3, 1, 1)
    kc.conv_bnstats_case(lib, "cuda", 4, 224, 224, 4, 64, 7, 2, 3)


def test_efficientnet_b3_hip_vs_reference_golden(golden_dir):
    """EfficientNet-b3 on the hand-written kernels against the REFERENCE'S OWN outputs (tests/golden/effnet_b3_small.npz:
    `EfficientNet.from_name('efficientnet-b3').extract_features` in train mode, drop-connect active under torch.manual_seed(5),
    tools/make_golden.py:gen_effnet): the drop-connect draws are made exactly as the reference made them -- the CPU generator
    under the same seed, one `torch.rand([B, 1, 1, 1])` per skip block in block order (network/efficientnet_pt/utils.py:82-91,
    model.py:91-93) -- and handed to the product through `effnet._drop_connect_uniform`; features, low features and the five
    stored gradients are compared with the stored values."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from seeded_init import seeded_state_dict
    import hifihr_amd.effnet as E
    g = np.load(os.path.join(golden_dir, "effnet_b3_small.npz"))
    net = E.EfficientNetB3()
    net.load_state_dict(seeded_state_dict(net))
    net = net.cuda().train()
    x = torch.tensor(g["x"]).cuda()
    draws = []
    old = E._drop_connect_uniform

    def cpu_draw(t):
        u = torch.rand([t.shape[0], 1, 1, 1], dtype=t.dtype)          # the reference's draw: CPU generator, same order
        draws.append(u)
        return u.to(t.device)
    E._drop_connect_uniform = cpu_draw
    try:
        torch.manual_seed(5)
        feat, low = net.extract_features(x)
    finally:
        E._drop_connect_uniform = old
    assert len(draws) == 19                                            # b3: 19 of the 26 blocks have a skip (block 0 has none)
    feat_nchw, low_nchw = feat.contiguous(), low.contiguous()
    np.testing.assert_allclose(low_nchw.detach().cpu().numpy(), g["low"], atol=2e-4 * float(np.abs(g["low"]).max()), rtol=0)
    np.testing.assert_allclose(feat_nchw.detach().cpu().numpy(), g["feat"], atol=2e-4 * float(np.abs(g["feat"]).max()), rtol=0)
    ((feat * torch.tensor(g["wf"]).cuda()).sum() + (low * torch.tensor(g["wl"]).cuda()).sum()).backward()
    worst = {}
    for key, grad in (("g_stem", net._conv_stem.weight.grad), ("g_b3_expand", net._blocks[3]._expand_conv.weight.grad),
                      ("g_b10_dw", net._blocks[10]._depthwise_conv.weight.grad), ("g_b20_se", net._blocks[20]._se_reduce.weight.grad),
                      ("g_head_bn", net._bn1.weight.grad)):
        ref = g[key]
        got = grad.detach().cpu().numpy().reshape(ref.shape)
        worst[key] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print("effnet b3 gradient errors (relative to max):", {k: f"{v:.1e}" for k, v in worst.items()})
    assert all(v <= 2e-3 for v in worst.values()), worst


@pytest.mark.parametrize("N,H,W,C,K,R", [(1, 8, 8, 32, 128, 1), (4, 14, 14, 256, 512, 1), (2, 6, 6, 32, 128, 3), (1, 8, 8, 32, 64, 1)])
def test_conv_relu_without_bias(lib, N, H, W, C, K, R):
    """hifihr_conv2d_fwd(bias = NULL, act = 1) clamps on every dispatch path (round-2 advisor finding: the 1x1 GEMM path did not)."""
    kc.conv_relu_nobias_case(lib, "cuda", N, H, W, C, K, R, 1, seed=K, pad=R // 2)


@pytest.mark.parametrize("N,H,C,K,stride", [(32, 56, 192, 3, 1), (32, 56, 192, 5, 2), (8, 14, 816, 5, 1), (32, 7, 2304, 3, 1)])
def test_dwconv_kernels(lib, N, H, C, K, stride):
    kc.dwconv_case(lib, "cuda", N, H, H, C, K, stride, seed=C + K)


@pytest.mark.parametrize("N,H,C,K,stride", [(16, 56, 192, 3, 1), (16, 56, 192, 5, 2), (8, 14, 816, 5, 1), (32, 7, 2304, 3, 1), (4, 112, 144, 3, 2)])
def test_dwconv_with_batchnorm_and_swish_on_load(lib, N, H, C, K, stride):
    """The expand half of an MBConv block without its activated tensor (network/efficientnet_pt/model.py:73-80)."""
    kc.dwconv_bnswish_case(lib, "cuda", N, H, H, C, K, stride, seed=C + K + 1)


@pytest.mark.parametrize("N,H,C,K", [(32, 14, 512, 512), (32, 28, 128, 128), (32, 14, 256, 256), (32, 56, 64, 64)])
def test_balanced_schedule_full_batch(lib, N, H, C, K):
    """The stream-K schedule at BASELINE's batch: forward and backward-data through the shared workspace."""
    assert kc.conv_case(lib, "cuda", N, H, H, C, K, 3, 1, 1, seed=C, rtol=3e-5) == 2


def test_balanced_schedule_bnstats(lib):
    kc.conv_bnstats_case(lib, "cuda", 32, 14, 14, 512, 512, 3, 1, 1, use_ws=True)


@pytest.mark.parametrize("N,H,C,K", [(32, 14, 256, 256), (32, 14, 512, 512), (8, 28, 128, 128), (3, 9, 128, 160)])
def test_winograd_path(lib, N, H, C, K):
    """Winograd F(2x2, 3x3) forward and backward-data (the layers 2-4 path) vs torch conv2d."""
    kc.wino_case(lib, "cuda", N, H, H, C, K, seed=C + K)


@pytest.mark.parametrize("N,H,C,K", [(32, 14, 256, 256), (32, 14, 512, 512), (32, 28, 128, 128), (32, 14, 256, 512), (3, 9, 128, 192), (48, 56, 128, 128)])
def test_winograd_f4_path(lib, N, H, C, K):
    """Winograd F(4x4, 3x3) (csrc/wino4.hip, the default of layers 2-4): forward (+ BN statistics, bias / ReLU), backward-data, the dual
    dy transform, slab backward-weight vs torch conv2d at the bench's sizes, a size that is not a multiple of 4 and a VGG-sized map."""
    if os.environ.get("HIFIHR_WINO_M") == "2":
        pytest.skip("HIFIHR_WINO_M=2 selects F(2x2, 3x3) for every layer")
    assert lib.wino_tile(N, H, H, C, K) == 4
    kc.wino_case(lib, "cuda", N, H, H, C, K, seed=C + K + H, m=4)


def test_weight_prep_equals_separate_transforms(lib):
    kc.weight_prep_case(lib, "cuda")


_RES50_SHAPES = [  # (H, C, K, R, stride, pad): every distinct convolution of the bottleneck trunk (layer-4 strides 1) except the stem
    (56, 64, 64, 1, 1, 0), (56, 64, 64, 3, 1, 1), (56, 64, 256, 1, 1, 0), (56, 256, 64, 1, 1, 0), (56, 256, 128, 1, 1, 0),
    (56, 128, 128, 3, 2, 1), (28, 128, 512, 1, 1, 0), (56, 256, 512, 1, 2, 0), (28, 512, 128, 1, 1, 0), (28, 128, 128, 3, 1, 1),
    (28, 512, 256, 1, 1, 0), (28, 256, 256, 3, 2, 1), (14, 256, 1024, 1, 1, 0), (28, 512, 1024, 1, 2, 0), (14, 1024, 256, 1, 1, 0),
    (14, 256, 256, 3, 1, 1), (14, 1024, 512, 1, 1, 0), (14, 512, 512, 3, 1, 1), (14, 512, 2048, 1, 1, 0), (14, 1024, 2048, 1, 1, 0),
    (14, 2048, 512, 1, 1, 0)]


@pytest.mark.parametrize("H,C,K,R,stride,pad", _RES50_SHAPES)
def test_resnet50_convolution_shapes(lib, H, C, K, R, stride, pad):
    """Forward, backward-data, backward-weight of every ResNet-50 / -101 convolution shape against F.conv2d (kernel level: the
    trunk-level comparison below can only be loose, see there)."""
    kc.conv_case(lib, "cuda", 2, H, H, C, K, R, stride, pad, seed=H + C + K, rtol=3e-5)


def test_resnet50_trunk_mfma_matches_torch_restatement():
    """The bottleneck trunk of the reference's res50 / res101 encoders (torchvision v1.5 bottleneck, layer-4 stride edits) on the
    hand-written kernels vs the torch restatement (oracle/torch_modules.py) with the same weights.  Train mode: features and low-level features.  Gradients
    are compared with batch-norm in eval mode and loosely: a randomly initialised 50-layer trunk is chaotic in train mode
    (perturbing the input of the torch flavour by 1e-7 moves its own layer-4 gradients by 10-20 %, tools/debug_res50.py) and
    still amplifies ReLU-boundary flips in eval mode; the kernels themselves are pinned shape by shape above."""
    from hifihr_amd.network import ResEncoder
    from oracle.torch_modules import ResEncoderRef
    torch.manual_seed(5)
    ref = ResEncoderRef(pretrain="res50").train()
    hip = ResEncoder(pretrain="res50").cuda().train()
    hip.load_state_dict(ref.state_dict())
    x = torch.rand(4, 3, 96, 96)
    with torch.no_grad():
        low_r, f_r = ref(x)
        low_h, f_h = hip(x.cuda())
    assert tuple(f_h.shape) == (4, 2048) and tuple(low_h.shape) == (4, 512, 12, 12)
    assert float((f_h.cpu() - f_r).abs().max()) <= 2e-3 * float(f_r.abs().max())
    assert float((low_h.cpu() - low_r).abs().max()) <= 2e-3 * float(low_r.abs().max())
    # same running statistics on both sides for the eval-mode pass: the HIP train-mode forward sums its batch statistics with float
    # atomics (run-to-run differences in the last bits), which this chaotic 50-layer trunk would amplify into the gradient check below
    hip.load_state_dict(ref.state_dict())
    ref.eval(); hip.eval()
    low_r, f_r = ref(x)
    low_h, f_h = hip(x.cuda())
    assert float((f_h.detach().cpu() - f_r.detach()).abs().max()) <= 1e-3 * float(f_r.detach().abs().max())
    gen = torch.Generator().manual_seed(1)
    wf, wl = torch.randn(f_r.shape, generator=gen), torch.randn(low_r.shape, generator=gen)
    ((f_r * wf).sum() + 0.1 * (low_r * wl).sum()).backward()
    ((f_h * wf.cuda()).sum() + 0.1 * (low_h * wl.cuda()).sum()).backward()
    worst = (0.0, "")
    for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        if q.grad is None or float(q.grad.abs().max()) < 1e-12:
            continue
        worst = max(worst, (float((p.grad.cpu() - q.grad).abs().max()) / float(q.grad.abs().max()), n))
    assert worst[0] < 5e-2, worst


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(1, 8, 14), (2, 12, 28), (32, 56, 56), (3, 2, 14), (2, 30, 42), (1, 224, 224), (1, 64, 512), (2, 20, 36)])
def test_conv_wino2_kernel(lib, N, H, W):
    """conv_wino2_kernel (64 -> 64, 3x3 / stride 1 as Winograd F(2x2, 3x3) with the transforms in registers): forward + statistics and
    backward-data vs F.conv2d at layer 1's size, VGG19 conv1_2's and the tile shapes of tests/test_hostsim_conv.py."""
    if os.environ.get("HIFIHR_CONV_WINO2") == "0":
        pytest.skip("HIFIHR_CONV_WINO2=0 switches the entry point off")
    kc.conv_wino2_case(lib, "cuda", N, H, W, seed=H + W)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,epi", [(32, 56, 56, False), (48, 224, 224, True), (16, 512, 512, True)])
def test_conv_wino2_agrees_with_the_direct_kernel_at_config_sizes(lib, N, H, W, epi):
    """The one-launch Winograd kernel against the direct halo kernel on the SAME inputs at BASELINE's sizes (layer 1 at batch 32; VGG19 conv1_2
    of the perceptual loss at 48 x 224^2 and, with a ragged last column tile, at 16 x 512^2), plus linearity in the input: sizes at which
    the CPU oracle is not run."""
    import torch
    if os.environ.get("HIFIHR_CONV_WINO2") == "0":
        pytest.skip("HIFIHR_CONV_WINO2=0 switches the entry point off")
    g = torch.Generator(device="cuda").manual_seed(N + H)
    x = torch.randn(N, H, W, 64, device="cuda", generator=g) + 0.3
    x2 = torch.randn(N, H, W, 64, device="cuda", generator=g)
    w = torch.randn(64, 3, 3, 64, device="cuda", generator=g) / 24.0
    b = torch.randn(64, device="cuda", generator=g) * 0.3 if epi else None
    U = torch.empty(16 * 64 * 64, device="cuda")
    lib.wino_weight_transform(w, U, 64, 64, 0)
    y_direct = torch.empty(N, H, W, 64, device="cuda"); y_wino = torch.empty_like(y_direct)
    lib.conv2d_fwd(x, w, b, y_direct, N, H, W, 64, 64, 3, 3, 1, 1, act=1 if epi else 0)
    lib.conv3x3_c64_wino(x, U, b, epi, y_wino, None, N, H, W)
    scale = float(y_direct.abs().max())
    assert float((y_wino - y_direct).abs().max()) <= 2e-5 * scale
    # linearity (without the epilogue): conv(x + x2) = conv(x) + conv(x2)
    ya = torch.empty_like(y_wino); yb = torch.empty_like(y_wino); yc = torch.empty_like(y_wino)
    lib.conv3x3_c64_wino(x, U, None, False, ya, None, N, H, W)
    lib.conv3x3_c64_wino(x2, U, None, False, yb, None, N, H, W)
    lib.conv3x3_c64_wino((x + x2).contiguous(), U, None, False, yc, None, N, H, W)
    assert float((yc - ya - yb).abs().max()) <= 2e-5 * float(yc.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(4, 224, 224), (2, 512, 512)])
def test_conv_wino2_vs_torch_conv2d_at_config_resolutions(lib, N, H, W):
    """The one-launch Winograd kernel against torch's own F.conv2d (CPU, fp32) at the resolutions of BASELINE configs[2] / [4] (VGG19
    conv1_2: 224^2 and 512^2 with its ragged last column tile) on a few images -- the config-size test above compares two kernels of this
    repository; this one pins the same code path to an independent implementation.  3e-5 of max |y| (F(2x2, 3x3) rounds at ~1e-6)."""
    import torch
    import torch.nn.functional as F
    if os.environ.get("HIFIHR_CONV_WINO2") == "0":
        pytest.skip("HIFIHR_CONV_WINO2=0 switches the entry point off")
    g = torch.Generator().manual_seed(H + N)
    x = torch.randn(N, 64, H, W, generator=g) + 0.3
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g) * 0.3
    ref = torch.relu(F.conv2d(x, w, b, 1, 1)).permute(0, 2, 3, 1).contiguous()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda(); wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    U = torch.empty(16 * 64 * 64, device="cuda")
    lib.wino_weight_transform(wd, U, 64, 64, 0)
    y = torch.empty(N, H, W, 64, device="cuda")
    lib.conv3x3_c64_wino(xd, U, b.cuda(), True, y, None, N, H, W)
    err = float((y.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 3e-5, err


@pytest.mark.gpu
def test_conv_wino2_bias_relu_epilogue(lib):
    if os.environ.get("HIFIHR_CONV_WINO2") == "0":
        pytest.skip("HIFIHR_CONV_WINO2=0 switches the entry point off")
    kc.conv_wino2_case(lib, "cuda", 2, 28, 28, seed=3, bias_relu=True)


@pytest.mark.parametrize("N,H,W,res", [(32, 56, 56, True), (32, 56, 56, False), (5, 28, 42, False), (1, 6, 14, True)])
def test_conv_c64_bwd_pair_equals_the_separate_launches(lib, N, H, W, res):
    """ResNet layer 1's backward at the config batch (and ragged shares): data gradient + weight gradient in one launch."""
    kc.conv_c64_bwd_pair_case(lib, "cuda", N, H, W, seed=N + W, with_res=res)


@pytest.mark.parametrize("B,H,W", [(32, 56, 56), (5, 28, 42)])
def test_halo_kernels_are_bit_reproducible(lib, B, H, W):
    """conv_halo_kernel / conv_halo_wgrad_kernel hold no atomics on their outputs: repeated launches on the same inputs agree bit for bit
    (a loader / MFMA-wave synchronisation bug would show as run-to-run differences), and they match torch on the CPU."""
    torch.manual_seed(B)
    x = torch.randn(B, H, W, 64, device="cuda"); w = torch.randn(64, 3, 3, 64, device="cuda") / 24.0; gy = torch.randn(B, H, W, 64, device="cuda")
    scratch = torch.empty(64 * 9 * 64, device="cuda")
    outs = []
    for _ in range(6):
        out = torch.empty(B, H, W, 64, device="cuda"); dx = torch.empty_like(out); dw = torch.zeros(64, 3, 3, 64, device="cuda")
        lib.conv2d_fwd(x, w, None, out, B, H, W, 64, 64, 3, 3, 1, 1)
        lib.conv2d_bwd_data(gy, w, dx, scratch, B, H, W, 64, 64, 3, 3, 1, 1)
        lib.conv2d_bwd_weight(x, gy, dw, B, H, W, 64, 64, 3, 3, 1, 1)
        outs.append((out, dx, dw))
    torch.cuda.synchronize()
    assert all(torch.equal(o[k], outs[0][k]) for o in outs for k in range(3))
    xr = x.cpu().permute(0, 3, 1, 2).requires_grad_(True); wr = w.cpu().permute(0, 3, 1, 2).requires_grad_(True)
    y = torch.nn.functional.conv2d(xr, wr, None, 1, 1)
    y.backward(gy.cpu().permute(0, 3, 1, 2))
    rel = lambda a, b: float((a.cpu() - b).abs().max()) / float(b.abs().max())
    assert rel(outs[0][0], y.detach().permute(0, 2, 3, 1)) < 3e-5 and rel(outs[0][1], xr.grad.permute(0, 2, 3, 1)) < 3e-5
    assert rel(outs[0][2], wr.grad.permute(0, 2, 3, 1)) < 2e-4


@pytest.mark.parametrize("N,H,C,K", [(32, 14, 256, 512), (8, 56, 256, 128), (4, 28, 512, 128), (2, 7, 2048, 512)])
def test_conv_1x1_runs_on_the_gemm_kernels(lib, N, H, C, K):
    assert lib.conv2d_describe(N, H, H, C, K, 1, 1, 1, 0, 0) .startswith("bgemm_nt_rows_kernel<")
    assert lib.conv2d_describe(N, H, H, C, K, 1, 1, 1, 0, 2).startswith("bgemm_")
    kc.conv_case(lib, "cuda", N, H, H, C, K, 1, 1, 0, seed=C + K)
    kc.conv_bnstats_case(lib, "cuda", N, H, H, C, K, 1, 1, 0)


@pytest.mark.parametrize("N,H,C,K", [(48, 7, 1392, 384), (48, 7, 232, 1392), (48, 14, 136, 816), (8, 14, 144, 240)])
def test_conv_1x1_ragged_channels_on_the_gemm_kernel(lib, N, H, C, K):
    """EfficientNet-b3's 1x1 convolutions whose channel counts are multiples of 4 but not of 32 / 128, at the configs[2] batch:
    bgemm_nt_rows_kernel<RAGGED> (zero-page operand segments past row N / column K, masked epilogue and statistics)."""
    assert lib.conv2d_describe(N, H, H, C, K, 1, 1, 1, 0, 0) .startswith("bgemm_nt_rows_kernel<")
    kc.conv_case(lib, "cuda", N, H, H, C, K, 1, 1, 0, seed=C + K)
    kc.conv_bnstats_case(lib, "cuda", N, H, H, C, K, 1, 1, 0)


def test_conv_stem_wgrad_three_channel_parameter(lib):
    kc.stem_c3_wgrad_case(lib, "cuda", N=32, H=224)
    kc.stem_c3_wgrad_case(lib, "cuda", N=3, H=112, seed=2)


def test_stem_kernels_at_batch_32(lib):
    """conv_stem_kernel / conv_stem_wgrad_kernel at the bench's size (32 x 224^2), the padding channel zero as in the encoder."""
    assert lib.conv2d_describe(32, 224, 224, 4, 64, 7, 7, 2, 3, 0) == "conv_stem_kernel"
    assert lib.conv2d_describe(32, 224, 224, 4, 64, 7, 7, 2, 3, 2) == "conv_stem_wgrad_kernel"
    kc.conv_case(lib, "cuda", 32, 224, 224, 4, 64, 7, 2, 3, seed=3, zero_last_channel=True, rtol=3e-5)


def test_conv_fork_residual_sum_equals_autograd_sum(monkeypatch):
    """ops._Conv2dMFMA(fork=True): the gradient of a block input's second consumer added inside the backward-data launch (conv_wino2_kernel /
    conv_igemm_kernel epilogues) gives the SAME parameter gradients as autograd's own elementwise sum (HIFIHR_CONV_FORK=0): same operands,
    one rounding either way."""
    import torch
    from hifihr_amd import ops
    from hifihr_amd.network import Resnet_4C
    torch.manual_seed(3)
    x = torch.randn(4, 3, 224, 224)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HIFIHR_CONV_FORK", mode)
        torch.manual_seed(5)
        enc = Resnet_4C("res18").cuda().train()
        low, feat = enc(ops.image_to_nhwc4(x.cuda()))
        (low.square().mean() + feat.square().mean()).backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None}
    assert grads["1"].keys() == grads["0"].keys() and len(grads["1"]) > 40
    for n in grads["1"]:
        a, b = grads["1"][n], grads["0"][n]
        # (not bit-identical: the batch-norm reductions and weight gradients in between use float atomics, whose order differs run to run --
        #  ~1e-5 of a gradient's maximum between two runs of the SAME mode; a wrong residual sum would be O(1))
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-12, n


def test_weight_gradient_transforms_of_several_layers_in_one_launch(lib):
    """hifihr_wino4_dw_transform_multi (the step's deferred F(4x4) weight-gradient transforms, ops._DeferredDw) == the per-layer launches."""
    kc.wino4_dw_multi_case(lib, "cuda")


@pytest.mark.parametrize("N,H,C,K1,K2", [(32, 56, 64, 128, 128), (32, 28, 128, 256, 256), (3, 30, 32, 128, 256)])
def test_strided_conv_and_downsample_conv_in_one_launch(lib, N, H, C, K1, K2):
    """hifihr_conv2d_fwd_bnstats_pair: conv1 (3x3 stride 2) + downsample[0] (1x1 stride 2) of layer2.0 / layer3.0 at B = 32, and a ragged size."""
    kc.conv_fwd_pair_case(lib, "cuda", N, H, H, C, K1, K2, seed=H + C)


@pytest.mark.parametrize("N,H,C,K", [(32, 56, 64, 128), (32, 28, 128, 256), (3, 31, 32, 48)])
def test_strided_dgrad_with_the_downsample_1x1_as_a_tap(lib, N, H, C, K):
    """hifihr_conv2d_bwd_data_pre_plus1x1: the backward-data of layer2.0 / layer3.0's conv1 with downsample[0]'s as a tap of parity class
    (0, 0), at B = 32 and at an odd size."""
    kc.conv_dgrad_plus1x1_case(lib, "cuda", N, H, H, C, K, seed=H + C)


@pytest.mark.parametrize("N,H,C,K", [(32, 56, 64, 128), (32, 28, 128, 256), (3, 31, 32, 48)])
def test_strided_wgrad_with_the_downsample_1x1_in_the_same_launch(lib, N, H, C, K):
    """hifihr_conv2d_bwd_weight_plus1x1: the weight gradients of layer2.0 / layer3.0's conv1 and downsample[0] in one launch, at B = 32 and odd."""
    kc.conv_wgrad_plus1x1_case(lib, "cuda", N, H, H, C, K, seed=H + C + 1)
