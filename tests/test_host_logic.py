"""CPU tests of the host-side mirror of the reference interface (pure torch code, no HIP): loss formulas and
projection helpers against vectors produced by executing the reference's own function sources
(tools/make_golden.py gen_losses / gen_ssim / gen_resnet18), config handling, synthetic-table loader."""
import os
import sys

import numpy as np
import torch

from hifihr_amd import losses as L
from hifihr_amd import options
from hifihr_amd.network import ResNet18Trunk, normalize_batch_3C
from hifihr_amd.traineval import proj_func

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_loss_helpers_vs_reference(golden_dir, synth_tables):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    t = lambda k: torch.tensor(g[k])
    np.testing.assert_allclose(L.bone_direction_loss(t("j"), t("jg")).numpy(), g["bone3d"], rtol=1e-5)
    np.testing.assert_allclose(L.bone_direction_loss(t("j2"), t("j2g")).numpy(), g["bone2d"], rtol=1e-5)
    faces = torch.as_tensor(synth_tables.faces.astype(np.int16)).unsqueeze(0)
    np.testing.assert_allclose(L.edge_length_loss(t("v"), t("vg"), faces).numpy(), g["edge"], rtol=1e-5)
    np.testing.assert_allclose(L.iou(t("m1"), t("m2")).numpy(), g["iou"], rtol=1e-6)
    np.testing.assert_allclose(proj_func(t("xyz"), t("K")).numpy(), g["proj"], rtol=1e-5, atol=1e-4)


def test_ssim_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    a = torch.tensor(g["a"], requires_grad=True)
    val = L.ssim(a, torch.tensor(g["b"]))
    val.backward()
    np.testing.assert_allclose(val.item(), g["ssim"], rtol=1e-6)
    np.testing.assert_allclose(a.grad.numpy(), g["ga"], atol=1e-9, rtol=1e-4)
    gen = torch.Generator().manual_seed(12)
    A = torch.rand(2, 3, 224, 224, generator=gen); B = torch.rand(2, 3, 224, 224, generator=gen)
    np.testing.assert_allclose(L.ssim(A, B).item(), g["ssim224"], rtol=1e-5)


def test_resnet18_trunk_vs_reference(golden_dir):
    """This package's ResNet-18 trunk (same state-dict names as torchvision) reproduces the reference's vendored
    ResNet with the layer4 stride edits, forward and backward, from name-seeded weights."""
    from seeded_init import seeded_state_dict
    g = np.load(os.path.join(golden_dir, "resnet18_small.npz"))
    net = ResNet18Trunk(layer4_stride=1)
    sd = seeded_state_dict(net)
    net.load_state_dict(sd)
    net.train()
    x = normalize_batch_3C(torch.tensor(g["x"]))
    h = net.maxpool(net.relu(net.bn1(net.conv1(x))))
    low = net.layer2(net.layer1(h))
    feat = net.layer4(net.layer3(low))
    np.testing.assert_allclose(low.detach().numpy(), g["low"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(feat.detach().numpy(), g["feat"], atol=2e-5, rtol=1e-4)
    ((low * torch.tensor(g["wl"])).sum() + (feat * torch.tensor(g["wf"])).sum()).backward()
    for key, grad in (("g_conv1", net.conv1.weight.grad), ("g_bn1", net.bn1.weight.grad),
                      ("g_l4c2", net.layer4[1].conv2.weight.grad[:8]), ("g_l2ds", net.layer2[0].downsample[0].weight.grad)):
        ref = g[key]
        assert np.abs(grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-6, key


def test_options_json_overlay_and_lambda_schedules(tmp_path):
    p = tmp_path / "c.json"
    p.write_text('{"lambda_pose_list": [0.01, 0.001, 0.00001], "lambda_pose_steps": [10, 20], "losses": ["joint_3d"], "zzz": 1}')
    a = options.make_args(str(p))
    assert a.lambda_pose == 0.01 and a.losses == ["joint_3d"] and a.zzz == 1
    assert options.update_lambdas_for_epoch(a, 9).lambda_pose == 0.01
    assert options.update_lambdas_for_epoch(a, 10).lambda_pose == 0.001
    assert options.update_lambdas_for_epoch(a, 25).lambda_pose == 0.00001
    b = options.baseline_config2_args()
    assert b.lambda_j3d == 200 and b.lambda_texture == 0.02 and b.train_batch == 32


def test_product_ops_refuse_cpu_tensors():
    """The hot path has no CPU fallback: a CPU tensor must raise, not silently compute."""
    import pytest
    from hifihr_amd._lib import HifihrError, require_cuda
    with pytest.raises(HifihrError):
        require_cuda(torch.zeros(3))
    # the host-side mirror of the reference modules dispatches by construction (impl / conv_impl / fused), never by device:
    # the HIP flavours refuse CPU tensors as well
    from hifihr_amd import options
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.network import HandEncoder, MMPool
    with pytest.raises(HifihrError):
        HandEncoder("mano", [10, 48, None], in_dim=512, impl="hip").train()(torch.zeros(4, 512))
    with pytest.raises(HifihrError):
        MMPool((1, 1), impl="hip")(torch.zeros(2, 8, 3, 3))
    args = options.baseline_config2_args(train_batch=2)
    outs = {"joints": torch.zeros(2, 21, 3), "mano_verts": torch.zeros(2, 778, 3), "shape_params": torch.zeros(2, 10),
            "pose_params": torch.zeros(2, 48), "mano_faces": torch.zeros(2, 4, 3, dtype=torch.int16)}
    ex = {"joints": torch.zeros(2, 21, 3), "verts": torch.zeros(2, 778, 3)}
    with pytest.raises(HifihrError):
        LossFunction(fused=True)(ex, outs, ["joint_3d", "vert_3d", "mshape", "mpose"], "FreiHand", args)


def test_efficientnet_b3_mirror_vs_reference(golden_dir):
    """hifihr_amd.effnet.EfficientNetB3 (torch path) reproduces the reference's EfficientNet-b3 extract_features
    (train mode, drop-connect under the same seed) forward and backward from name-seeded weights."""
    from seeded_init import seeded_state_dict
    from hifihr_amd.effnet import EfficientNetB3, b3_block_table
    g = np.load(os.path.join(golden_dir, "effnet_b3_small.npz"))
    tbl = b3_block_table()
    assert [t[4] for t in tbl][:6] == [24, 24, 32, 32, 32, 48] and tbl[-1][4] == 384 and len(tbl) == 26
    net = EfficientNetB3("aten")
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"])
    net.load_state_dict(seeded_state_dict(net))
    net.train()
    torch.manual_seed(5)
    feat, low = net.extract_features(torch.tensor(g["x"]))
    np.testing.assert_allclose(feat.detach().numpy(), g["feat"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(low.detach().numpy(), g["low"], atol=2e-5, rtol=1e-4)
    ((feat * torch.tensor(g["wf"])).sum() + (low * torch.tensor(g["wl"])).sum()).backward()
    for key, grad in (("g_stem", net._conv_stem.weight.grad), ("g_b3_expand", net._blocks[3]._expand_conv.weight.grad),
                      ("g_b10_dw", net._blocks[10]._depthwise_conv.weight.grad), ("g_b20_se", net._blocks[20]._se_reduce.weight.grad),
                      ("g_head_bn", net._bn1.weight.grad)):
        ref = g[key]
        assert np.abs(grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, key
