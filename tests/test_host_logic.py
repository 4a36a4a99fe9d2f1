"""CPU tests: the oracle's torch restatements (oracle/) and the product's host-side helpers against vectors produced by executing
the reference's own sources (tools/make_golden.py gen_losses / gen_ssim / gen_resnet18 / gen_effnet), config handling, that the
product refuses CPU tensors, and that nothing under oracle/ imports the product."""
import os
import sys

import numpy as np
import torch

from hifihr_amd import losses as L
from hifihr_amd import options
from hifihr_amd.traineval import proj_func
from oracle import loss_oracle as LO
from oracle.torch_modules import normalize_batch_3C

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
    # the oracle's independent restatements of the same helpers
    con = torch.ones(t("j").shape[0], 21, 1)
    np.testing.assert_allclose(LO.bone_direction_loss(t("j"), t("jg"), con).numpy(), g["bone3d"], rtol=1e-5)
    np.testing.assert_allclose(LO.bone_direction_loss(t("j2"), t("j2g"), con).numpy(), g["bone2d"], rtol=1e-5)
    np.testing.assert_allclose(LO.edge_length_loss(t("v"), t("vg"), faces).numpy(), g["edge"], rtol=1e-5)
    np.testing.assert_allclose(LO.iou(t("m1"), t("m2")).numpy(), g["iou"], rtol=1e-6)
    np.testing.assert_allclose(LO.proj_func(t("xyz"), t("K")).numpy(), g["proj"], rtol=1e-5, atol=1e-4)


def test_ssim_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    a = torch.tensor(g["a"], requires_grad=True)
    val = LO.ssim(a, torch.tensor(g["b"]))
    val.backward()
    np.testing.assert_allclose(val.item(), g["ssim"], rtol=1e-6)
    np.testing.assert_allclose(a.grad.numpy(), g["ga"], atol=1e-9, rtol=1e-4)
    gen = torch.Generator().manual_seed(12)
    A = torch.rand(2, 3, 224, 224, generator=gen); B = torch.rand(2, 3, 224, 224, generator=gen)
    np.testing.assert_allclose(LO.ssim(A, B).item(), g["ssim224"], rtol=1e-5)


def test_resnet18_trunk_vs_reference(golden_dir):
    """The oracle's ResNet-18 trunk restatement (same state-dict names as torchvision) reproduces the reference's vendored
    ResNet with the layer4 stride edits, forward and backward, from name-seeded weights (the product trunk is checked against
    the same vectors on the GPU: tests/test_gpu_conv.py)."""
    from seeded_init import seeded_state_dict
    from oracle.torch_modules import Resnet4CRef
    g = np.load(os.path.join(golden_dir, "resnet18_small.npz"))
    enc = Resnet4CRef("res18")
    net = enc.model
    sd = seeded_state_dict(net)
    net.load_state_dict(sd)
    enc.train()
    low, feat = enc(normalize_batch_3C(torch.tensor(g["x"])))
    np.testing.assert_allclose(low.detach().numpy(), g["low"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(feat.detach().numpy(), g["feat"], atol=2e-5, rtol=1e-4)
    ((low * torch.tensor(g["wl"])).sum() + (feat * torch.tensor(g["wf"])).sum()).backward()
    for key, grad in (("g_conv1", net.conv1.weight.grad), ("g_bn1", net.bn1.weight.grad),
                      ("g_l4c2", net.layer4[1].conv2.weight.grad[:8]), ("g_l2ds", net.layer2[0].downsample[0].weight.grad)):
        ref = g[key]
        assert np.abs(grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-6, key


def test_resnet18_trunk_vs_reference_batch8(golden_dir):
    """The same restatement against the batch-of-8 fixture (fourteen gradients incl. both batch-norm parameters of the stem)."""
    import kernel_cases as kc
    from seeded_init import seeded_state_dict
    from oracle.torch_modules import Resnet4CRef
    g = np.load(os.path.join(golden_dir, "resnet18_b8.npz"))
    x, wl, wf = kc.resnet18_b8_inputs(g)
    enc = Resnet4CRef("res18")
    enc.model.load_state_dict(seeded_state_dict(enc.model))
    enc.train()
    low, feat = enc(normalize_batch_3C(x))
    ((low * wl).sum() + (feat * wf).sum()).backward()
    kc.resnet18_b8_check(g, enc.model, low, feat, out_atol=2e-5, grad_rtol=2e-4, grad_l2=2e-4)


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
    # the product modules have ONE path (the hand-written kernels): every one of them refuses CPU tensors
    from hifihr_amd import options
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.network import HandEncoder, LightEstimator, MMPool, ResEncoder
    from hifihr_amd.perceptual import PerceptualLoss
    with pytest.raises(HifihrError):
        HandEncoder("mano", [10, 48, None], in_dim=512).train()(torch.zeros(4, 512))
    with pytest.raises(HifihrError):
        MMPool((1, 1))(torch.zeros(2, 8, 3, 3))
    with pytest.raises(HifihrError):
        ResEncoder("res18")(torch.zeros(1, 3, 32, 32))
    with pytest.raises(HifihrError):
        LightEstimator(128)(torch.zeros(1, 128, 28, 28))
    with pytest.raises(HifihrError):
        PerceptualLoss()(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
    args = options.baseline_config2_args(train_batch=2)
    outs = {"joints": torch.zeros(2, 21, 3), "mano_verts": torch.zeros(2, 778, 3), "shape_params": torch.zeros(2, 10),
            "pose_params": torch.zeros(2, 48), "mano_faces": torch.zeros(2, 4, 3, dtype=torch.int16)}
    ex = {"joints": torch.zeros(2, 21, 3), "verts": torch.zeros(2, 778, 3)}
    with pytest.raises(HifihrError):
        LossFunction()(ex, outs, ["joint_3d", "vert_3d", "mshape", "mpose"], "FreiHand", args)


def test_oracle_is_independent_of_the_product():
    """Nothing under oracle/ imports hifihr_amd (the checker must not share code with what it checks), and the one data
    definition both sides carry -- the declared NIMBLE texture stand-in basis -- holds the same numbers."""
    import ast
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    for fn in sorted(os.listdir(root)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(root, fn)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            assert not any(m.split(".")[0] == "hifihr_amd" for m in mods), (fn, mods)
    from hifihr_amd.models import texture_stand_in_basis as mine
    from oracle.model_oracle import texture_stand_in_basis as theirs
    assert torch.equal(mine(10), theirs(10))


def test_step_path_calls_no_library_gemm():
    """The modules the training step runs through contain no GEMM call that would dispatch to rocBLAS / hipBLASLt (torch.bmm, matmul,
    mm, einsum, the @ operator, F.linear / nn.Linear.forward, F.conv2d): every matrix product of the step is a hand-written kernel
    behind the C-ABI (round 1 ended with the Winograd GEMMs on torch.bmm).  Checked on the syntax tree, so comments and docstrings
    do not count.  hifihr_amd/synth.py (the synthetic-data generator, runs before the timed region) is not on the step's path."""
    import ast
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hifihr_amd")
    banned = {"bmm", "matmul", "mm", "einsum", "baddbmm", "addmm", "linear", "conv2d", "tensordot"}
    for fn in ("models.py", "network.py", "effnet.py", "perceptual.py", "losses.py", "ops.py", "traineval.py", "optim.py", "dist.py", "data.py"):
        tree = ast.parse(open(os.path.join(root, fn)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.BinOp) and isinstance(node.op, ast.MatMult):
                raise AssertionError(f"{fn}:{node.lineno}: '@' matrix product")
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in banned:
                owner = node.func.value
                name = owner.id if isinstance(owner, ast.Name) else getattr(owner, "attr", "")
                if name == "np":          # 3x3 host-side numpy products of the augmentation parameters: not a device GEMM
                    continue
                if name in ("torch", "F", "functional") or node.func.attr in ("bmm", "matmul", "mm", "einsum"):
                    raise AssertionError(f"{fn}:{node.lineno}: {name}.{node.func.attr}(...)")


def test_efficientnet_b3_mirror_vs_reference(golden_dir):
    """The oracle's EfficientNet-b3 restatement reproduces the reference's extract_features (train mode, drop-connect under
    the same seed) forward and backward from name-seeded weights; the product's block table matches SURVEY Appendix A."""
    from seeded_init import seeded_state_dict
    from hifihr_amd.effnet import b3_block_table
    from oracle.torch_modules import EfficientNetB3Ref
    g = np.load(os.path.join(golden_dir, "effnet_b3_small.npz"))
    tbl = b3_block_table()
    assert [t[4] for t in tbl][:6] == [24, 24, 32, 32, 32, 48] and tbl[-1][4] == 384 and len(tbl) == 26
    net = EfficientNetB3Ref()
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


def test_perceptual_loss_layout_and_loss_term():
    """utils/perceptual_loss.py:27-45 cannot be imported (torchvision is absent): pin the restatement by known answers --
    torchvision's VGG19 `features` indices / channel widths up to layer 14, the state-dict names, and the loss term of
    losses.py:392-396 (fake = re_img * seg + imgs * (1 - seg))."""
    from types import SimpleNamespace
    from hifihr_amd.perceptual import PerceptualLoss, vgg19_feature_layout
    lay = vgg19_feature_layout(14)
    assert [(i, k) for i, k, _, _ in lay if k != "relu"] == [(0, "conv"), (2, "conv"), (4, "pool"), (5, "conv"), (7, "conv"), (9, "pool"),
                                                           (10, "conv"), (12, "conv"), (14, "conv")]
    assert [(ci, co) for _, k, ci, co in lay if k == "conv"] == [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256)]
    assert lay[-1][1] == "conv"                               # features[:15] ends on conv3_3 WITHOUT its ReLU
    from oracle.torch_modules import PerceptualLossRef
    pl = PerceptualLoss(seed=1)                                # product module: parameters / names / loader (its forward needs a GPU)
    assert sum(p.numel() for p in pl.parameters()) == 1_735_488 and not any(p.requires_grad for p in pl.parameters())
    sd = {"features." + k: v + 0.01 for k, v in pl.model.state_dict().items()}
    sd["classifier.0.weight"] = torch.zeros(1)                # whole-model state dicts carry more than the features
    pl2 = PerceptualLoss(seed=2)
    pl2.load_vgg19_features(sd)
    assert torch.equal(pl2.model[14].bias, pl.model[14].bias + 0.01)
    ref = PerceptualLossRef(seed=1)                            # the torch restatement draws the same seeded weights, same names
    assert list(ref.model.state_dict().keys()) == list(pl.model.state_dict().keys())
    for (k, a), (_, b) in zip(ref.model.state_dict().items(), pl.model.state_dict().items()):
        assert torch.equal(a, b), k
    gen = torch.Generator().manual_seed(0)
    imgs, re_img = torch.rand(2, 3, 32, 32, generator=gen), torch.rand(2, 3, 32, 32, generator=gen)
    seg = (torch.rand(2, 32, 32, generator=gen) > 0.5).long()
    args = SimpleNamespace(lambda_percep=0.5, base_loss_fn="L2")
    dic = LO.LossFunctionRef(perceptual=ref)({"imgs": imgs, "segms_gt": seg}, {"re_img": re_img}, ["perceptual"], "FreiHand", args)
    s = seg.unsqueeze(1)
    norm = lambda t: (t - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    want = 0.5 * torch.nn.functional.mse_loss(ref.model(norm(re_img * s + imgs * (1 - s))), ref.model(norm(imgs)))
    assert abs(dic["perceptual"].item() - want.item()) <= 1e-7
    # identical inputs: exactly zero
    assert ref(imgs, imgs).item() == 0.0


def test_data_dic_ho3d_branch_known_answers():
    """utils/traineval_util.py:156-201 + utils/fh_utils.py:604-629 (not importable: skimage): the joint permutation table
    as written there, the (1, -1, -1) flips, and the round trip synthetic FreiHAND batch -> HO-3D conventions -> data_dic."""
    from hifihr_amd.traineval import HO3D2Frei, Frei2HO3D, data_dic
    from hifihr_amd.synth import to_ho3d_sample
    mapping = {0: 0, 1: 13, 2: 14, 3: 15, 4: 16, 5: 1, 6: 2, 7: 3, 8: 17, 9: 4, 10: 5, 11: 6, 12: 18, 13: 10, 14: 11, 15: 12, 16: 19,
               17: 7, 18: 8, 19: 9, 20: 20}                      # myId -> ho3dId, fh_utils.py:607-612
    ho = torch.arange(21.0).view(1, 21, 1).repeat(2, 1, 3)
    fr = HO3D2Frei(ho)
    for my, hid in mapping.items():
        assert float(fr[0, my, 0]) == hid
    assert torch.equal(Frei2HO3D(fr), ho)
    gen = torch.Generator().manual_seed(5)
    B = 3
    K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 500.0; K[:, 0, 2] = 110.0; K[:, 1, 2] = 115.0; K[:, 2, 2] = 1.0
    joints = torch.randn(B, 21, 3, generator=gen) * 0.05 + torch.tensor([0.0, 0.0, 0.6])
    frei = {"trans_images": torch.rand(B, 3, 224, 224, generator=gen), "trans_Ks": K, "trans_joints": joints,
            "trans_verts": torch.randn(B, 778, 3, generator=gen), "trans_masks": (torch.rand(B, 3, 224, 224, generator=gen) > 0.5).float(),
            "scales": torch.ones(B), "idxs": torch.arange(B)}
    args = options.make_args()
    a = data_dic(frei, "FreiHand", "training", args, device="cpu")
    h = data_dic(to_ho3d_sample(frei), "HO3D", "training", args, device="cpu")
    for k in ("imgs", "Ks", "Ps", "joints", "masks", "segms_gt"):
        assert torch.equal(a[k], h[k]), k
    assert torch.allclose(a["j2d_gt"], h["j2d_gt"], atol=1e-4)
    assert "verts" not in h and "scales" not in h and h["root_xyz"].shape == (B, 3)
    assert h["segms_gt"].dtype == torch.int64 and tuple(h["Ps"].shape) == (B, 3, 4) and float(h["Ps"][:, :, 3].abs().max()) == 0.0
    # evaluation queries carry no trans_ prefix
    ev = data_dic({"images": frei["trans_images"], "Ks": K, "joints": joints, "idxs": torch.arange(B)}, "FreiHand", "evaluation", args, "cpu")
    assert torch.equal(ev["joints"], joints) and "verts" not in ev and "segms_gt" not in ev


def test_batched_affine_terms_equal_per_sample_path():
    """hifihr_amd.data.batch_affine_terms (stacked numpy) == the per-sample restatement of utils/handutils.py:63-101 that the
    golden PIL vectors pin, bit for bit, over many rotations (incl. the ones of tests/golden/data_path.npz)."""
    import time
    from hifihr_amd.data import affine_for_rotation, batch_affine_terms, pil_affine_fixed_terms
    rng = np.random.RandomState(0)
    for res in (224, 96):
        rots = np.concatenate([rng.uniform(-np.pi, np.pi, 3000), [0.3, -2.1, 3.0, 0.0, 1.5707963, np.pi, -np.pi]])
        center = np.asarray([res // 2, res // 2])
        fixed, post, rmat = batch_affine_terms(center, res, [res, res], rots)
        for i, r in enumerate(rots):
            total, p = affine_for_rotation(center, res, [res, res], r)
            assert pil_affine_fixed_terms(total) == fixed[i].tolist(), (res, r)
            assert np.array_equal(p, post[i])
            want = np.array([[np.cos(r), -np.sin(r), 0], [np.sin(r), np.cos(r), 0], [0, 0, 1]]).astype(np.float32)
            assert np.array_equal(rmat[i], want)
    t0 = time.perf_counter()
    batch_affine_terms(np.asarray([112, 112]), 224, [224, 224], rots[:32])
    assert time.perf_counter() - t0 < 0.01


def test_train_front_end_reads_reference_configs(tmp_path):
    """train_hrnet.py --config_json: JSON overlay with unknown keys accepted (reference train_hrnet.py:505-510), lambda_* from
    the first list element, nimble configs mapped to the declared MANO + texture stand-in."""
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import train_hrnet as T
    cfg = {"train_datasets": ["FreiHand"], "hand_model": "nimble", "pretrain": "effb3", "train_batch": 48, "some_unknown_key": 1,
           "lambda_pose_list": [0.01, 0.001], "lambda_pose_steps": [10], "losses": ["joint_3d", "mtex", "perceptual"],
           "base_out_path": str(tmp_path), "save_interval": 5, "optimizer": "AdamW"}
    f = tmp_path / "c.json"
    f.write_text(json.dumps(cfg))
    args = T.build_args(T.parse(["--config_json", str(f), "--override", '{"total_epochs": 3}']))
    assert args.hand_model == "mano" and args.texture_stand_in == 10 and args.pretrain == "effb3" and args.train_batch == 48
    assert args.lambda_pose == 0.01 and args.some_unknown_key == 1 and args.total_epochs == 3 and args.save_interval == 5
    assert args.state_output == os.path.join(str(tmp_path), "model") and args.mode == ["training"] and args.optimizer == "AdamW"
    d = T.build_args(T.parse([]))
    assert d.hand_model == "mano" and d.texture_stand_in == 0 and d.pretrain == "res18" and d.if_test is True


def test_bench_gpus_n_launches_n_ranks_and_fails_loudly_when_one_dies():
    """`python bench.py --gpus 2` outside a launcher starts two rank processes itself (bench.launch_ranks).  Without a GPU every rank
    refuses to run (the HIP path has no CPU fallback), so here the launcher's failure path is what runs: both children started with
    their own RANK, the parent exits non-zero and names the ranks -- it must never print a line claiming n_gpus == 1."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU form of this test is tests/test_gpu_dp.py::test_bench_gpus_2_plain_invocation_starts_two_ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIFIHR_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank(s) failed" in r.stderr and "(0," in r.stderr and "(1," in r.stderr
    assert "bench.py needs a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_captured_step_refuses_a_batch_without_its_step_terms():
    """A captured step built on a batch that carries root_xyz / joints_rel / verts_rel / cam_ndc reads those: load_batch must not leave the
    previous batch's in place when a batch arrives without them."""
    import pytest
    import torch
    from hifihr_amd.traineval import _check_step_terms
    static = {"imgs": torch.zeros(1), "joints": torch.zeros(1), "joints_rel": torch.zeros(1), "cam_ndc": torch.zeros(1)}
    _check_step_terms(static, {"imgs": torch.zeros(1), "joints": torch.zeros(1), "joints_rel": torch.zeros(1), "cam_ndc": torch.zeros(1)})
    _check_step_terms({"imgs": torch.zeros(1)}, {"imgs": torch.zeros(1)})
    with pytest.raises(KeyError, match="joints_rel"):
        _check_step_terms(static, {"imgs": torch.zeros(1), "joints": torch.zeros(1)})


def test_adam_device_state_is_uploaded_only_when_the_host_view_changes():
    """hifihr_amd.optim.FusedAdam in graph mode with the step counter on the device (hifihr_adam_step_counted): prepare_step() uploads the
    48-byte state when the device does not hold what the coming step needs -- first use, a scheduler step, a restored counter -- and not
    in between (the kernel advances its own counter).  Host logic only: the library is a stub that records the uploads."""
    import struct
    import torch
    from hifihr_amd._lib import HifihrLib
    from hifihr_amd.optim import FlatParams, FusedAdam
    net = torch.nn.Linear(4, 3)
    opt = FusedAdam(FlatParams(net), lr=1e-3)

    class Stub:
        adam_state_image = staticmethod(HifihrLib.adam_state_image)
    uploads = []

    class State:                                        # stands in for the device tensor
        def copy_(self, image):
            uploads.append(struct.unpack("<dddddii", bytes(image.numpy().tobytes())))
    opt.graph_mode, opt._counted, opt._lib, opt._state = True, True, Stub(), State()
    def replayed_step():
        opt.prepare_step(); opt.note_step_done()        # (GraphedTrainStep.__call__: prepare, graph.replay(), note)
    for _ in range(3):
        replayed_step()
    assert len(uploads) == 1 and uploads[0][0] == 1e-3 and uploads[0][5] == 0 and opt.step_count == 3      # (lr, ..., completed steps = 0)
    opt.param_groups[0]["lr"] = 5e-4                    # MultiStepLR fires
    replayed_step(); replayed_step()
    assert len(uploads) == 2 and uploads[1][0] == 5e-4 and uploads[1][5] == 3
    assert abs(uploads[1][3] - 0.9 ** 3) < 1e-15 and abs(uploads[1][4] - 0.999 ** 3) < 1e-15                # the running products beta^t
    opt.step_count = 40                                 # a restored snapshot / checkpoint
    replayed_step()
    assert len(uploads) == 3 and uploads[2][5] == 40 and opt.step_count == 41
    replayed_step()
    assert len(uploads) == 3 and opt.step_count == 42
    # a prepared step whose replay never ran (an exception in between, a caller that prepares twice): the next prepare_step takes the
    # count back and re-uploads -- the host counter stays the number of steps enqueued, the device counter is put right
    opt.prepare_step()
    assert opt.step_count == 43
    opt.prepare_step()
    assert opt.step_count == 43 and len(uploads) == 4 and uploads[3][5] == 42
    opt.note_step_done()
    replayed_step()
    assert len(uploads) == 4 and opt.step_count == 44
    # an eager launch in graph mode must have been prepared
    import pytest
    opt.flatp.flat.data = opt.flatp.flat.data                      # (CPU tensors: step() stops at require_cuda before any launch)
    with pytest.raises(Exception):
        opt.step()
    # leaving graph mode forgets what the device holds
    opt.disable_graph_mode()
    assert opt._state_sig is None and not opt._prepared


def test_staging_ring_releases_its_slots_in_groups():
    """hifihr_amd.data.FreiHandDeviceCache._stage / _release (host side of the batch parameters): every pinned slot exists after the first call,
    the slots are handed out in ring order, and a release event is recorded behind the LAST slot of each group only (one barrier packet per
    _GROUP steps instead of one per step) -- on a CPU 'device' nothing is recorded, the bookkeeping is the same."""
    import torch
    from hifihr_amd.data import FreiHandDeviceCache as C

    class Probe:
        _RING, _GROUP = C._RING, C._GROUP
        _stage, _release, _pinned = C._stage, C._release, C._pinned
        device = torch.device("cpu")
    p = Probe()
    order = []
    for _ in range(C._RING + 5):
        slot, host = p._stage(25 * 4)
        assert host.numel() == 100 and host.dtype == torch.int32
        order.append(slot)
        p._release(slot)
    assert len(p._slots) == C._RING and all(s is not None and s.numel() >= 100 for s in p._slots)
    assert order == [(i + 1) % C._RING for i in range(C._RING + 5)]
    assert C._RING % C._GROUP == 0 and len(p._events) == C._RING // C._GROUP
    # a larger batch later: the slot grows, the others stay
    slot, host = p._stage(25 * 64)
    assert host.numel() == 1600 and p._slots[slot].numel() >= 1600
