"""GPU parity of the pooling and fused-loss kernels (csrc/pool.hip, csrc/losses.hip) at the sizes of BASELINE configs[1]
against plain PyTorch fp32 (CPU) -- through the C ABI."""
import numpy as np
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hifihr_amd._lib import get_lib
    assert torch.cuda.is_available()
    return get_lib()


@pytest.mark.parametrize("B,H,C,ties", [(32, 14, 512, True), (4, 7, 1536, False)])
def test_mmpool(lib, B, H, C, ties):
    kc.mmpool_case(lib, "cuda", B, H, H, C, seed=C, ties=ties)


@pytest.mark.parametrize("N,H,C,ties", [(8, 112, 64, True), (2, 57, 64, False)])
def test_maxpool3x3s2(lib, N, H, C, ties):
    kc.maxpool_case(lib, "cuda", N, H, H, C, seed=H, ties=ties)


@pytest.mark.parametrize("B,mse", [(32, False), (6, True)])
def test_geom_losses(lib, B, mse):
    kc.geom_loss_case(lib, "cuda", B, 778, 1538, mse, seed=B)


@pytest.mark.parametrize("B,with_g", [(8, True), (3, False)])
def test_photo_losses(lib, B, with_g):
    kc.photo_loss_case(lib, "cuda", B, 224, 224, seed=B, with_g=with_g)


@pytest.mark.parametrize("B,I,O,act,bn", [(32, 512, 1024, 1, True), (32, 1024, 512, 1, True), (32, 512, 128, 1, False), (32, 128, 48, 0, False),
                                          (32, 32, 3, 0, False), (128, 512, 128, 1, False), (64, 1536, 1024, 1, True)])
def test_linear_heads(lib, B, I, O, act, bn):
    kc.linear_case(lib, "cuda", B, I, O, act, bn, seed=I + O)


def test_conv_halo_bias_relu_epilogue(lib):
    """VGG19's conv1_2 of the perceptual loss (64 -> 64 on 224 x 224, bias + ReLU) runs on conv_halo_kernel."""
    kc.conv_bias_relu_case(lib, "cuda", 4, 224, 224, 64, 64, 3, 1, seed=5, pad=1)
    kc.conv_bias_relu_case(lib, "cuda", 3, 19, 42, 64, 64, 3, 1, seed=6, pad=1)


@pytest.mark.parametrize("N,H,C,K,R,stride", [(32, 28, 128, 48, 1, 2), (32, 14, 48, 48, 3, 1), (32, 12, 48, 64, 3, 2), (32, 56, 32, 48, 1, 4)])
def test_light_estimator_convs(lib, N, H, C, K, R, stride):
    kc.conv_bias_relu_case(lib, "cuda", N, H, H, C, K, R, stride, seed=C + K)
    kc.conv_case(lib, "cuda", N, H, H, C, K, R, stride, 0, seed=K)          # dgrad / wgrad of the same geometry


def test_light_split_and_loss_total(lib):
    """The LightEstimator's output split (network/res_encoder.py:205-210) and the sum of the loss terms (train_hrnet.py:98-104), one launch
    each way each."""
    kc.light_split_case(lib, "cuda", B=32)
    kc.loss_total_case(lib, "cuda")


@pytest.mark.parametrize("N,H,C,ksp", [(32, 12, 48, (3, 1, 1)), (32, 5, 64, (2, 2, 0))])
def test_light_estimator_pools(lib, N, H, C, ksp):
    kc.maxpool_case(lib, "cuda", N, H, H, C, seed=C, ties=True, ksp=ksp)


@pytest.mark.parametrize("B,H,C,SQ", [(32, 112, 40, 10), (32, 28, 288, 12), (32, 14, 816, 34), (8, 7, 2304, 96)])
def test_squeeze_excite(lib, B, H, C, SQ):
    kc.se_case(lib, "cuda", B, H, H, C, SQ, seed=C)


@pytest.mark.parametrize("B,H,kind", [(2, 64, "l2"), (3, 40, "both")])
def test_perceptual_loss_matches_torch_restatement(B, H, kind):
    """VGG19 features[0:15] on the MFMA convolutions (bias / bias+ReLU epilogues, 2x2 max-pools) vs the torch restatement
    (oracle/torch_modules.PerceptualLossRef) with the same weights (reference utils/perceptual_loss.py:38-45): loss and dL/dfake."""
    from hifihr_amd.perceptual import PerceptualLoss
    from oracle.torch_modules import PerceptualLossRef
    hip = PerceptualLoss(type=kind, seed=3).cuda()
    ref = PerceptualLossRef(type=kind, seed=3)
    gen = torch.Generator().manual_seed(B)
    with torch.no_grad():                       # non-zero biases so the epilogue is exercised
        for m in ref.model:
            if hasattr(m, "bias"):
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))
    hip.load_vgg19_features({"features." + k: v for k, v in ref.model.state_dict().items()})
    fake = torch.rand(B, 3, H, H, generator=gen)
    real = torch.rand(B, 3, H, H, generator=gen)
    fr = fake.clone().requires_grad_(True)
    lr = ref(fr, real)
    lr.backward()
    fh = fake.cuda().requires_grad_(True)
    lh = hip(fh, real.cuda())
    lh.backward()
    assert abs(lh.item() - lr.item()) <= 1e-4 * abs(lr.item()) + 1e-7, (lh.item(), lr.item())
    # a pre-activation within rounding of zero may sit on the other side of its ReLU in the two implementations, which moves
    # the gradient of the few pixels under that unit: bound the error in norm and the number of such pixels, not the maximum
    g, gr = fh.grad.cpu(), fr.grad
    err = (g - gr).abs()
    assert float(err.norm()) <= 2e-2 * float(gr.norm()), (float(err.norm()), float(gr.norm()))
    assert float(err.median()) <= 1e-5 * float(gr.abs().max())
    assert float((err > 1e-3 * float(gr.abs().max())).float().mean()) <= 0.02


def test_procrustes_alignment_vs_reference(lib, golden_dir):
    kc.procrustes_case(lib, "cuda", golden_dir)


def test_freihand_augment_vs_reference_pil(lib, golden_dir):
    kc.augment_case(lib, "cuda", golden_dir)


def test_device_cache_batch_matches_reference_sample(golden_dir):
    """hifihr_amd.data.FreiHandDeviceCache.batch == the reference's per-sample augmentation (data/dataset.py:223-280) for the
    golden rotations: pixels bit-exact, K / joints to fp32 rounding; and it feeds data_dic unchanged."""
    import os
    from hifihr_amd import options
    from hifihr_amd.data import FreiHandDeviceCache
    from hifihr_amd.traineval import data_dic
    g = np.load(os.path.join(golden_dir, "data_path.npz"))
    ids = [1, 2, 3]                                    # the three 96 x 96 cases
    cache = FreiHandDeviceCache(np.stack([g[f"img{i}"] for i in ids]), np.stack([g[f"mask{i}"] for i in ids]),
                                np.stack([g[f"K{i}"] for i in ids]), np.stack([g[f"joints{i}"] for i in ids]),
                                np.zeros((3, 778, 3), np.float32))
    order = [2, 0, 1]
    s = cache.batch(order, rots=[float(g[f"rot{ids[k]}"]) for k in order])
    for b, k in enumerate(order):
        i = ids[k]
        assert torch.equal(s["trans_images"][b].cpu(), torch.from_numpy(g[f"timg{i}"]).permute(2, 0, 1).float().div(255))
        assert torch.equal(s["trans_masks"][b, 1].cpu(), torch.round(torch.from_numpy(g[f"tmask{i}"]).float().div(255)))
        np.testing.assert_allclose(s["trans_Ks"][b].cpu().numpy(), g[f"tK{i}"], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(s["trans_joints"][b].cpu().numpy(), g[f"tjoints{i}"], rtol=1e-6, atol=1e-7)
    ex = data_dic(s, "FreiHand", "training", options.make_args(), device="cuda")
    assert ex["imgs"].data_ptr() == s["trans_images"].data_ptr() and ex["segms_gt"].dtype == torch.int64     # no copies on the way


def test_freihand_batch_two_launches(lib):
    kc.freihand_batch_case(lib, "cuda", seed=1)
    kc.freihand_batch_case(lib, "cuda", seed=3, B=32, n=40, res=224, J=21, V=778)


def test_device_cache_batch_examples_matches_data_dic(golden_dir):
    """FreiHandDeviceCache.batch_examples (one staged copy, two launches, optionally straight into a captured step's static inputs)
    == data_dic(cache.batch(...)): pixels / masks / indices exact, camera / joint terms to fp32 rounding."""
    import os
    from hifihr_amd import options
    from hifihr_amd.data import FreiHandDeviceCache
    from hifihr_amd.traineval import data_dic
    g = np.load(os.path.join(golden_dir, "data_path.npz"))
    ids = [1, 2, 3]
    rng = np.random.default_rng(0)
    cache = FreiHandDeviceCache(np.stack([g[f"img{i}"] for i in ids]), np.stack([g[f"mask{i}"] for i in ids]),
                                np.stack([g[f"K{i}"] for i in ids]), np.stack([g[f"joints{i}"] for i in ids]),
                                (rng.normal(0, 0.05, (3, 778, 3)) + np.array([0, 0, 0.6])).astype(np.float32))
    order, rots = [2, 0, 1, 1], [0.3, -1.2, 2.9, 0.0]
    want = data_dic(cache.batch(order, rots=rots), "FreiHand", "training", options.make_args(), device="cuda")
    got = cache.batch_examples(order, rots=rots)
    assert set(got) == set(want) == set(FreiHandDeviceCache.EXAMPLE_KEYS)
    static = {k: torch.full_like(v, 3) for k, v in want.items()}
    static["extra"] = torch.zeros(1, device="cuda")
    into = cache.batch_examples(order, rots=rots, out=static)
    for res in (got, into):
        for k in ("imgs", "masks", "segms_gt", "idxs", "scales"):
            assert res[k].dtype == want[k].dtype and torch.equal(res[k], want[k]), k
        for k in ("Ks", "Ps", "joints", "verts", "j2d_gt"):
            assert res[k].shape == want[k].shape
            assert float((res[k] - want[k]).abs().max()) <= 2e-6 * max(1.0, float(want[k].abs().max())) * (5 if k == "j2d_gt" else 1), k
    assert all(into[k].data_ptr() == static[k].data_ptr() for k in into)
    with pytest.raises(ValueError):
        cache.batch_examples(order[:2], rots=rots[:2], out=static)
    # the A/B form whose kernels read the parameters from the pinned host slot themselves (HIFIHR_BATCH_DIRECT_PARAMS=1): the same bits;
    # more batches than the staging ring has slots (the grouped release events)
    from hifihr_amd import data as data_mod
    prev = data_mod._DIRECT_PARAMS
    try:
        data_mod._DIRECT_PARAMS = True
        for _ in range(FreiHandDeviceCache._RING + 3):
            direct = cache.batch_examples(order, rots=rots)
        torch.cuda.synchronize()
        assert all(torch.equal(direct[k], got[k]) for k in got)
    finally:
        data_mod._DIRECT_PARAMS = prev


def test_ho3d_crop_resize_vs_pillow(lib, golden_dir):
    kc.ho3d_batch_case(lib, "cuda", golden_dir)


def test_ho3d_device_cache_matches_oracle_sample():
    """hifihr_amd.data.HO3DDeviceCache.batch on full-size 480 x 640 frames == oracle/ho3d_oracle.ho3d_sample (the restatement pinned to
    the reference's window lines and to Pillow): pixels and masks bit for bit, uv21_crop exact, K_crop to fp32 rounding; and it feeds the
    HO3D branch of data_dic."""
    from hifihr_amd import options
    from hifihr_amd.data import HO3DDeviceCache
    from hifihr_amd.traineval import data_dic
    from oracle import ho3d_oracle as ho
    rng = np.random.default_rng(3)
    n = 3
    imgs = rng.integers(0, 256, (n, 480, 640, 3), dtype=np.uint8)
    masks = np.zeros((n, 480, 640), np.uint8)
    Ks = np.tile(np.array([[614.6, 0, -320.1], [0, -614.2, -239.5], [0, 0, -1]], np.float32), (n, 1, 1))       # camMat . diag(1, -1, -1)
    xyz = np.zeros((n, 21, 3), np.float32)
    for i, (cx, cy, spread) in enumerate(((200.0, 150.0, 0.02), (520.0, 400.0, 0.05), (320.0, 240.0, 0.12))):
        z = -0.6 - 0.1 * rng.random(21)
        xyz[i, :, 2] = z
        xyz[i, :, 0] = ((cx - 320.1) / 614.6 + spread * rng.normal(size=21)) * -z
        xyz[i, :, 1] = -((cy - 239.5) / 614.2 + spread * rng.normal(size=21)) * -z
        yy, xx = np.mgrid[:480, :640]
        masks[i][(yy - cy) ** 2 + (xx - cx) ** 2 < (40 + 30 * i) ** 2] = 255
    cache = HO3DDeviceCache(imgs, masks, Ks, xyz)
    order = [2, 0, 1, 1]
    noise = rng.normal(size=(4, 2)).astype(np.float32) * 5
    snoise = (0.9 - 0.1 * rng.random(4)).astype(np.float32)
    s = cache.batch(order, center_noise=noise, scale_noise=snoise)
    for b, i in enumerate(order):
        want = ho.ho3d_sample(imgs[i], masks[i], cache.uv21_host[i], Ks[i], noise[b], snoise[b])
        assert torch.equal(s["img_crop"][b].cpu(), torch.from_numpy(want["img_crop"])), b
        assert torch.equal(s["hand_mask_crop"][b].cpu(), torch.from_numpy(want["hand_mask_crop"])), b
        assert torch.equal(s["uv21_crop"][b].cpu(), torch.from_numpy(want["uv21_crop"])), b
        np.testing.assert_allclose(s["K_crop"][b].cpu().numpy(), want["K_crop"], rtol=1e-6, atol=1e-3)
        assert torch.equal(s["xyz21"][b].cpu(), torch.from_numpy(xyz[i]))
    assert 0.02 < float(s["hand_mask_crop"].mean()) < 0.9                       # the windows do contain the hand
    ex = data_dic(s, "HO3D", "training", options.make_args(), device="cuda")
    assert ex["imgs"].shape == (4, 3, 224, 224) and ex["segms_gt"].dtype == torch.int64 and ex["j2d_gt"].shape == (4, 21, 2)
    r = cache.batch(order, generator=torch.Generator().manual_seed(1))          # the loader's own noise draws
    assert r["img_crop"].shape == (4, 3, 224, 224) and torch.isfinite(r["K_crop"]).all()
    # evaluation split: the window comes from the hand bounding box, root_xyz is returned with y / z negated (dataset.py:1071-1080)
    boxes = np.array([[[150.0, 100.0], [260.0, 210.0]], [[400.0, 300.0], [630.0, 470.0]], [[10.0, 20.0], [90.0, 140.0]]], np.float32)
    roots = rng.normal(size=(n, 3)).astype(np.float32)
    ev = HO3DDeviceCache(imgs, masks, Ks, xyz, bboxes=boxes, root_xyz=roots)
    e = ev.batch(order, center_noise=noise, scale_noise=snoise)
    for b, i in enumerate(order):
        win = ho.crop_window(boxes[i], noise[b], float(snoise[b]))
        top, left, size = float(win["y1"]), float(win["x1"]), float(win["crop_size_scales"])
        want = ho.resized_crop_u8(imgs[i], top, left, size, size, 224, "bilinear").transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        assert torch.equal(e["img_crop"][b].cpu(), torch.from_numpy(want)), b
        assert torch.equal(e["root_xyz"][b].cpu(), torch.from_numpy(roots[i] * np.array([1, -1, -1], np.float32)))


def test_evaluator_summary_matches_formulas(golden_dir):
    """hifihr_amd.evaluate.Evaluator vs the formulas of train_hrnet.py:149-161, 227-243 written out with torch."""
    import os
    from hifihr_amd import ops
    from hifihr_amd.evaluate import Evaluator, align_w_scale
    g = np.load(os.path.join(golden_dir, "eval.npz"))
    gen = torch.Generator().manual_seed(2)
    ev = Evaluator()
    B = 3
    for half in (0, 1):
        sl = slice(half * B, half * B + B)
        out = {"joints": torch.from_numpy(g["pr_j"][sl]).cuda(), "mano_verts": torch.from_numpy(g["pr_v"][sl]).cuda(),
               "re_img": torch.rand(B, 3, 224, 224, generator=gen).cuda()}
        ex = {"imgs": torch.rand(B, 3, 224, 224, generator=gen).cuda(), "segms_gt": (torch.rand(B, 224, 224, generator=gen) > 0.6).long().cuda()}
        ev.collect(out, ex, "FreiHand")
        if half == 1:
            m = ex["segms_gt"].unsqueeze(1).float()
            last = (out["re_img"] * m, m * ex["imgs"])
    s = ev.summary(g["gt_j"], g["gt_v"])
    assert abs(s["pose_3d"] - float(g["mpjpe"])) <= 1e-6 * float(g["mpjpe"]) and abs(s["vert_3d"] - float(g["mpvpe"])) <= 1e-6 * float(g["mpvpe"])
    assert s["lpips"] is None and 0 < s["psnr"] < 30 and 0 < s["l1"] < 1
    mse = torch.nn.functional.mse_loss(*last)
    assert s["l2"] > 0 and abs(float(ops.ssim(*[t.contiguous() for t in last])) - float(ev.texture[-1]["ssim"])) < 1e-7
    assert abs(float(-10 * mse.log10()) - float(ev.texture[-1]["psnr"])) < 1e-5
    import json as _json
    import tempfile
    with tempfile.TemporaryDirectory() as td:                       # utils/train_utils.py:242-254
        assert ev.dump(os.path.join(td, "test", "3", "pred.json")) == (6, 6)
        xyz, verts = _json.load(open(os.path.join(td, "test", "3", "pred.json")))
        assert len(xyz) == 6 and len(xyz[0]) == 21 and len(verts[0]) == 778 and abs(xyz[4][7][2] - float(g["pr_j"][4, 7, 2])) < 1e-7
    al, err = align_w_scale(torch.from_numpy(g["gt_j"]).float().cuda(), torch.from_numpy(g["pr_j"]).cuda(), return_error=True)
    np.testing.assert_allclose(al.cpu().numpy(), g["al_j"], atol=2e-7)
    assert abs(float(err.mean()) - float(g["mpjpe"])) <= 1e-6 * float(g["mpjpe"])


def test_grouped_linear_layers_equal_single_launches(lib):
    kc.linear_group_case(lib, "cuda")


@pytest.mark.parametrize("B,in_dim,train", [(80, 1536, True), (128, 512, True), (48, 1536, True), (16, 512, False)])
def test_hand_encoder_module_matches_torch_restatement(B, in_dim, train):
    """The product HandEncoder (fused linear + BatchNorm1d + ReLU launches, grouped heads; B > 64 and evaluation mode: linear ->
    csrc/bn.hip batch-norm) against oracle/torch_modules.HandEncoderRef with the same weights: every output and the gradients of the
    first layer / a head, at the batch sizes the reference's configs use per GPU (48-128) -- no torch.nn fallback at any size."""
    from hifihr_amd.network import HandEncoder
    from oracle.torch_modules import HandEncoderRef
    torch.manual_seed(B)
    ref = HandEncoderRef("mano", [10, 48, 10], in_dim=in_dim)
    hip = HandEncoder("mano", [10, 48, 10], in_dim=in_dim).cuda()
    hip.load_state_dict(ref.state_dict())
    ref.train(train); hip.train(train)
    x = torch.randn(B, in_dim)
    xr = x.clone().requires_grad_(True); xh = x.cuda().requires_grad_(True)
    o_r, o_h = ref(xr), hip(xh)
    keys = ("pose_params", "shape_params", "texture_params", "scale", "trans", "rot")
    for k in keys:
        a, b = o_h[k].detach().cpu(), o_r[k].detach()
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), k
    gen = torch.Generator().manual_seed(1)
    ws = {k: torch.randn(o_r[k].shape, generator=gen) for k in keys}
    sum((o_r[k] * ws[k]).sum() for k in keys).backward()
    sum((o_h[k] * ws[k].cuda()).sum() for k in keys).backward()
    torch.cuda.synchronize()
    for name in ("base_layers.0.weight", "base_layers.1.weight", "pose_reg.2.weight", "trans_reg.3.bias"):
        a = dict(hip.named_parameters())[name].grad.cpu(); b = dict(ref.named_parameters())[name].grad
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-7, name
    assert float((xh.grad.cpu() - xr.grad).abs().max()) <= 2e-3 * float(xr.grad.abs().max()) + 1e-7
    if train:                                               # running statistics advanced identically
        assert float((hip.base_layers[1].running_mean.cpu() - ref.base_layers[1].running_mean).abs().max()) <= 1e-5


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("name", ["he_mano512", "he_nimble1536", "le512", "le32", "mmpool"])
def test_head_modules_match_reference_fixture(name, train):
    """The product HandEncoder / LightEstimator / MMPool (csrc/mlp.hip, conv epilogues, csrc/pool.hip) against what the REFERENCE's own
    classes (network/res_encoder.py:53-167, :169-209, :247-265) computed on the same seeded weights and input
    (tests/golden/heads.npz, tools/make_golden.gen_heads): every output, the gradient wrt the input and every parameter, the running
    statistics.  tests/test_oracle_heads.py pins the torch restatements the other tests use to the same file."""
    import heads_fixture as hf
    from test_oracle_heads import compare_with_fixture
    from hifihr_amd import network
    cls, cargs, _ = hf.CASES[name]
    torch.manual_seed(0)
    mod = getattr(network, cls)(*cargs)
    worst = compare_with_fixture(mod, name, train, "cuda", 2e-4, 2e-3)
    print(f"{name} train={train}: worst observed / bound = {worst:.3g}")


@pytest.mark.parametrize("B,K,n", [(32, 10, 2336), (48, 10, 1024 * 1024 * 3), (16, 10, 256 * 256 * 3)])
def test_texture_pca_decode(lib, B, K, n):
    """csrc/texpca.hip: the stand-in's 778 x 3 vertex colours and UV-map sizes (NIMBLE's texture maps are 1024^2 [recalled])."""
    kc.texture_pca_case(lib, "cuda", B, K, n, seed=K + B)


def test_light_split_matches_hardtanh_and_slices():
    """ops.light_split (one autograd node for `hardtanh(lights[:, :3])`, `lights[:, 3:]`, reference network/res_encoder.py:205-210) against
    the plain torch expression, values and gradients; outputs contiguous (the renderer takes them without a copy)."""
    import torch
    from hifihr_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(32, 6, device="cuda") * 2).requires_grad_(True)
    c, d = ops.light_split(x)
    assert c.is_contiguous() and d.is_contiguous()
    wc, wd = torch.randn_like(c), torch.randn_like(d)
    ((c * wc).sum() + (d * wd).sum()).backward()
    g = x.grad.clone(); x.grad = None
    c2, d2 = torch.nn.functional.hardtanh(x[:, :3]), x[:, 3:]
    ((c2 * wc).sum() + (d2 * wd).sum()).backward()
    assert torch.equal(c, c2) and torch.equal(d, d2) and torch.equal(g, x.grad)


@pytest.mark.parametrize("B,K,n", [(48, 10, 64 * 64 * 3), (5, 10, 512 * 512 * 3), (48, 30, 80)])
def test_texture_pca_backward_batch_tiles(lib, B, K, n):
    """texpca_bwd_kernel's (pieces of n, batch tiles) grid: one image per workgroup for a small texture, tiles of images for a large one,
    a batch that does not divide into the tiles."""
    kc.texture_pca_case(lib, "cuda", B, K, n, seed=B + K)
