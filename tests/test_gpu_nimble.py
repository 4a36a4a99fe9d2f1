"""The NIMBLE-shaped branch (hand_model == 'nimble': csrc/lbs.hip + csrc/texpca.hip + the renderer on the 5 990-vertex skin) on the GPU
against oracle/lbs_oracle.py / oracle/model_oracle.py.  The tables are seeded synthetic stand-ins (hifihr_amd/nimble_tables.py): the
reference's MyNIMBLELayer and its assets are absent, so this pins the kernels to the oracle's formulation at
NIMBLE's sizes, not to NIMBLE's numbers."""
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


@pytest.fixture(scope="module")
def nimble_tables():
    from hifihr_amd.nimble_tables import synthetic_nimble_tables
    return synthetic_nimble_tables(0)


@pytest.mark.parametrize("V,J,S,B", [(300, 7, 5, 3), (64, 1, 0, 2), (257, 32, 32, 2)])
def test_lbs_random_tables(lib, V, J, S, B):
    kc.lbs_case(lib, "cuda", kc.random_lbs_tables(V, J, S, seed=V + J), B, seed=J)


def test_lbs_on_mano_tables(lib, synth_tables):
    t = synth_tables
    kc.lbs_case(lib, "cuda", (t.v_template, t.shapedirs, t.J_regressor, t.weights, kc.MANO_PARENTS16), 32, seed=4)


@pytest.mark.parametrize("B", [2, 32])
def test_lbs_at_nimble_size(lib, nimble_tables, B):
    t = nimble_tables
    kc.lbs_case(lib, "cuda", (t.v_template, t.shapedirs, t.J_regressor, t.weights, t.parents), B, seed=B, pose_scale=0.3)


def _nimble_inputs(B, seed, image_size=224):
    gen = torch.Generator().manual_seed(seed)
    hp = {"pose_params": torch.randn(B, 30, generator=gen) * 0.4, "shape_params": torch.randn(B, 20, generator=gen),
          "texture_params": torch.randn(B, 10, generator=gen), "scale": torch.zeros(B, 1), "trans": torch.zeros(B, 3), "rot": None}
    images = torch.rand(B, 3, image_size, image_size, generator=gen)
    f = 1.6 * image_size
    Ks = torch.tensor([[f, 0.0, image_size / 2], [0.0, f, image_size / 2], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
    Ks[:, 0, 2] += torch.randn(B, generator=gen) * 6.0
    root_xyz = torch.tensor([0.0, 0.0, 0.55]).repeat(B, 1, 1) + torch.randn(B, 1, 3, generator=gen) * torch.tensor([0.02, 0.02, 0.04])
    light = {"colors": 0.4 + 0.3 * torch.rand(B, 3, generator=gen), "directions": torch.nn.functional.normalize(torch.randn(B, 3, generator=gen), dim=1)}
    return hp, images, Ks, root_xyz, light


@pytest.mark.parametrize("dat_name,mode_train,uv", [("FreiHand", True, False), ("HO3D", False, False), ("FreiHand", True, True)])
def test_nimble_model_tail_matches_oracle(nimble_tables, synth_tables, dat_name, mode_train, uv):
    """hand layer -> Mano2Frei -> root-relative -> skin render: every output of the nimble branch against the oracle, and the gradient
    of a scalar of (joints, mano_verts, nimble_joints, re_img) with respect to pose / shape / texture parameters."""
    from hifihr_amd.models import Model
    from oracle import model_oracle as mor
    B = 2
    if uv:                                   # the texture as an IMAGE sampled through per-face uvs (TexturesUV) instead of vertex colours
        import copy
        from hifihr_amd.nimble_tables import add_synthetic_uv
        nimble_tables = add_synthetic_uv(copy.copy(nimble_tables))
    model = Model(True, "cuda", False, "nimble", False, "res18", nimble_tables=nimble_tables, mano_tables=synth_tables).cuda()
    assert model.hand_layer.uv_texture == uv
    hp, images, Ks, root_xyz, light = _nimble_inputs(B, seed=11)
    hp_ref = {k: (v.clone().requires_grad_(True) if v is not None and k.endswith("_params") else v) for k, v in hp.items()}
    ref = mor.nimble_forward_tail(nimble_tables, hp_ref, images, Ks, root_xyz, light, dat_name=dat_name, mode_train=mode_train)
    hp_dev = {k: (v.cuda().requires_grad_(True) if v is not None and k.endswith("_params") else (v.cuda() if v is not None else None)) for k, v in hp.items()}
    out = model.hand_layer(hp_dev, handle_collision=False)
    out.update(hp_dev)
    out = model._nimble_tail(dat_name, mode_train, images.cuda(), out, {k: v.cuda() for k, v in light.items()}, Ks.cuda(), root_xyz.cuda())
    for k, tol in (("joints", 3e-6), ("mano_verts", 3e-6), ("nimble_joints", 3e-6), ("skin_verts", 3e-6)):
        assert float((out[k].detach().cpu() - ref[k].detach()).abs().max()) <= tol, k
    same = (out["face_id"].cpu().numpy() == ref["face_id"].numpy())
    assert same.mean() > 0.9995, same.mean()                      # edge pixels may flip under f32 projection rounding
    ok = torch.from_numpy(np.asarray(same)).view(B, 224, 3, 224, 3).permute(0, 1, 3, 2, 4).reshape(B, 224, 224, 9).all(-1)
    diff = (out["re_img"].detach().cpu() - ref["re_img"].detach()).abs().amax(1)
    # 1e-4: the stand-in skin's triangles are slivers (499 rings x 12 segments), so f32 barycentrics are conditioned worse than on MANO
    assert float(diff[ok].max()) <= 1e-4 and ok.float().mean() > 0.995
    assert out["mano_faces"].shape == (B, 1538, 3) and out["re_sil"].shape == (B, 1, 224, 224)
    gen = torch.Generator().manual_seed(5)
    wj, wv, wn = torch.randn(B, 21, 3, generator=gen), torch.randn(B, 778, 3, generator=gen), torch.randn(B, 25, 3, generator=gen)
    wi = torch.randn(B, 3, 224, 224, generator=gen) * ok.unsqueeze(1) * 1e-3
    (ref["joints"] * wj).sum().add((ref["mano_verts"] * wv).sum()).add((ref["nimble_joints"] * wn).sum()).add((ref["re_img"] * wi).sum()).backward()
    (out["joints"] * wj.cuda()).sum().add((out["mano_verts"] * wv.cuda()).sum()).add((out["nimble_joints"] * wn.cuda()).sum()).add(
        (out["re_img"] * wi.cuda()).sum()).backward()
    for k in ("pose_params", "shape_params", "texture_params"):
        g, r = hp_dev[k].grad.cpu(), hp_ref[k].grad
        assert float((g - r).abs().max()) <= 2e-3 * float(r.abs().max()), (k, float((g - r).abs().max()), float(r.abs().max()))


def test_nimble_model_trains(nimble_tables, synth_tables):
    """The full step with hand_model='nimble' (encoder -> heads -> NIMBLE-shaped layer -> skin render -> the loss list of
    full_rhd_freihand.json -> fused Adam) on a synthetic FreiHAND-shaped batch: finite terms, gradients reach every head, and the loss
    goes down over a few steps on one batch."""
    from hifihr_amd import ops, options, synth
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.models import Model
    from hifihr_amd.optim import FlatParams, FusedAdam
    from hifihr_amd.traineval import data_dic, train_step
    torch.manual_seed(0)
    dev = torch.device("cuda")
    args = options.baseline_config3_args(train_batch=4, pretrain="res18", hand_model="nimble")
    model = Model(True, dev, False, "nimble", False, "res18", nimble_tables=nimble_tables, mano_tables=synth_tables).to(dev).train()
    mano = ops.ManoLayerHandle(synth_tables)                          # the data side: FreiHAND's ground truth is MANO
    rend = ops.RendererHandle(synth_tables.faces, 778, image_size=224, aa=3)
    ex = data_dic(synth.make_batch(mano, rend, 4, device=dev), "FreiHand", "training", args, device=dev)
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-4)
    lossfn, hist = LossFunction(), []
    for it in range(5):
        loss, dic = train_step(model, lossfn, opt, ex, args)
        assert all(torch.isfinite(dic[k]) for k in args.losses), {k: float(dic[k]) for k in args.losses}
        if it == 0:
            for name in ("pose_reg", "shape_reg", "tex_reg"):
                g = [p.grad for p in getattr(model.hand_encoder, name).parameters()]
                assert all(x is not None and torch.isfinite(x).all() for x in g) and any(float(x.abs().max()) > 0 for x in g), name
        hist.append(float(loss))
    assert hist[-1] < hist[0], hist


def test_nimble_tail_at_config_batch(nimble_tables, synth_tables):
    """BASELINE configs[2] shape of the NIMBLE-shaped layer: B = 48 skins of 5 990 vertices / 11 976 faces with the texture image sampled
    through per-face uvs (TexturesUV), hand layer -> joints -> root-relative -> render, against oracle/model_oracle.nimble_forward_tail
    on a slice of the batch (the oracle renders 48 skins in minutes; images 0, 17 and 47 are compared): geometry 3e-6, face ids of the
    672^2 samples > 99.95 % identical (f32 projection rounding moves a few edge samples), pixels 1e-4 where the ids agree."""
    import copy
    from hifihr_amd.models import Model
    from hifihr_amd.nimble_tables import add_synthetic_uv
    from oracle import model_oracle as mor
    B = 48
    tabs = add_synthetic_uv(copy.copy(nimble_tables))
    model = Model(True, "cuda", False, "nimble", False, "res18", nimble_tables=tabs, mano_tables=synth_tables).cuda()
    hp, images, Ks, root_xyz, light = _nimble_inputs(B, seed=23)
    hp_dev = {k: (v.cuda() if v is not None else None) for k, v in hp.items()}
    with torch.no_grad():
        out = model.hand_layer(hp_dev, handle_collision=False)
        out.update(hp_dev)
        out = model._nimble_tail("FreiHand", True, images.cuda(), out, {k: v.cuda() for k, v in light.items()}, Ks.cuda(), root_xyz.cuda())
    assert out["re_img"].shape == (B, 3, 224, 224) and out["skin_verts"].shape == (B, 5990, 3)
    pick = [0, 17, 47]
    sub = lambda t: t[pick] if t is not None else None
    with torch.no_grad():
        ref = mor.nimble_forward_tail(tabs, {k: sub(v) for k, v in hp.items()}, images[pick], Ks[pick], root_xyz[pick],
                                      {k: v[pick] for k, v in light.items()}, dat_name="FreiHand", mode_train=True)
    for k, tol in (("joints", 3e-6), ("mano_verts", 3e-6), ("nimble_joints", 3e-6), ("skin_verts", 3e-6)):
        assert float((out[k][pick].cpu() - ref[k]).abs().max()) <= tol, k
    same = (out["face_id"][pick].cpu().numpy() == ref["face_id"].numpy())
    assert same.mean() > 0.9995, same.mean()
    ok = torch.from_numpy(np.asarray(same)).view(3, 224, 3, 224, 3).permute(0, 1, 3, 2, 4).reshape(3, 224, 224, 9).all(-1)
    diff = (out["re_img"][pick].cpu() - ref["re_img"]).abs().amax(1)
    assert float(diff[ok].max()) <= 1e-4 and ok.float().mean() > 0.995
