"""The reference's .t7 checkpoint layout (utils/train_utils.py:14-202): state-dict names / shapes of every sub-module
pinned against the reference's own classes (tests/golden/state_dict_names.json, tools/make_golden.py), and the
torch.optim.Adam <-> fused flat Adam state conversion, by a round trip through a stand-in of the reference's side
(plain torch modules with the classifier head, torch.optim.Adam over model.parameters())."""
import argparse
import json
import os

import pytest
import torch
import torch.nn as nn

from hifihr_amd import checkpoint as ck
from hifihr_amd.optim import FlatParams, FusedAdam


@pytest.fixture(scope="module")
def names(golden_dir):
    return json.load(open(os.path.join(golden_dir, "state_dict_names.json")))


def _shapes(m):
    return [[n, list(t.shape)] for n, t in m.state_dict().items()]


def test_submodule_state_dicts_match_reference_classes(names):
    from hifihr_amd.effnet import EffiEncoder
    from hifihr_amd.network import HandEncoder, LightEstimator, ResEncoder
    assert _shapes(HandEncoder("mano", [10, 48, None], in_dim=1536)) == names["hand_encoder[mano,1536]"]["state"]
    assert _shapes(HandEncoder("nimble", [20, 30, 10], in_dim=1536)) == names["hand_encoder[nimble,1536]"]["state"]
    assert _shapes(HandEncoder("mano", [10, 48, None], in_dim=512)) == names["hand_encoder[mano,512]"]["state"]
    from oracle.torch_modules import EffiEncoderRef, LightEstimatorRef, ResEncoderRef
    # the product modules (hand-written kernels) and their torch restatements (oracle/) both carry the reference's names
    for flavour, (LE, EE, RE) in (("product", (LightEstimator, EffiEncoder, ResEncoder)),
                                  ("oracle", (LightEstimatorRef, EffiEncoderRef, ResEncoderRef))):
        assert _shapes(LE(32)) == names["light_estimator[32]"]["state"], flavour
        assert _shapes(LE(512)) == names["light_estimator[512]"]["state"], flavour
        # image encoders: the reference's names minus the unused classifier head, in the same order (parameters too)
        eff = EE("effb3")
        want = [["encoder." + n, s] for n, s in names["efficientnet-b3"]["state"] if not n.startswith("_fc.")]
        assert _shapes(eff) == want, flavour
        assert [n for n, _ in eff.named_parameters()] == ["encoder." + n for n in names["efficientnet-b3"]["params"] if not n.startswith("_fc.")]
        res = RE(pretrain="res18")
        want = [["mmpool.p", [1]]] + [["encoder1.model." + n, s] for n, s in names["resnet18"]["state"] if not n.startswith("fc.")]
        assert _shapes(res) == want, flavour
        res50 = RE(pretrain="res50")
        want = [["mmpool.p", [1]]] + [["encoder1.model." + n, s] for n, s in names["resnet50"]["state"] if not n.startswith("fc.")]
        assert _shapes(res50) == want, flavour
    assert names["resnet50"]["state"][-2:] == [["fc.weight", [1000, 2048]], ["fc.bias", [1000]]]
    assert names["mmpool"]["state"] == [["p", [1]]]
    assert names["resnet18"]["state"][-2:] == [["fc.weight", [1000, 512]], ["fc.bias", [1000]]]
    assert names["efficientnet-b3"]["state"][-2:] == [["_fc.weight", [1000, 1536]], ["_fc.bias", [1000]]]


class _Mine(nn.Module):
    """hifihr_amd.models.Model's trainable part (its MANO / renderer handles hold no parameters and need a GPU to build)."""

    def __init__(self, pretrain):
        super().__init__()
        from hifihr_amd.effnet import EffiEncoder
        from hifihr_amd.network import HandEncoder, LightEstimator, ResEncoder
        if pretrain in ("res18", "res50"):
            self.base_encoder = ResEncoder(pretrain=pretrain, if_4c=False)
            feat, low = (512, 128) if pretrain == "res18" else (2048, 512)
        else:
            self.base_encoder, feat, low = EffiEncoder("effb3"), 1536, 32
        self.hand_encoder = HandEncoder("mano", [10, 48, None], in_dim=feat)
        self.light_estimator = LightEstimator(low)


class _RefSide(nn.Module):
    """What the reference holds for the same architecture: identical sub-modules plus the encoder's classifier head."""

    def __init__(self, model, head):
        super().__init__()
        import copy
        self.base_encoder = copy.deepcopy(model.base_encoder)
        holder = self.base_encoder
        for part in head.split(".")[:-1]:
            holder = getattr(holder, part)
        o, i = ck._head_shape(head, model.base_encoder.state_dict())
        setattr(holder, head.split(".")[-1], nn.Linear(i, o))
        self.hand_encoder = copy.deepcopy(model.hand_encoder)
        self.light_estimator = copy.deepcopy(model.light_estimator)


@pytest.mark.parametrize("pretrain,head", [("res18", "encoder1.model.fc"), ("res50", "encoder1.model.fc"), ("effb3", "encoder._fc")])
def test_t7_round_trip_with_reference_side(tmp_path, pretrain, head):
    torch.manual_seed(0)
    model = _Mine(pretrain)
    ref = _RefSide(model, head)
    assert [n for n, _ in ref.named_parameters()] == ck.reference_param_names(model)
    # the reference's side takes two Adam steps with seeded gradients (the head gets none, like in training) and saves
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.999))
    rsch = torch.optim.lr_scheduler.MultiStepLR(ropt, [80, 160], 0.5)
    gen = torch.Generator().manual_seed(1)
    for _ in range(2):
        for n, p in ref.named_parameters():
            p.grad = None if n.startswith("base_encoder." + head) else 0.01 * torch.randn(p.shape, generator=gen)
        ropt.step()
    rsch.step()
    args = argparse.Namespace(pretrain_model=str(tmp_path / "texturehand_7.t7"), state_output=str(tmp_path / "out"), save_mode="separately")
    torch.save({"args": args, "optimizer": ropt.state_dict(), "epoch": 7, "scheduler": rsch.state_dict(),
                "base_encoder": ref.base_encoder.state_dict(), "hand_encoder": ref.hand_encoder.state_dict(),
                "light_estimator": ref.light_estimator.state_dict()}, args.pretrain_model)
    # this build loads it ...
    model2 = _Mine(pretrain)
    flat = FlatParams(model2)
    opt = FusedAdam(flat, lr=5e-2)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [80, 160], 0.5)
    _, epoch, _, _ = ck.load_model(model2, opt, sch, args)
    assert epoch == 7 and opt.step_count == 2 and opt.param_groups[0]["lr"] == 1e-3 and sch.last_epoch == 1
    rp = dict(ref.named_parameters())
    for n, p in model2.named_parameters():
        assert torch.equal(p.detach(), rp[n].detach()), n
    idx = {n: i for i, n in enumerate(ck.reference_param_names(model2))}
    rstate = ropt.state_dict()["state"]
    for p, o in zip(flat.params, flat.offsets):
        n = next(k for k, v in model2.named_parameters() if v is p)
        assert torch.equal(flat._view(opt.exp_avg, p, o), rstate[idx[n]]["exp_avg"]), n
        assert torch.equal(flat._view(opt.exp_avg_sq, p, o), rstate[idx[n]]["exp_avg_sq"]), n
    # ... and writes a file the reference's strict loaders accept, state intact
    files = ck.save_model(model2, opt, sch, 1, 7, args)
    assert [os.path.basename(f) for f in files] == ["texturehand_8.t7"]
    sd = torch.load(files[0], weights_only=False)
    assert sd["epoch"] == 8 and set(sd) >= {"args", "optimizer", "scheduler", "epoch", "base_encoder", "hand_encoder", "light_estimator"}
    ref2 = _RefSide(model, head)
    for sub in ("base_encoder", "hand_encoder", "light_estimator"):
        getattr(ref2, sub).load_state_dict(sd[sub], strict=True)
        assert all(v.is_contiguous() for v in sd[sub].values())
    for (n, a), (_, b) in zip(ref2.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), n                      # classifier head included: it was carried through
    ropt2 = torch.optim.Adam(ref2.parameters(), lr=1.0)
    ropt2.load_state_dict(sd["optimizer"])
    s1, s2 = ropt.state_dict(), ropt2.state_dict()
    assert s2["param_groups"][0]["lr"] == 1e-3 and s1["state"].keys() == s2["state"].keys()
    for k in s1["state"]:
        for f in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(s1["state"][k][f], s2["state"][k][f])
        assert float(s1["state"][k]["step"]) == float(s2["state"][k]["step"])


def test_freeze_model_modules_semantics():
    """utils/train_utils.py:205-240 + utils/visualize_util.py:932-939: only_train_texture freezes the image encoder and the
    pose / shape path of the HandEncoder (batch-norm momentum 0, requires_grad off), the texture / light heads keep training;
    parameters held directly by the frozen root keep their flag (the reference's quirk: ResEncoder... has none, MMPool.p sits in a child)."""
    import argparse
    m = _Mine("res18")
    m.hand_encoder.tex_reg = nn.Sequential(nn.Linear(512, 128), nn.ReLU(), nn.Linear(128, 10))
    frozen = ck.freeze_model_modules(m, argparse.Namespace(only_train_texture=True, only_train_regressor=False))
    assert frozen == ["base_encoder", "hand_encoder.base_layers", "hand_encoder.pose_reg", "hand_encoder.shape_reg"]
    assert not any(p.requires_grad for p in m.base_encoder.parameters())
    assert all(b.momentum == 0 for b in m.base_encoder.modules() if isinstance(b, nn.BatchNorm2d))
    assert not any(p.requires_grad for p in m.hand_encoder.pose_reg.parameters())
    assert m.hand_encoder.base_layers[1].momentum == 0
    assert all(p.requires_grad for p in m.hand_encoder.tex_reg.parameters()) and all(p.requires_grad for p in m.hand_encoder.trans_reg.parameters())
    assert all(p.requires_grad for p in m.light_estimator.parameters())
    flat = FlatParams(m)                                   # frozen parameters stay out of the flat buffers
    assert flat.param_count() == sum(p.numel() for p in m.parameters() if p.requires_grad) < 2_000_000
    lone = nn.Linear(4, 4)
    ck.rec_freeze(lone)                                    # no children: nothing is frozen
    assert lone.weight.requires_grad
    assert ck.freeze_model_modules(_Mine("res18"), argparse.Namespace()) == []
