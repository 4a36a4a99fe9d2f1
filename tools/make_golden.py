#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code in the build container.

Runs only where /root/reference exists (it never travels to the GPU box); only the small input /
output vectors it writes are committed.  What is imported from the reference, unmodified:

  utils/my_mano.py            ManoLayer (__init__ + forward)      -> mano_real.npz, mano_synth.npz
  utils/manopth/*             batch_rodrigues                      -> rodrigues.npz
  utils/pytorch_ssim          ssim                                 -> ssim.npz
  utils/Freihand_GNN_mano/network/resnet.py  (vendored torchvision ResNet) -> resnet18.npz
  network/efficientnet_pt     EfficientNet.from_name('efficientnet-b3')  -> effnet_b3_small.npz, state_dict_names.json
  utils/handutils.py          get_affine_transform, transform_img  -> data_path.npz
  (from source, see _extract_functions / _extract_defs: loss helpers, align_w_scale, HO3D2Frei / Frei2HO3D, and the
   HandEncoder / LightEstimator / MMPool classes: state-dict names, and outputs + gradients on seeded weights -> heads.npz)
  losses.py                   LossFunction.__call__ (:226-453) + utils/perceptual_loss.py PerceptualLoss, from source
                              (torchvision's VGG19 replaced by a seeded VGG19-shaped stack)        -> loss_dict.npz
  models_res_nimble.py        the resolve / re_sil / maskRGBs lines (:209-220) and get_ndc_fx_fy_cx_cy (:228-235) -> model_tail.npz

Stand-ins are installed ONLY for bookkeeping modules the container lacks (chumpy pickle classes,
pytorch3d.structures.Meshes container, cv2) and for the chumpy-based table loader
`ready_arguments`, which is replaced by one returning the same arrays (real pkl, or this repo's
synthetic MANO-shaped tables).  No arithmetic of the reference is replaced.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HIFIHR_REFERENCE", "/root/reference")
OUT = os.environ.get("HIFIHR_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "tests"))

from hifihr_amd.mano_tables import load_mano_pkl, synthetic_mano_tables  # noqa: E402


def install_standins():
    ch, chch, reo = (types.ModuleType(n) for n in ("chumpy", "chumpy.ch", "chumpy.reordering"))

    class Ch:
        def __setstate__(self, st):
            self.__dict__.update(st)

    class Select(Ch):
        pass

    chch.Ch = ch.Ch = Ch
    chch.MatVecMult = None
    reo.Select = Select
    ch.ch, ch.reordering = chch, reo
    p3d, st, ms = (types.ModuleType(n) for n in ("pytorch3d", "pytorch3d.structures", "pytorch3d.structures.meshes"))

    class Meshes:
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    ms.Meshes = st.Meshes = Meshes
    st.meshes, p3d.structures = ms, st
    sys.modules.update({"chumpy": ch, "chumpy.ch": chch, "chumpy.reordering": reo, "cv2": types.ModuleType("cv2"),
                        "pytorch3d": p3d, "pytorch3d.structures": st, "pytorch3d.structures.meshes": ms})


class _R:
    def __init__(self, a):
        self.r = np.asarray(a)


_TABLES = {}


def _ready_arguments(path, posekey4vposed="pose"):
    t = _TABLES["current"]
    return {
        "shapedirs": _R(t.shapedirs.astype(np.float64)), "betas": _R(np.zeros(10)),
        "posedirs": _R(t.posedirs.astype(np.float64)), "v_template": _R(t.v_template.astype(np.float64)),
        "weights": _R(t.weights.astype(np.float64)),
        "J_regressor": scipy.sparse.csc_matrix(t.J_regressor.astype(np.float64)),
        "f": t.faces.astype(np.uint32), "hands_components": t.hands_components.astype(np.float64),
        "hands_mean": t.hands_mean.astype(np.float64),
        "kintree_table": np.array([[4294967295, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], list(range(16))]),
    }


def reference_mano_layer(tables):
    _TABLES["current"] = tables
    import utils.my_mano as mm
    return mm.ManoLayer(center_idx=9, flat_hand_mean=False, side="right",
                        mano_root=os.path.join(REF, "utils", "mano"), use_pca=True, ncomps=48)  # = my_mano.py:35-36


def mano_cases(seed):
    g = torch.Generator().manual_seed(seed)
    pose = 0.5 * torch.randn(4, 48, generator=g)
    beta = 0.5 * torch.randn(4, 10, generator=g)
    pose = torch.cat([pose, torch.zeros(1, 48)], 0)            # zero pose (axis-angle exactly 0)
    beta = torch.cat([beta, torch.zeros(1, 10)], 0)
    one = torch.zeros(1, 48); one[0, 0:3] = torch.tensor([0.3, -0.2, 0.9])   # global rotation only
    pose = torch.cat([pose, one], 0)
    beta = torch.cat([beta, 0.5 * torch.randn(1, 10, generator=g)], 0)
    wv = torch.randn(pose.shape[0], 778, 3, generator=g)
    wj = torch.randn(pose.shape[0], 21, 3, generator=g)
    return pose, beta, wv, wj


def gen_mano(name, tables, seed):
    layer = reference_mano_layer(tables)
    pose, beta, wv, wj = mano_cases(seed)
    pose.requires_grad_(True); beta.requires_grad_(True)
    verts, jtr = layer(pose, beta)
    ((verts * wv).sum() + (jtr * wj).sum()).backward()
    np.savez_compressed(os.path.join(OUT, name), pose=pose.detach().numpy(), beta=beta.detach().numpy(),
                        wv=wv.numpy(), wj=wj.numpy(), verts=verts.detach().numpy(), jtr=jtr.detach().numpy(),
                        gpose=pose.grad.numpy(), gbeta=beta.grad.numpy())
    print(name, "verts", tuple(verts.shape), "|verts|max", float(verts.abs().max()))


def gen_rodrigues():
    from utils.manopth import rodrigues_layer
    g = torch.Generator().manual_seed(7)
    aa = torch.randn(64, 3, generator=g)
    aa[0] = 0.0
    aa[1] = torch.tensor([1e-9, 0.0, 0.0])
    aa[2] = torch.tensor([1e-5, -1e-5, 2e-5])
    aa[3] = torch.tensor([3.14159, 0.0, 0.0])
    aa.requires_grad_(True)
    rot = rodrigues_layer.batch_rodrigues(aa)
    w = torch.randn(64, 9, generator=g)
    (rot * w).sum().backward()
    np.savez_compressed(os.path.join(OUT, "rodrigues.npz"), aa=aa.detach().numpy(), rot=rot.detach().numpy(),
                        w=w.numpy(), gaa=aa.grad.numpy())
    print("rodrigues ok")


def gen_ssim():
    import utils.pytorch_ssim as ps
    g = torch.Generator().manual_seed(11)
    a = torch.rand(2, 3, 64, 64, generator=g)
    b = (a + 0.25 * torch.rand(2, 3, 64, 64, generator=g)).clamp(0, 1)
    a.requires_grad_(True)
    val = ps.ssim(a, b)
    val.backward()
    # full-size scalar only (inputs regenerated from the seed by the test)
    g2 = torch.Generator().manual_seed(12)
    A = torch.rand(2, 3, 224, 224, generator=g2)
    Bm = torch.rand(2, 3, 224, 224, generator=g2)
    val224 = ps.ssim(A, Bm)
    np.savez_compressed(os.path.join(OUT, "ssim.npz"), a=a.detach().numpy(), b=b.numpy(), ssim=val.detach().numpy(),
                        ga=a.grad.numpy(), ssim224=val224.numpy())
    print("ssim", float(val), float(val224))


def gen_resnet18():
    """Reference's vendored torchvision ResNet-18 (utils/Freihand_GNN_mano/network/resnet.py) with the three stride
    edits of network/res_encoder.py:360-362, name-seeded weights (tools/seeded_init.py), train mode, fwd + bwd."""
    from seeded_init import seeded_state_dict
    spec = importlib.util.spec_from_file_location(
        "ref_resnet", os.path.join(REF, "utils", "Freihand_GNN_mano", "network", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = mod.resnet18()
    net.layer4[0].downsample[0].stride = (1, 1)
    net.layer4[0].conv1.stride = (1, 1)
    net.layer4[0].conv2.stride = (1, 1)
    net.load_state_dict(seeded_state_dict(net))
    net.train()
    g = torch.Generator().manual_seed(99)
    x = torch.rand(2, 3, 96, 96, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
    xn = (x - mean) / std                                        # normalize_batch_3C, res_encoder.py:212-216
    h = net.maxpool(net.relu(net.bn1(net.conv1(xn))))
    h = net.layer1(h)
    low = net.layer2(h)
    feat = net.layer4(net.layer3(low))
    wl = torch.randn(low.shape, generator=g); wf = torch.randn(feat.shape, generator=g)
    ((low * wl).sum() + (feat * wf).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "resnet18_small.npz"), x=x.numpy(), low=low.detach().numpy(),
                        feat=feat.detach().numpy(), wl=wl.numpy(), wf=wf.numpy(),
                        g_conv1=net.conv1.weight.grad.numpy(), g_bn1=net.bn1.weight.grad.numpy(),
                        g_l4c2=net.layer4[1].conv2.weight.grad.numpy()[:8], g_l2ds=net.layer2[0].downsample[0].weight.grad.numpy())
    print("resnet18", tuple(low.shape), tuple(feat.shape))


def gen_resnet18_b8():
    """The same trunk on a batch of EIGHT 64x64 images (round 3): with 8 x 4 x 4 = 128 samples per channel in the deepest train-mode
    batch-norms (the batch-of-2 fixture has 18) the gradients are well conditioned, so the GPU test can hold the MFMA trunk's
    weight gradients to 2e-3 instead of 1.5e-2.  Inputs and loss weights are NOT stored: the test regenerates them from the
    seed (torch's CPU generator is deterministic) and checks them against the stored float64 checksums."""
    from seeded_init import seeded_state_dict
    spec = importlib.util.spec_from_file_location(
        "ref_resnet", os.path.join(REF, "utils", "Freihand_GNN_mano", "network", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = mod.resnet18()
    net.layer4[0].downsample[0].stride = (1, 1)
    net.layer4[0].conv1.stride = (1, 1)
    net.layer4[0].conv2.stride = (1, 1)
    net.load_state_dict(seeded_state_dict(net))
    net.train()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(8, 3, 64, 64, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
    xn = (x - mean) / std                                        # normalize_batch_3C, res_encoder.py:212-216
    h = net.maxpool(net.relu(net.bn1(net.conv1(xn))))
    h = net.layer1(h)
    low = net.layer2(h)
    feat = net.layer4(net.layer3(low))
    wl = torch.randn(low.shape, generator=g); wf = torch.randn(feat.shape, generator=g)
    ((low * wl).sum() + (feat * wf).sum()).backward()
    # (convolution gradients: the first 8 output channels of each, to keep the fixture small)
    grads = {"g_" + n.replace(".", "_"): (p.grad.numpy()[:8] if p.grad.dim() == 4 else p.grad.numpy()) for n, p in net.named_parameters()
             if n in ("conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer1.1.bn2.weight", "layer2.0.conv1.weight",
                      "layer2.0.downsample.0.weight", "layer2.1.conv2.weight", "layer3.0.conv2.weight", "layer3.1.bn1.bias",
                      "layer4.0.downsample.0.weight", "layer4.0.bn1.weight", "layer4.1.conv2.weight", "layer3.1.conv1.weight")}
    np.savez_compressed(os.path.join(OUT, "resnet18_b8.npz"), low=low.detach().numpy(), feat=feat.detach().numpy(),
                        checksums=np.array([x.double().sum().item(), wl.double().sum().item(), wf.double().sum().item()]), **grads)
    print("resnet18 b8", tuple(low.shape), tuple(feat.shape), sorted(grads))


def gen_effnet():
    """Reference EfficientNet.from_name('efficientnet-b3').extract_features with name-seeded weights (tools/seeded_init.py):
    train mode (batch statistics + drop-connect under torch.manual_seed(5)), forward + gradients."""
    from seeded_init import seeded_state_dict
    from network.efficientnet_pt.model import EfficientNet
    net = EfficientNet.from_name("efficientnet-b3")
    net.load_state_dict(seeded_state_dict(net))
    net.train()
    g = torch.Generator().manual_seed(77)
    x = torch.rand(2, 3, 96, 96, generator=g)
    torch.manual_seed(5)
    feat, low = net.extract_features(x)
    wf = torch.randn(feat.shape, generator=g); wl = torch.randn(low.shape, generator=g)
    ((feat * wf).sum() + (low * wl).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "effnet_b3_small.npz"), x=x.numpy(), feat=feat.detach().numpy(), low=low.detach().numpy(),
                        wf=wf.numpy(), wl=wl.numpy(), g_stem=net._conv_stem.weight.grad.numpy(),
                        g_b3_expand=net._blocks[3]._expand_conv.weight.grad.numpy(),
                        g_b10_dw=net._blocks[10]._depthwise_conv.weight.grad.numpy(),
                        g_b20_se=net._blocks[20]._se_reduce.weight.grad.numpy(), g_head_bn=net._bn1.weight.grad.numpy(),
                        n_params=sum(p.numel() for n, p in net.named_parameters() if not n.startswith("_fc")))
    print("effnet", tuple(feat.shape), tuple(low.shape))


def main():
    os.makedirs(OUT, exist_ok=True)
    install_standins()
    m = types.ModuleType("utils.mano.webuser.smpl_handpca_wrapper_HAND_only")
    m.ready_arguments = _ready_arguments
    sys.modules[m.__name__] = m
    gen_mano("mano_synth.npz", synthetic_mano_tables(0), seed=0)
    pkl = os.path.join(REF, "data", "MANO_RIGHT.pkl")
    if os.path.exists(pkl):
        gen_mano("mano_real.npz", load_mano_pkl(pkl), seed=1)
    gen_rodrigues()
    gen_ssim()
    gen_resnet18()
    gen_resnet18_b8()
    gen_losses()
    gen_effnet()
    gen_state_dict_names()
    gen_heads()
    gen_data_path()
    gen_eval()
    gen_loss_dict()
    gen_model_tail()


# ---- loss helpers: the reference functions cannot be imported (module-level pytorch3d / torchvision imports in
# utils/losses_util.py, skimage in utils/fh_utils.py), so their SOURCE is extracted with ast and executed here,
# in the build container, to produce vectors.  Only the vectors are committed.
def _extract_functions(path, names):
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "np": np, "nn": torch.nn}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns


def gen_losses():
    lu = _extract_functions(os.path.join(REF, "utils", "losses_util.py"), {"bone_direction_loss", "edge_length_loss", "IOU", "iou"})
    fh = _extract_functions(os.path.join(REF, "utils", "fh_utils.py"), {"proj_func", "Mano2Frei"})
    g = torch.Generator().manual_seed(21)
    B = 3
    j = 0.1 * torch.randn(B, 21, 3, generator=g); jg = 0.1 * torch.randn(B, 21, 3, generator=g)
    j2 = 100 * torch.rand(B, 21, 2, generator=g); j2g = 100 * torch.rand(B, 21, 2, generator=g)
    con = torch.ones(B, 21, 1)
    v = 0.1 * torch.randn(B, 778, 3, generator=g); vg = 0.1 * torch.randn(B, 778, 3, generator=g)
    faces = torch.as_tensor(synthetic_mano_tables(0).faces.astype(np.int16)).unsqueeze(0).repeat(B, 1, 1)
    m1 = (torch.rand(B, 1, 32, 32, generator=g) > 0.5).float(); m2 = (torch.rand(B, 1, 32, 32, generator=g) > 0.5).float()
    K = torch.tensor([[500.0, 0, 112], [0, 510.0, 100], [0, 0, 1]]).repeat(B, 1, 1)
    xyz = torch.randn(B, 21, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.6])
    out = dict(
        j=j.numpy(), jg=jg.numpy(), j2=j2.numpy(), j2g=j2g.numpy(), v=v.numpy(), vg=vg.numpy(), m1=m1.numpy(), m2=m2.numpy(),
        K=K.numpy(), xyz=xyz.numpy(),
        bone3d=lu["bone_direction_loss"](j, jg, con).numpy(), bone2d=lu["bone_direction_loss"](j2, j2g, con).numpy(),
        edge=lu["edge_length_loss"](v, vg, faces).numpy(), iou=lu["iou"](m1, m2).numpy(),
        proj=fh["proj_func"](xyz, K).numpy(), mano2frei=fh["Mano2Frei"](xyz).numpy())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("losses ok", float(out["bone3d"]), float(out["edge"]), float(out["iou"]))


def _extract_defs(path, names, ns):
    """Like _extract_functions, for classes too: executes the named top-level defs of a reference file in `ns`."""
    import ast
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns


def gen_state_dict_names():
    """Names and shapes of the reference modules' state dicts (the .t7 checkpoint layout of utils/train_utils.py:116-202).
    network/res_encoder.py imports torchvision / timm at module level, so its classes are executed from source;
    Resnet_4C wraps torchvision's resnet18 = the vendored utils/Freihand_GNN_mano/network/resnet.py."""
    import json
    import io
    import contextlib
    from torch import nn
    from torch.nn import init
    import torch.nn.functional as F
    ns = {"torch": torch, "nn": nn, "init": init, "F": F}
    _extract_defs(os.path.join(REF, "network", "res_encoder.py"), {"HandEncoder", "LightEstimator", "MMPool", "weights_init"}, ns)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        mods = {
            "hand_encoder[mano,1536]": ns["HandEncoder"]("mano", [10, 48, None], in_dim=1536),
            "hand_encoder[nimble,1536]": ns["HandEncoder"]("nimble", [20, 30, 10], in_dim=1536),
            "hand_encoder[mano,512]": ns["HandEncoder"]("mano", [10, 48, None], in_dim=512),
            "light_estimator[32]": ns["LightEstimator"](32),
            "light_estimator[512]": ns["LightEstimator"](512),
            "mmpool": ns["MMPool"]((1, 1)),
        }
        from network.efficientnet_pt.model import EfficientNet
        mods["efficientnet-b3"] = EfficientNet.from_name("efficientnet-b3")
        spec = importlib.util.spec_from_file_location("ref_resnet", os.path.join(REF, "utils", "Freihand_GNN_mano", "network", "resnet.py"))
        rn = importlib.util.module_from_spec(spec); spec.loader.exec_module(rn)
        mods["resnet18"] = rn.resnet18()
        mods["resnet50"] = rn.resnet50()
    for k, m in mods.items():
        out[k] = {"state": [[n, list(t.shape)] for n, t in m.state_dict().items()],
                  "params": [n for n, _ in m.named_parameters()]}
    with open(os.path.join(OUT, "state_dict_names.json"), "w") as fh:
        json.dump(out, fh)
    print("state dict names", {k: len(v["state"]) for k, v in out.items()})


def gen_heads():
    """tests/golden/heads.npz: the reference's HandEncoder / LightEstimator / MMPool (network/res_encoder.py:53-167, :169-209, :247-265),
    executed from source (the module imports torchvision / timm at its top), on the weights and inputs of tests/heads_fixture.py:
    every output, the (sampled) gradients of a fixed random projection wrt the input and every parameter, the advanced running
    statistics -- train and eval mode."""
    import io
    import contextlib
    from torch import nn
    from torch.nn import init
    import torch.nn.functional as F
    import heads_fixture as hf
    ns = {"torch": torch, "nn": nn, "init": init, "F": F}
    _extract_defs(os.path.join(REF, "network", "res_encoder.py"), {"HandEncoder", "LightEstimator", "MMPool", "weights_init"}, ns)
    out = {}
    for name, (cls, cargs, _) in hf.CASES.items():
        for train in (True, False):
            with contextlib.redirect_stdout(io.StringIO()):
                m = ns[cls](*cargs)
            hf.fill_state(m, name)
            outs, grads, bufs = hf.run_case(m, name, train)
            tag = f"{name}/{'train' if train else 'eval'}"
            for k, v in outs.items():
                out[f"{tag}/out/{k}"] = v.detach().numpy()
            for k, v in grads.items():
                out[f"{tag}/grad/{k}"] = hf.sample(v)
            for k, v in bufs.items():
                out[f"{tag}/buf/{k}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "heads.npz"), **out)
    print("heads ok", len(out), "arrays,", sum(v.size for v in out.values()), "values")


def gen_data_path():
    """utils/handutils.py get_affine_transform + transform_img (PIL nearest-neighbour affine), as FreiHAND training samples
    use them (data/dataset.py:223-280), on seeded uint8 images; plus the K / joint updates of the same lines."""
    from PIL import Image
    import utils.handutils as hu
    rng = np.random.RandomState(11)
    out = {}
    for i, (res, rot) in enumerate([(224, 0.3), (96, -2.1), (96, 3.0), (96, 0.0), (64, 1.5707963)]):
        img = rng.randint(0, 256, size=(res, res, 3)).astype(np.uint8)
        mask = (rng.rand(res, res) > 0.5).astype(np.uint8)[:, :, None].repeat(3, 2) * 255
        center = np.asarray([res // 2, res // 2])
        aff, post = hu.get_affine_transform(center, res, [res, res], rot=rot)
        timg = np.asarray(hu.transform_img(Image.fromarray(img), aff, [res, res]))
        tmask = np.asarray(hu.transform_img(Image.fromarray(mask), aff, [res, res]))
        K = np.array([[500.0 + 10 * i, 0, res / 2 + 3], [0, 505.0, res / 2 - 2], [0, 0, 1]], dtype=np.float32)
        joints = (rng.randn(21, 3) * 0.05 + np.array([0, 0, 0.6])).astype(np.float32)
        rot_mat = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)
        out.update({f"img{i}": img, f"mask{i}": mask[:, :, 0], f"rot{i}": np.float64(rot), f"aff{i}": aff, f"post{i}": post,
                    f"timg{i}": timg, f"tmask{i}": tmask[:, :, 0], f"K{i}": K, f"tK{i}": post.dot(K).astype(np.float32),
                    f"joints{i}": joints, f"tjoints{i}": rot_mat.dot(joints.transpose(1, 0)).transpose()})
    out["n"] = np.int64(5)
    np.savez_compressed(os.path.join(OUT, "data_path.npz"), **out)
    print("data path ok")


def gen_ho3d_path():
    """tests/golden/ho3d_path.npz: the HO-3D sample assembly (data/dataset.py:1105-1215).
    (1) the crop window, the cropped 2-D joints and K_crop: the reference's own lines (:1106-1162, :1186-1188, :1206-1210) executed from
        source on seeded joints / intrinsics, with the torch RNG seeded so that its two noise draws can be regenerated;
    (2) the image / mask crops: Pillow itself, `Image.crop(box).resize((224, 224), filter)` = the PIL backend of torchvision's
        `resized_crop` (absent from this image; [recalled]) for the windows of (1) on seeded uint8 frames -- bilinear for the image
        (:1164, the default), bicubic for the hand mask (:1175)."""
    import textwrap
    from PIL import Image
    lines = open(os.path.join(REF, "data", "dataset.py")).read().split("\n")
    win_src = textwrap.dedent("\n".join(lines[1105:1162]))
    uv_src = textwrap.dedent("\n".join(lines[1185:1188]))
    k_src = textwrap.dedent("\n".join(lines[1205:1210]))
    assert "ho_scope = 0" in win_src and "sample['x2']=x2" in win_src and "uv21_crop = torch.stack" in uv_src and "K_crop = torch.mm" in k_src
    rng = np.random.RandomState(5)
    out = {}
    H, W = 120, 160                                             # frames at a quarter of the HO-3D size keep the fixture small; the window code clamps at 640 x 480 as written
    n = 6
    for i in range(n):
        cy, cx, r = rng.randint(40, 80), rng.randint(50, 110), rng.randint(10, 30)
        yy, xx = np.mgrid[:H, :W]
        # low-frequency content (keeps the fixture small) with a block of full-range noise around the hand (the resampler's rounding)
        img = np.stack([127.5 + 127.5 * np.sin(xx / (9.0 + c) + i) * np.cos(yy / (7.0 + 2 * c)) for c in range(3)], 2).astype(np.uint8)
        img[cy - 16:cy + 16, cx - 16:cx + 16] = rng.randint(0, 256, size=img[cy - 16:cy + 16, cx - 16:cx + 16].shape).astype(np.uint8)
        mask = np.zeros((H, W), np.uint8)
        mask[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = 255
        spread = [6.0, 15.0, 30.0, 50.0, 2.5, 22.0][i]
        uv21 = torch.tensor(np.stack([cx + rng.randn(21) * spread, cy + rng.randn(21) * spread], 1).astype(np.float32))
        Ks = torch.tensor([[614.6 + i, 0, 320.1], [0, 614.2, 239.5 - i], [0, 0, 1]], dtype=torch.float32)

        class _Self:
            inp_res1 = 224
        torch.manual_seed(100 + i)
        ns = {"torch": torch, "uv21": uv21, "uv6": None, "self": _Self(), "sample": {}, "Ks": Ks}
        exec(win_src, ns)
        exec(uv_src, ns)
        exec(k_src, ns)
        torch.manual_seed(100 + i)                             # the same two draws, in the code's order (:1120, :1126)
        noise = 5 * torch.randn([2])
        scale_noise = (1 - 1.1) * torch.rand(1) + 1 - 0.1
        y1, x1, size = ns["y1"].item(), ns["x1"].item(), ns["crop_size_scales"].item()
        box = (x1, y1, x1 + size, y1 + size)
        img_crop = np.asarray(Image.fromarray(img).crop(box).resize((224, 224), Image.BILINEAR))
        mask_crop = np.asarray(Image.fromarray(mask).crop(box).resize((224, 224), Image.BICUBIC))
        out.update({f"img{i}": img, f"mask{i}": mask, f"uv21_{i}": uv21.numpy(), f"K{i}": Ks.numpy(), f"noise{i}": noise.numpy(),
                    f"scale_noise{i}": scale_noise.numpy(), f"crop_center{i}": ns["crop_center"].numpy(), f"scale{i}": ns["scale"].numpy(),
                    f"size{i}": ns["crop_size_scales"].numpy(), f"x1_{i}": ns["x1"].numpy(), f"y1_{i}": ns["y1"].numpy(),
                    f"uv21_crop{i}": ns["uv21_crop"].numpy(), f"K_crop{i}": ns["K_crop"].numpy()})
        if i in (0, 1, 2, 5):                                  # pixel crops for an up-scaling, two near-1:1 and a down-scaling window
            out.update({f"img_crop{i}": img_crop, f"mask_crop{i}": mask_crop})
        else:
            out.pop(f"img{i}"); out.pop(f"mask{i}")
    out["n"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "ho3d_path.npz"), **out)
    print("ho3d path ok", [float(out[f"size{i}"]) for i in range(n)])


def gen_eval():
    """utils/train_utils.py align_w_scale (scipy orthogonal_procrustes) + the MPJPE / MPVPE reduction of
    train_hrnet.py:227-243, and the HO-3D joint maps of utils/fh_utils.py:604-629."""
    from scipy.linalg import orthogonal_procrustes
    tu = _extract_functions(os.path.join(REF, "utils", "train_utils.py"), {"align_w_scale"})
    tu["orthogonal_procrustes"] = orthogonal_procrustes
    fh = _extract_functions(os.path.join(REF, "utils", "fh_utils.py"), {"HO3D2Frei", "Frei2HO3D", "RHD2Frei"})
    rng = np.random.RandomState(5)
    B = 6
    gt_j = (rng.randn(B, 21, 3) * 0.04).astype(np.float64); gt_v = (rng.randn(B, 778, 3) * 0.04).astype(np.float64)
    def perturb(x):
        out = []
        for b in range(x.shape[0]):
            q, _ = np.linalg.qr(rng.randn(3, 3))
            if b % 2 == 0 and np.linalg.det(q) < 0:
                q[:, 0] *= -1
            out.append((x[b] @ q.T) * (0.7 + 0.6 * rng.rand()) + rng.randn(3) * 0.1 + rng.randn(*x[b].shape) * 0.004)
        return np.stack(out).astype(np.float32)
    pr_j, pr_v = perturb(gt_j), perturb(gt_v)
    al_j = np.stack([tu["align_w_scale"](gt_j[b], pr_j[b]) for b in range(B)])
    al_v = np.stack([tu["align_w_scale"](gt_v[b], pr_v[b]) for b in range(B)])
    mpjpe = np.linalg.norm(al_j - gt_j, ord=2, axis=-1).mean()
    mpvpe = np.linalg.norm(al_v - gt_v, ord=2, axis=-1).mean()
    j = torch.arange(2 * 21 * 3, dtype=torch.float32).view(2, 21, 3)
    np.savez_compressed(os.path.join(OUT, "eval.npz"), gt_j=gt_j, gt_v=gt_v.astype(np.float32), pr_j=pr_j, pr_v=pr_v, al_j=al_j,
                        al_v=al_v.astype(np.float32), mpjpe=mpjpe, mpvpe=mpvpe, j=j.numpy(), ho3d2frei=fh["HO3D2Frei"](j).numpy(),
                        frei2ho3d=fh["Frei2HO3D"](j).numpy(), rhd2frei=fh["RHD2Frei"](j).numpy())
    print("eval ok", mpjpe, mpvpe)


def _vgg19_features_seeded(seed=0):
    """A torchvision-vgg19().features-shaped nn.Sequential with torchvision's own initialisation (kaiming_normal fan_out, zero bias)
    drawn from a seeded generator in layer order: stands in for the downloaded weights, which do not exist offline."""
    from torch import nn
    cfg = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
    gen = torch.Generator().manual_seed(seed)
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            conv = nn.Conv2d(cin, v, 3, 1, 1)
            with torch.no_grad():
                w = torch.empty(v, cin, 3, 3)
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu", generator=gen)
                conv.weight.copy_(w)
                conv.bias.zero_()
            layers += [conv, nn.ReLU(inplace=False)]
            cin = v
    return nn.Sequential(*layers)


def reference_loss_function():
    """The reference's OWN `LossFunction` (losses.py:226-453) and `PerceptualLoss` (utils/perceptual_loss.py), executed from
    source.  Stand-ins only for what the container lacks: torchvision (vgg19 -> the seeded stack above, transforms.Normalize ->
    (x - mean) / std) and `.cuda()` (identity on this CPU-only box)."""
    import torch.nn as nn
    import torch.nn.functional as torch_f
    import utils.pytorch_ssim as pytorch_ssim
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")

    class _Weights:
        DEFAULT = None

    class _VGG:
        def __init__(self):
            self.features = _vgg19_features_seeded(0)

    class _Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tv.models.vgg19 = lambda weights=None: _VGG()
    tv.models.VGG19_Weights = _Weights
    tv.transforms.Normalize = _Normalize
    nn.Module.cuda = lambda self, *a, **k: self                 # build container only (no GPU): the reference calls .cuda() at init
    ns = {"torch": torch, "nn": nn, "F": torch_f, "torchvision": tv, "transforms": tv.transforms}
    _extract_defs(os.path.join(REF, "utils", "perceptual_loss.py"), {"PerceptualLoss"}, ns)
    lu = _extract_functions(os.path.join(REF, "utils", "losses_util.py"), {"bone_direction_loss", "edge_length_loss", "IOU", "iou"})
    ns2 = {"torch": torch, "nn": nn, "torch_f": torch_f, "pytorch_ssim": pytorch_ssim, "PerceptualLoss": ns["PerceptualLoss"],
           "bone_direction_loss": lu["bone_direction_loss"], "edge_length_loss": lu["edge_length_loss"], "iou": lu["iou"]}
    _extract_defs(os.path.join(REF, "losses.py"), {"LossFunction"}, ns2)
    return ns2["LossFunction"]()


from loss_cases import loss_dict_case  # noqa: E402  (tests/loss_cases.py: the seeded inputs, shared with the tests)


def gen_loss_dict():
    """tests/golden/loss_dict.npz: every term the reference's LossFunction.__call__ returns for the loss lists of BASELINE
    configs[1] (cfg2, L2 base loss), configs[2] (cfg3 + scale / mscale / iou) and configs[4] (ho3d + texture_con), plus
    d(sum of the selected terms)/d(re_img, joints, mano_verts, pose_params) for the first list."""
    lf = reference_loss_function()
    out = {}
    for name in ("cfg2", "cfg3", "ho3d"):
        args, ex, o, dat = loss_dict_case(name)
        leaves = {}
        if name != "cfg3":
            for k in ("re_img", "joints", "mano_verts", "pose_params", "shape_params"):
                o[k] = o[k].clone().requires_grad_(True)
                leaves[k] = o[k]
        d = lf(ex, o, args.losses, dat, args)
        for k, v in d.items():
            out[f"{name}/{k}"] = np.float64(v.detach().double().item())
        if leaves:
            total = sum(d[k] for k in args.losses if k in d)
            total.backward()
            for k, t in leaves.items():
                if t.grad is not None:
                    out[f"{name}/grad/{k}"] = t.grad.numpy()
        out[f"{name}/keys"] = np.array(sorted(d.keys()))
        print("loss_dict", name, {k: float(v) for k, v in d.items()})
    np.savez_compressed(os.path.join(OUT, "loss_dict.npz"), **out)


def gen_model_tail():
    """tests/golden/model_tail.npz: the reference's resolve + output lines (models_res_nimble.py:209-220) and its
    get_ndc_fx_fy_cx_cy (:228-235), executed from source on seeded inputs."""
    import ast
    import textwrap
    import torch.nn.functional as F
    path = os.path.join(REF, "models_res_nimble.py")
    lines = open(path).read().split("\n")
    body = textwrap.dedent("\n".join(lines[209:220]))             # rendered_images.permute ... outputs['maskRGBs'] = ...
    assert "avg_pool2d" in body and "maskRGBs" in body and "permute" in body, body
    g = torch.Generator().manual_seed(31)
    B, H, aa = 2, 16, 3
    rgba = torch.rand(B, H * aa, H * aa, 4, generator=g)
    rgba[..., 3] = (torch.rand(B, H * aa, H * aa, generator=g) > 0.7).float()        # hard alpha
    images = torch.rand(B, 3, H, H, generator=g)

    class _Self:
        aa_factor = aa
    ns = {"rendered_images": rgba.clone(), "images": images, "self": _Self(), "F": F, "outputs": {}, "torch": torch}
    exec(body, ns)
    o = ns["outputs"]
    tree = ast.parse(open(path).read())
    fn = [n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "Model" for n in c.body
          if isinstance(n, ast.FunctionDef) and n.name == "get_ndc_fx_fy_cx_cy"][0]
    ns2 = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns2)
    Ks = torch.tensor([[[520.0, 0, 118.5, 0], [0, 498.0, 101.25, 0], [0, 0, 1, 0]], [[610.0, 0, 112.0, 0], [0, 611.5, 95.0, 0], [0, 0, 1, 0]]])
    fl, pp = ns2["get_ndc_fx_fy_cx_cy"](None, Ks)
    np.savez_compressed(os.path.join(OUT, "model_tail.npz"), rgba=rgba.numpy(), images=images.numpy(), re_img=o["re_img"].numpy(),
                        re_sil=o["re_sil"].numpy(), maskRGBs=o["maskRGBs"].numpy(), Ks=Ks.numpy(), focal=fl.numpy(), principal=pp.numpy())
    print("model tail ok", float(o["re_sil"].max()))


if __name__ == "__main__":
    only = os.environ.get("GOLDEN_ONLY")
    if only in ("names", "heads", "data", "eval", "loss_dict", "model_tail", "ho3d"):
        os.makedirs(OUT, exist_ok=True)
        install_standins()
        {"names": gen_state_dict_names, "heads": gen_heads, "data": gen_data_path, "eval": gen_eval, "loss_dict": gen_loss_dict,
         "model_tail": gen_model_tail, "ho3d": gen_ho3d_path}[only]()
    elif only == "mano_real":                   # the real tables (container only; the output is derived MANO data: never committed)
        os.makedirs(OUT, exist_ok=True)
        install_standins()
        m = types.ModuleType("utils.mano.webuser.smpl_handpca_wrapper_HAND_only")
        m.ready_arguments = _ready_arguments
        sys.modules[m.__name__] = m
        gen_mano("mano_real.npz", load_mano_pkl(os.environ.get("HIFIHR_MANO_PKL") or os.path.join(REF, "data", "MANO_RIGHT.pkl")), seed=1)
    elif os.environ.get("GOLDEN_ONLY") == "losses":
        os.makedirs(OUT, exist_ok=True)
        gen_losses()
    elif os.environ.get("GOLDEN_ONLY") == "effnet":
        gen_effnet()
    elif os.environ.get("GOLDEN_ONLY") == "resnet18_b8":
        gen_resnet18_b8()
    else:
        main()
