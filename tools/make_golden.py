#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code in the build container.

Runs only where /root/reference exists (it never travels to the GPU box); only the small input /
output vectors it writes are committed.  What is imported from the reference, unmodified:

  utils/my_mano.py            ManoLayer (__init__ + forward)      -> mano_real.npz, mano_synth.npz
  utils/manopth/*             batch_rodrigues                      -> rodrigues.npz
  utils/pytorch_ssim          ssim                                 -> ssim.npz
  utils/Freihand_GNN_mano/network/resnet.py  (vendored torchvision ResNet) -> resnet18.npz
  network/efficientnet_pt     (not used yet)

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
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tools"))

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
    gen_losses()
    gen_effnet()


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


if __name__ == "__main__":
    if os.environ.get("GOLDEN_ONLY") == "losses":
        os.makedirs(OUT, exist_ok=True)
        gen_losses()
    elif os.environ.get("GOLDEN_ONLY") == "effnet":
        gen_effnet()
    else:
        main()
