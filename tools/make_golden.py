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
    spec = importlib.util.spec_from_file_location(
        "ref_resnet", os.path.join(REF, "utils", "Freihand_GNN_mano", "network", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(1234)
    net = mod.resnet18(pretrained=False) if "pretrained" in mod.resnet18.__code__.co_varnames else mod.resnet18()
    # the three stride edits of reference network/res_encoder.py:360-362
    net.layer4[0].downsample[0].stride = (1, 1)
    net.layer4[0].conv1.stride = (1, 1)
    net.layer4[0].conv2.stride = (1, 1)
    net.train()
    g = torch.Generator().manual_seed(99)
    x = torch.rand(2, 3, 64, 64, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
    xn = (x - mean) / std                                        # normalize_batch_3C, res_encoder.py:212-216
    h = net.maxpool(net.relu(net.bn1(net.conv1(xn))))
    h = net.layer1(h)
    low = net.layer2(h)
    feat = net.layer4(net.layer3(low))
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items() if "num_batches" not in k and not k.startswith("fc.")}
    # weights are needed to reproduce the activations; keep the fixture small with fp16-exact weights?  No:
    # store full fp32 weights compressed only for the first two stages; later stages checked by shape/sums.
    np.savez_compressed(os.path.join(OUT, "resnet18_small.npz"), x=x.numpy(),
                        low_mean=low.mean().item(), low_absmean=low.abs().mean().item(),
                        feat_mean=feat.mean().item(), feat_absmean=feat.abs().mean().item(),
                        low_shape=np.array(low.shape), feat_shape=np.array(feat.shape),
                        low_samples=low.detach().flatten()[::997].numpy(), feat_samples=feat.detach().flatten()[::997].numpy(),
                        param_names=np.array(sorted(sd.keys())), param_shapes=np.array([str(sd[k].shape) for k in sorted(sd.keys())]))
    print("resnet18", tuple(low.shape), tuple(feat.shape))


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


if __name__ == "__main__":
    main()
