"""Seeded `examples` / `outputs` dictionaries for the LossFunction parity vectors (tests/golden/loss_dict.npz).
Shared by tools/make_golden.py (which feeds them to the reference's own LossFunction.__call__ in the build container) and by the
tests (which feed the same tensors to the oracle restatement on the CPU and to the HIP path on the GPU)."""
import numpy as np
import torch

from hifihr_amd.mano_tables import synthetic_mano_tables


def loss_dict_case(name, B=3, H=64, seed=0):
    """Seeded `examples` / `outputs` dicts of one configuration (keys and shapes as train_hrnet.py:50-113 builds them)."""
    from hifihr_amd import options
    g = torch.Generator().manual_seed(1000 + seed)
    args = {"cfg2": options.baseline_config2_args, "cfg3": options.baseline_config3_args, "ho3d": options.baseline_config5_args}[name]()
    if name == "cfg3":
        args.losses = list(args.losses) + ["scale", "mscale", "iou"]      # every FreiHand-side term once
    if name == "cfg2":
        args.base_loss_fn = "L2"
    faces = torch.as_tensor(synthetic_mano_tables(0).faces.astype(np.int16)).unsqueeze(0).repeat(B, 1, 1)
    seg = (torch.rand(B, H, H, generator=g) > 0.6).float()
    sil = (torch.rand(B, 1, H, H, generator=g) > 0.55).float() * 255.0
    imgs = torch.rand(B, 3, H, H, generator=g)
    ex = {"imgs": imgs, "segms_gt": seg, "joints": 0.08 * torch.randn(B, 21, 3, generator=g), "verts": 0.08 * torch.randn(B, 778, 3, generator=g),
          "j2d_gt": 224 * torch.rand(B, 21, 2, generator=g), "scales": 0.03 + 0.01 * torch.rand(B, generator=g)}
    out = {"joints": 0.08 * torch.randn(B, 21, 3, generator=g), "mano_verts": 0.08 * torch.randn(B, 778, 3, generator=g),
           "j2d": 224 * torch.rand(B, 21, 2, generator=g), "mano_faces": faces, "re_img": torch.rand(B, 3, H, H, generator=g), "re_sil": sil,
           "pose_params": torch.randn(B, 48, generator=g), "shape_params": torch.randn(B, 10, generator=g),
           "texture_params": torch.randn(B, 10, generator=g)}
    out["maskRGBs"] = imgs * (sil > 0).float().repeat(1, 3, 1, 1)
    if name == "ho3d":
        ex["texture_con"] = 0.2 + torch.rand(B, generator=g)              # switches the self-supervised photometric terms on
    dat = "HO3D" if name == "ho3d" else "FreiHand"
    return args, ex, out, dat
