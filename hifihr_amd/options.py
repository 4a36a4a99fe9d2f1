"""Config surface of the hot path: the argparse defaults the reference's step reads
(reference options/train_options.py:4-201) plus the JSON overlay and lambda-list handling of
reference train_hrnet.py:503-519.  Only keys that select or weight the hot path are kept."""
from __future__ import annotations

import argparse
import json

_DEFAULTS = dict(
    new_model=True, render=True, light_estimation=True, four_channel=False, hand_model="mano", pretrain="res18",
    use_mean_shape=False, base_loss_fn="L1", losses=["joint_3d", "vert_3d", "mpose", "mshape", "edge_length", "sil",
                                                      "texture", "mrgb", "ssim_tex"],
    train_batch=32, val_batch=16, num_workers=8, init_lr=0.001, force_init_lr=-1, lr_steps=[50], lr_gamma=0.001,
    optimizer="Adam", total_epochs=100, semi_ratio=None,
    lambda_texture=0.003, lambda_silhouette=0.005, lambda_j2d_gt_list=[0.00001], lambda_j2d_gt_steps=[],
    lambda_j3d=100.0, lambda_vert_3d=100.0, lambda_shape_list=[0.00001], lambda_shape_steps=[],
    lambda_pose_list=[0.0001], lambda_pose_steps=[], lambda_tex_reg_list=[0.00001], lambda_tex_reg_steps=[],
    lambda_mrgb=1e-3, lambda_iou=1e-3, lambda_bone_direc=0.1, lambda_bone_direc_3d=0.1, lambda_edge_len=0.1,
    lambda_percep=1e-5, lambda_ssim_tex=0.001, lambda_scale=100.0, lambda_mscale=0.1, lambda_laplacian=0.1,
    ROOT=9, ROOT_NIMBLE=11,
)

# lambda values of reference config/FreiHAND/full_rhd_freihand.json, used by the
# BASELINE config-2 composition (res18 + MANO + render) defined in SURVEY.md section 8(d)
FREIHAND_FULL_LAMBDAS = dict(
    lambda_j3d=200, lambda_vert_3d=150, lambda_edge_len=100, lambda_silhouette=0.0008,
    lambda_tex_reg_steps=[260, 270], lambda_tex_reg_list=[5e-3, 5e-5, 5e-7], lambda_pose_steps=[10, 20],
    lambda_pose_list=[0.01, 0.001, 0.00001], lambda_shape_steps=[270, 300], lambda_shape_list=[0.1, 0.001, 0.00001],
    lambda_percep=1e-8, lambda_mrgb=2e-3, lambda_ssim_tex=0.01, lambda_texture=0.02, lambda_bone_direc_3d=6,
    init_lr=0.001, force_init_lr=0.000001, lr_steps=[80, 160, 200, 300, 350, 400, 450], lr_gamma=0.5,
)


def finalize(args):
    """train_hrnet.py:516-519: lambda_* <- first element of *_list."""
    args.lambda_pose = args.lambda_pose_list[0]
    args.lambda_j2d_gt = args.lambda_j2d_gt_list[0]
    args.lambda_shape = args.lambda_shape_list[0]
    args.lambda_tex_reg = args.lambda_tex_reg_list[0]
    return args


def update_lambdas_for_epoch(args, epoch):
    """train_hrnet.py:454-465: step schedules of lambda_pose / _j2d_gt / _shape / _tex_reg."""
    for name in ("pose", "j2d_gt", "shape", "tex_reg"):
        steps, vals = getattr(args, f"lambda_{name}_steps"), getattr(args, f"lambda_{name}_list")
        idx = sum(1 for s in steps if epoch >= s)
        setattr(args, f"lambda_{name}", vals[min(idx, len(vals) - 1)])
    return args


def make_args(config_json: str | None = None, **overrides) -> argparse.Namespace:
    args = argparse.Namespace(**{k: (list(v) if isinstance(v, list) else v) for k, v in _DEFAULTS.items()})
    if config_json:
        with open(config_json) as fh:
            for k, v in json.load(fh).items():          # unknown keys accepted silently, like train_hrnet.py:505-510
                setattr(args, k, v)
    for k, v in overrides.items():
        setattr(args, k, v)
    return finalize(args)


def baseline_config2_args(**overrides):
    """BASELINE.json configs[1]: FreiHAND batch=32, ResNet-18 + MANO LBS + silhouette/texture losses."""
    kw = dict(FREIHAND_FULL_LAMBDAS)
    kw.update(overrides)
    return make_args(None, **kw)


# reference config/FreiHAND/full_rhd_freihand.json ("losses", "train_batch", "pretrain"; hand_model is "nimble" there)
FREIHAND_FULL_LOSSES = ["joint_3d", "vert_3d", "mpose", "mshape", "mtex", "bone_direc_3d", "edge_length",
                        "texture", "mrgb", "sil", "ssim_tex", "perceptual"]


def baseline_config3_args(**overrides):
    """BASELINE.json configs[2]: full_rhd_freihand.json -- EfficientNet-b3, batch 48, texture + perceptual losses.  The
    NIMBLE layer is not available: the composition runs with MANO + the texture stand-in of models.Model
    and a VGG19 carrying seeded random weights."""
    kw = dict(FREIHAND_FULL_LAMBDAS)
    kw.update(pretrain="effb3", train_batch=48, losses=list(FREIHAND_FULL_LOSSES), hand_model="mano")
    kw.update(overrides)
    return make_args(None, **kw)


# reference config/HO3D/weak_rhd_ho3d.json
HO3D_WEAK_LAMBDAS = dict(
    lambda_j3d=200, lambda_vert_3d=200, lambda_edge_len=100, lambda_silhouette=0.00001, lambda_iou=0.0008,
    lambda_tex_reg_steps=[260, 270], lambda_tex_reg_list=[5e-3, 5e-5, 5e-7], lambda_pose_steps=[260, 270, 280],
    lambda_pose_list=[0.01, 0.001, 0.0001, 0.00001], lambda_shape_steps=[270, 300], lambda_shape_list=[0.1, 0.001, 0.00001],
    lambda_j2d_gt_list=[0.01], lambda_bone_direc=0.1, lambda_percep=1e-8, lambda_ssim_tex=0.002, lambda_texture=0.01,
    init_lr=0.001, lr_steps=[80, 160, 200, 300, 350, 400, 450], lr_gamma=0.5,
)
HO3D_WEAK_LOSSES = ["joint_2d", "bone_direc", "mpose", "mshape", "mtex", "texture", "mrgb", "sil", "ssim_tex", "perceptual"]


def baseline_config5_args(**overrides):
    """BASELINE.json configs[4]: HO-3D weak supervision (2-D joints + photometric terms), 16 images per GPU.  Same
    stand-ins as `baseline_config3_args`; dat_name is "HO3D" (no scale term, absolute ground truth, OpenGL-convention
    intrinsics flipped by data_dic)."""
    kw = dict(HO3D_WEAK_LAMBDAS)
    kw.update(pretrain="effb3", train_batch=16, losses=list(HO3D_WEAK_LOSSES), hand_model="mano")
    kw.update(overrides)
    return make_args(None, **kw)
