"""Evaluation pass of the reference's loop (reference train_hrnet.py:119-161 collection + texture metrics, :216-272
MPJPE / MPVPE after Procrustes alignment; utils/train_utils.py:267-290 align_w_scale) with the alignment batched on the
device (csrc/eval.hip) instead of a per-sample numpy / scipy loop.  SURVEY.md section 8(f) N2.

LPIPS (train_hrnet.py:156) needs AlexNet weights that cannot be downloaded here: reported as None unless the caller
passes an `lpips_fn`."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ._lib import get_lib, require_cuda
from .traineval import Frei2HO3D


def align_w_scale(mtx1, mtx2, return_error=False):
    """utils/train_utils.py:267-290 for batches: aligns mtx2 [B,N,3] (prediction) to mtx1 [B,N,3] (ground truth) with
    translation, isotropic scale and an orthogonal map (reflections allowed, like scipy's orthogonal_procrustes).
    Returns the aligned mtx2 (and, with return_error, the per-sample mean point distance to mtx1)."""
    require_cuda(mtx1, mtx2)
    gt, pred = mtx1.float().contiguous(), mtx2.float().contiguous()
    assert gt.shape == pred.shape and gt.dim() == 3 and gt.shape[2] == 3, (gt.shape, pred.shape)
    aligned = torch.empty_like(pred)
    err = torch.empty(pred.shape[0], device=pred.device)
    get_lib().procrustes_error(pred, gt, aligned, err)
    return (aligned, err / pred.shape[1]) if return_error else aligned


def aligned_error(pred, gt):
    """Mean per-point distance after alignment, one value per sample (no aligned copy is written)."""
    require_cuda(pred, gt)
    pred, gt = pred.float().contiguous(), gt.float().contiguous()
    err = torch.empty(pred.shape[0], device=pred.device)
    get_lib().procrustes_error(pred, gt, None, err)
    return err / pred.shape[1]


class Evaluator:
    """Accumulates what the evaluation loop keeps (train_hrnet.py:119-161) and reduces it as :216-272 does."""

    def __init__(self, ssim_fn=None, lpips_fn=None):
        self.xyz_pred, self.verts_pred, self.texture = [], [], []
        if ssim_fn is None:
            from . import ops
            ssim_fn = ops.ssim
        self.ssim_fn, self.lpips_fn = ssim_fn, lpips_fn

    def collect(self, outputs, examples, dat_name, render=True):
        joints = outputs["joints"].detach()
        if dat_name == "HO3D":           # back to the HO-3D joint order and OpenGL axes for the challenge dump (:128-132)
            joints = Frei2HO3D(joints) * torch.tensor([1.0, -1.0, -1.0], device=joints.device).view(1, 1, 3)
        self.xyz_pred.append(joints)
        self.verts_pred.append(outputs["mano_verts"].detach())
        if render and "re_img" in outputs:
            if dat_name == "HO3D":
                m = (outputs["re_sil"] > 0).float()
                mask_rgbs, mask_re = examples["imgs"] * m, outputs["re_img"] * m
            else:
                m = examples["segms_gt"].unsqueeze(1).float()
                mask_rgbs, mask_re = m * examples["imgs"], outputs["re_img"] * m
            mse = F.mse_loss(mask_re, mask_rgbs)
            rec = {"psnr": -10 * mse.log10(), "ssim": self.ssim_fn(mask_re.contiguous(), mask_rgbs.contiguous()),
                   "l1": F.l1_loss(mask_re, mask_rgbs), "l2": mse}
            if self.lpips_fn is not None:
                rec["lpips"] = self.lpips_fn(mask_re * 2 - 1, mask_rgbs * 2 - 1).mean()
            self.texture.append(rec)              # device scalars: one host sync at `summary`, not one per batch

    def dump(self, pred_out_path):
        """utils/train_utils.py:242-254: `[xyz_pred_list, verts_pred_list]` as JSON -- the file the HO-3D challenge server takes
        (train_hrnet.py:277-293; HO-3D joints were already put back into the HO-3D order / OpenGL axes by `collect`) and the
        reference's `pred.json` for FreiHAND."""
        import json
        import os
        xyz = torch.cat(self.xyz_pred).cpu().tolist() if self.xyz_pred else []
        verts = torch.cat(self.verts_pred).cpu().tolist() if self.verts_pred else []
        os.makedirs(os.path.dirname(os.path.abspath(pred_out_path)), exist_ok=True)
        with open(pred_out_path, "w") as fo:
            json.dump([xyz, verts], fo)
        return len(xyz), len(verts)

    def summary(self, xyz_gt=None, verts_gt=None):
        """xyz_gt [n,21,3] / verts_gt [n,778,3]: evaluation_xyz.json / evaluation_verts.json.  Returns a dict with
        'pose_3d' / 'vert_3d' (MPJPE / MPVPE in the inputs' unit, metres in FreiHAND; the reference prints x100 = cm) and
        the batch-averaged texture metrics."""
        out = {}
        if xyz_gt is not None:
            pred = torch.cat(self.xyz_pred)
            out["pose_3d"] = float(aligned_error(pred, torch.as_tensor(xyz_gt, dtype=torch.float32).to(pred.device)).mean())
        if verts_gt is not None:
            pred = torch.cat(self.verts_pred)
            out["vert_3d"] = float(aligned_error(pred, torch.as_tensor(verts_gt, dtype=torch.float32).to(pred.device)).mean())
        if self.texture:
            for k in self.texture[0]:
                out[k] = float(torch.stack([r[k] for r in self.texture]).mean())
            out.setdefault("lpips", None)
        return out
