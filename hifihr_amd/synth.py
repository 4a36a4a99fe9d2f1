"""Synthetic FreiHAND-shaped training batches (the datasets are not available offline); generator per
SURVEY.md section 8(d).  Sample keys and shapes follow the reference's training queries
(reference data/dataset.py:153-289): trans_images f32[3,224,224] in [0,1), trans_Ks f32[3,3],
trans_joints f32[21,3], trans_verts f32[778,3], trans_masks f32[3,224,224] in {0,1}, scales, idxs."""
from __future__ import annotations

import math

import torch

from . import ops


@torch.no_grad()
def make_batch(mano_handle: ops.ManoLayerHandle, renderer: ops.RendererHandle, B: int, first_index: int = 0,
               device="cuda", image_size=224, images="noise"):
    """Deterministic in (first_index, B): sample i uses seed 1234 + first_index + i.
    images: "noise" (uniform noise, unrelated to the pose: timing / parity inputs) or "render" (the posed hand, lit, over dim noise:
    an image the pose can be read from, for training runs that are expected to learn)."""
    rows = []
    for i in range(B):
        g = torch.Generator().manual_seed(1234 + first_index + i)
        f = (450.0 + 200.0 * torch.rand(1, generator=g)) * (image_size / 224.0)      # focal / principal-point offset in pixels of THIS image size
        d = (20.0 * torch.rand(2, generator=g) - 10.0) * (image_size / 224.0)
        theta = (2 * torch.rand(1, generator=g) - 1) * math.pi
        pose = torch.cat([0.5 * torch.randn(3, generator=g), 0.8 * torch.randn(45, generator=g)])
        beta = 0.5 * torch.randn(10, generator=g)
        root = torch.stack([0.1 * torch.rand(1, generator=g) - 0.05, 0.1 * torch.rand(1, generator=g) - 0.05,
                            0.45 + 0.3 * torch.rand(1, generator=g)]).view(3)
        rows.append((f, d, theta, pose, beta, root, int(g.initial_seed())))
    f = torch.cat([r[0] for r in rows]); d = torch.stack([r[1] for r in rows]); th = torch.cat([r[2] for r in rows])
    pose = torch.stack([r[3] for r in rows]).to(device); beta = torch.stack([r[4] for r in rows]).to(device)
    root = torch.stack([r[5] for r in rows]).to(device)
    c = image_size / 2
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = f; K[:, 1, 1] = f; K[:, 0, 2] = c + d[:, 0]; K[:, 1, 2] = c + d[:, 1]; K[:, 2, 2] = 1
    # in-plane rotation about the image centre, as reference data/dataset.py:237-260 (post_rot_trans . K)
    cs, sn = torch.cos(th), torch.sin(th)
    R = torch.zeros(B, 3, 3)
    R[:, 0, 0] = cs; R[:, 0, 1] = -sn; R[:, 1, 0] = sn; R[:, 1, 1] = cs; R[:, 2, 2] = 1
    T = torch.eye(3).repeat(B, 1, 1); T[:, 0, 2] = c; T[:, 1, 2] = c
    Ti = torch.eye(3).repeat(B, 1, 1); Ti[:, 0, 2] = -c; Ti[:, 1, 2] = -c
    K = (T @ R @ Ti @ K).to(device)
    # the rotated intrinsics are no longer upper-triangular; the reference rotates the 3-D points by R_z as well
    # (data/dataset.py:271-280) so that K stays a plain pinhole.  Do the same: rotate points, keep pinhole K.
    Kp = torch.zeros(B, 3, 3, device=device)
    Kp[:, 0, 0] = f.to(device); Kp[:, 1, 1] = f.to(device); Kp[:, 2, 2] = 1
    Kp[:, 0, 2] = K[:, 0, 2]; Kp[:, 1, 2] = K[:, 1, 2]
    verts, _ = ops.mano_lbs(mano_handle, pose, beta)
    joints, verts_rel, _ = ops.mano_joints_root_relative(mano_handle, verts, 9)
    Rz = R.to(device)
    verts_w = torch.einsum("bij,bvj->bvi", Rz, verts_rel) + root.unsqueeze(1)
    joints_w = torch.einsum("bij,bvj->bvi", Rz, joints) + root.unsqueeze(1)
    s = float(image_size)
    cam = torch.stack([-(Kp[:, 0, 0] * 2 / s), -(Kp[:, 1, 1] * 2 / s), -(Kp[:, 0, 2] - s / 2) * 2 / s,
                       -(Kp[:, 1, 2] - s / 2) * 2 / s], dim=-1).contiguous()
    lc = torch.zeros(B, 3, device=device); ld = torch.tensor([0.0, 0.0, -1.0], device=device).repeat(B, 1)
    col = torch.ones(778, 3, device=device)
    rgba, _ = ops.render(renderer, verts_w.contiguous(), col, cam, lc, ld)
    mask = (rgba[:, 3:4] > 0).float().repeat(1, 3, 1, 1)
    gi = torch.Generator(device="cpu").manual_seed(1234 + first_index)
    imgs = torch.rand(B, 3, image_size, image_size, generator=gi)
    if images == "render":
        lit, _ = ops.render(renderer, verts_w.contiguous(), col, cam, torch.full((B, 3), 0.7, device=device),
                            torch.tensor([0.35, 0.35, -0.87], device=device).repeat(B, 1))
        a = mask.cpu()
        imgs = a * lit[:, :3].clamp(0, 1).cpu() + (1 - a) * 0.25 * imgs
    elif images != "noise":
        raise ValueError(f"images={images!r}")
    scales = (joints_w[:, 9] - joints_w[:, 10]).norm(dim=-1)
    return {
        "trans_images": imgs, "trans_Ks": Kp.cpu(), "trans_joints": joints_w.cpu(), "trans_verts": verts_w.cpu(),
        "trans_masks": mask.cpu(), "scales": scales.cpu(), "idxs": torch.arange(first_index, first_index + B),
    }


def to_ho3d_sample(sample: dict, crop: int = 448) -> dict:
    """The same synthetic batch in the HO-3D loader's keys and conventions (what utils/traineval_util.py:156-201 undoes):
    OpenGL camera (columns 2-3 of K and the joints' y / z negated), HO-3D joint order, a `crop`-pixel image crop that
    nearest-neighbour resizing brings back to 224 (crop = 2 x 224: every pixel repeated 2 x 2)."""
    from .traineval import Frei2HO3D, proj_func
    base = sample["trans_images"].shape[-1]
    assert crop % base == 0
    r = crop // base
    flip = torch.tensor([1.0, -1.0, -1.0])
    K, joints = sample["trans_Ks"], sample["trans_joints"]
    return {
        "img_crop": sample["trans_images"].repeat_interleave(r, dim=2).repeat_interleave(r, dim=3),
        "K_crop": K * flip.view(1, 1, 3),
        "xyz21": Frei2HO3D(joints * flip.view(1, 1, 3)),
        "uv21_crop": Frei2HO3D(proj_func(joints, K)),
        "root_xyz": joints[:, 0] * flip,
        "hand_mask_crop": sample["trans_masks"],
        "idxs": sample["idxs"],
    }


def make_ho3d_frames(mano_handle, renderer, n: int, first_index: int = 0, device="cuda", frame_hw=(480, 640), images="noise"):
    """Synthetic HO-3D-shaped raw frames for `data.HO3DDeviceCache`: the 224 x 224 synthetic hand of `make_batch` pasted into a 480 x 640
    frame at a per-sample offset (grey background), the intrinsics moved with it, everything in the HO-3D loader's conventions
    (reference data/dataset.py:1065-1068, 1093: `Ks = camMat . cam_extr` with the OpenGL flip, joints in the HO-3D order with y / z negated --
    what utils/traineval_util.py:156-201 undoes).  -> images_u8 [n,480,640,3], hand_masks_u8 [n,480,640], Ks [n,3,3], xyz21 [n,21,3] (host)."""
    from .traineval import Frei2HO3D
    FH, FW = frame_hw
    imgs = torch.full((n, FH, FW, 3), 96, dtype=torch.uint8)
    masks = torch.zeros(n, FH, FW, dtype=torch.uint8)
    Ks, xyz = torch.zeros(n, 3, 3), torch.zeros(n, 21, 3)
    flip = torch.tensor([1.0, -1.0, -1.0])
    step = 32
    for lo in range(0, n, step):
        m = min(step, n - lo)
        b = make_batch(mano_handle, renderer, m, first_index=first_index + lo, device=device, images=images)
        im = (b["trans_images"].clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu()
        mk = (b["trans_masks"][:, 0] > 0.5).to(torch.uint8).mul(255).cpu()
        K, J = b["trans_Ks"].cpu(), b["trans_joints"].cpu()
        for i in range(m):
            g = torch.Generator().manual_seed(4321 + first_index + lo + i)
            oy = int(torch.randint(0, FH - 224 + 1, (1,), generator=g)); ox = int(torch.randint(0, FW - 224 + 1, (1,), generator=g))
            imgs[lo + i, oy:oy + 224, ox:ox + 224] = im[i]
            masks[lo + i, oy:oy + 224, ox:ox + 224] = mk[i]
            Kf = K[i].clone(); Kf[0, 2] += ox; Kf[1, 2] += oy
            Ks[lo + i] = Kf * flip.view(1, 3)
            xyz[lo + i] = Frei2HO3D((J[i] * flip.view(1, 3)).unsqueeze(0))[0]
    return {"images_u8": imgs, "hand_masks_u8": masks, "Ks": Ks, "xyz21": xyz}
