"""ORACLE (test infrastructure, not product code): generic linear-blend skinning and the NIMBLE-shaped hand layer, plain torch on the CPU.

The formulation is ManoLayer's (reference utils/my_mano.py:386-451) without pose-corrective blend shapes, for any mesh size and
kinematic tree: the reference's own NIMBLE layer is an un-vendored submodule, so this oracle is checked against
oracle/mano_oracle.py on MANO-shaped tables (tests/test_oracle_lbs.py) and is otherwise "parity unpinned" for NIMBLE itself.
Differentiable through torch autograd; no imports from the product.
"""
from __future__ import annotations

import torch


def rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """axis-angle [N,3] -> rotation matrices [N,3,3] through the quaternion (reference utils/manopth/rodrigues_layer.py:15-54)."""
    l1 = torch.norm(theta + 1e-8, p=2, dim=1)
    angle = l1.unsqueeze(-1)
    normalized = theta / angle
    half = angle * 0.5
    quat = torch.cat([torch.cos(half), torch.sin(half) * normalized], dim=1)
    quat = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def lbs_forward(v_template, shapedirs, j_regressor, weights, parents, theta, beta):
    """theta [B,J,3], beta [B,S] -> verts [B,V,3], posed joints [B,J,3] (my_mano.py:386-451 with th_pose_map dropped)."""
    v_template, shapedirs, j_regressor, weights = (torch.as_tensor(a, dtype=torch.float32) for a in (v_template, shapedirs, j_regressor, weights))
    B, J = theta.shape[0], theta.shape[1]
    v_shaped = v_template.unsqueeze(0) + torch.einsum("vck,bk->bvc", shapedirs, beta)
    joints = torch.einsum("jv,bvc->bjc", j_regressor, v_shaped)
    R = rodrigues(theta.reshape(-1, 3)).view(B, J, 3, 3)
    Rg, tg = [R[:, 0]], [joints[:, 0]]
    for i in range(1, J):
        p = int(parents[i])
        Rg.append(Rg[p] @ R[:, i])
        tg.append((Rg[p] @ (joints[:, i] - joints[:, p]).unsqueeze(-1)).squeeze(-1) + tg[p])
    Rg, tg = torch.stack(Rg, 1), torch.stack(tg, 1)                                   # [B,J,3,3], [B,J,3]
    t_rel = tg - (Rg @ joints.unsqueeze(-1)).squeeze(-1)                              # A_j = [Rg | tg - Rg J]
    Rv = torch.einsum("vj,bjrc->bvrc", weights, Rg)
    tv = torch.einsum("vj,bjr->bvr", weights, t_rel)
    verts = (Rv @ v_shaped.unsqueeze(-1)).squeeze(-1) + tv
    return verts, tg


def nimble_layer(t, pose_params, shape_params, texture_params=None):
    """The NIMBLE-shaped layer of hifihr_amd.models.MyNIMBLELayer restated: PCA pose decode -> LBS -> the MANO-topology regression, the
    21 MANO-ordered joints, per-vertex colours.  `t` is any object with the NimbleTables fields (numpy arrays)."""
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32)
    B, J = pose_params.shape[0], t.weights.shape[1]
    theta = (f32(t.pose_mean) + pose_params @ f32(t.pose_basis)).view(B, J, 3)
    verts, joints = lbs_forward(t.v_template, t.shapedirs, t.J_regressor, t.weights, t.parents, theta, shape_params)
    corner = torch.as_tensor(t.faces, dtype=torch.long)[torch.as_tensor(t.mano_vreg_fidx, dtype=torch.long)]          # [778,3]
    mano_verts = (verts[:, corner] * f32(t.mano_vreg_bc).view(1, 778, 3, 1)).sum(2)
    out = {"nimble_joints": joints, "verts": verts, "mano_verts": mano_verts,
           "joints": joints[:, torch.as_tensor(t.joint21, dtype=torch.long)], "rot": None}
    if texture_params is not None:
        out["textures"] = (f32(t.tex_mean) + texture_params @ f32(t.tex_basis)).view(B, -1, 3)
    return out
