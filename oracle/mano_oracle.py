"""ORACLE (test infrastructure, not product code): CPU restatement of the reference MANO layer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates, in plain torch-CPU ops so that autograd supplies the reference gradients,

  * ManoLayer.forward                      reference utils/my_mano.py:315-483
  * batch_rodrigues / quat2mat             reference utils/manopth/rodrigues_layer.py:43-54, 15-40
  * th_posemap_axisang / subtract_flat_id  reference utils/manopth/tensutils.py:6-12, 34-42
  * dense_pose_Trainer.xyz_from_vertice    reference utils/Freihand_GNN_mano/Freihand_trainer_mano_fullsup.py:175-215
  * root-relative step of Model.forward    reference models_res_nimble.py:160-166

Pinned: tests/test_oracle_mano.py checks it against tests/golden/mano_*.npz, which
tools/make_golden.py produced by running the reference's own ManoLayer in the build container.
"""
from __future__ import annotations

import torch

LEV1 = [1, 4, 7, 10, 13]
LEV2 = [2, 5, 8, 11, 14]
LEV3 = [3, 6, 9, 12, 15]
REORDER16 = [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15]         # my_mano.py:433
REORDER21 = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # my_mano.py:462-466
TIPS = [745, 317, 444, 556, 673]                                            # my_mano.py:457
# Freihand_trainer_mano_fullsup.py:177-195 : regressed joint manoId -> FreiHAND slot, tips from verts
XYZ_MAP = {0: 0, 1: 5, 2: 6, 3: 7, 4: 9, 5: 10, 6: 11, 7: 17, 8: 18, 9: 19, 10: 13, 11: 14, 12: 15,
           13: 1, 14: 2, 15: 3}
XYZ_TIPS = {4: 744, 8: 320, 12: 443, 16: 555, 20: 672}


def _t(a, dtype):
    return torch.as_tensor(a, dtype=dtype)


def quat2mat(quat):
    nq = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def batch_rodrigues(axisang):
    """[N,3] -> [N,9]; note the +1e-8 is added to the vector before the norm (rodrigues_layer.py:45)."""
    angle = torch.norm(axisang + 1e-8, p=2, dim=1).unsqueeze(-1)
    normalized = axisang / angle
    angle = angle * 0.5
    quat = torch.cat([torch.cos(angle), torch.sin(angle) * normalized], dim=1)
    return quat2mat(quat).reshape(-1, 9)


def _with_zeros(m34):
    pad = m34.new_tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).repeat(m34.shape[0], 1, 1)
    return torch.cat([m34, pad], 1)


def mano_forward(tables, pose, beta, dtype=torch.float32, center_idx=9):
    """pose [B,48] (3 global axis-angle + 45 PCA coeffs), beta [B,10] -> verts [B,778,3], jtr [B,21,3].
    Also returns intermediates used by tests."""
    comps = _t(tables.hands_components, dtype)
    mean = _t(tables.hands_mean, dtype).unsqueeze(0)
    shapedirs = _t(tables.shapedirs, dtype)
    posedirs = _t(tables.posedirs, dtype)
    v_template = _t(tables.v_template, dtype).unsqueeze(0)
    j_reg = _t(tables.J_regressor, dtype)
    weights = _t(tables.weights, dtype)
    B = pose.shape[0]

    full_hand_pose = pose[:, 3:48].mm(comps)                                   # :340
    full_pose = torch.cat([pose[:, :3], mean + full_hand_pose], 1)             # :345-348
    rot_map = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, 16 * 9)        # tensutils.py:6-12
    eye = torch.eye(3, dtype=dtype).view(1, 9).repeat(B, 16)
    pose_map = rot_map - eye
    root_rot = rot_map[:, :9].view(B, 3, 3)                                    # :352
    rot_map = rot_map[:, 9:]
    pose_map = pose_map[:, 9:]

    v_shaped = torch.matmul(shapedirs, beta.transpose(1, 0)).permute(2, 0, 1) + v_template   # :386-388
    j = torch.matmul(j_reg, v_shaped)                                          # :389
    v_posed = v_shaped + torch.matmul(posedirs, pose_map.transpose(0, 1)).permute(2, 0, 1)   # :392-393

    root_j = j[:, 0, :].contiguous().view(B, 3, 1)
    root_trans = _with_zeros(torch.cat([root_rot, root_j], 2))                 # :398-399
    all_rots = rot_map.view(B, 15, 3, 3)
    lev1_rots = all_rots[:, [i - 1 for i in LEV1]]
    lev2_rots = all_rots[:, [i - 1 for i in LEV2]]
    lev3_rots = all_rots[:, [i - 1 for i in LEV3]]
    lev1_j, lev2_j, lev3_j = j[:, LEV1], j[:, LEV2], j[:, LEV3]

    all_tf = [root_trans.unsqueeze(1)]
    lev1_rel = _with_zeros(torch.cat([lev1_rots, (lev1_j - root_j.transpose(1, 2)).unsqueeze(3)], 3).view(-1, 3, 4))
    root_flt = root_trans.unsqueeze(1).repeat(1, 5, 1, 1).view(B * 5, 4, 4)
    lev1_flt = torch.matmul(root_flt, lev1_rel)                                # :414-418
    all_tf.append(lev1_flt.view(B, 5, 4, 4))
    lev2_rel = _with_zeros(torch.cat([lev2_rots, (lev2_j - lev1_j).unsqueeze(3)], 3).view(-1, 3, 4))
    lev2_flt = torch.matmul(lev1_flt, lev2_rel)                                # :421-424
    all_tf.append(lev2_flt.view(B, 5, 4, 4))
    lev3_rel = _with_zeros(torch.cat([lev3_rots, (lev3_j - lev2_j).unsqueeze(3)], 3).view(-1, 3, 4))
    lev3_flt = torch.matmul(lev2_flt, lev3_rel)                                # :427-430
    all_tf.append(lev3_flt.view(B, 5, 4, 4))

    results = torch.cat(all_tf, 1)[:, REORDER16]                               # :433-434
    joint_js = torch.cat([j, j.new_zeros(B, 16, 1)], 2)
    tmp2 = torch.matmul(results, joint_js.unsqueeze(3))
    results2 = (results - torch.cat([tmp2.new_zeros(B, 16, 4, 3), tmp2], 3)).permute(0, 2, 3, 1)  # :437-439
    T = torch.matmul(results2, weights.transpose(0, 1))                        # :441  [B,4,4,778]
    rest_h = torch.cat([v_posed.transpose(2, 1), torch.ones((B, 1, 778), dtype=dtype)], 1)
    verts = (T * rest_h.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]         # :450-451
    jtr = results[:, :, :3, 3]
    jtr = torch.cat([jtr, verts[:, TIPS]], 1)[:, REORDER21]                    # :457-466
    if center_idx is not None:                                                 # :471-475 (th_trans == 0 branch)
        center = jtr[:, center_idx].unsqueeze(1)
        jtr = jtr - center
        verts = verts - center
    return verts, jtr, {"v_posed": v_posed, "j": j, "results": results, "full_pose": full_pose}


def xyz_from_vertice(tables, verts, dtype=torch.float32):
    """verts [B,778,3] -> FreiHAND-ordered joints [B,21,3] (the caller's .permute(1,0,2) included;
    reference models_res_nimble.py:153)."""
    j_reg_t = _t(tables.J_regressor, dtype).t()                                # [778,16]
    joints = torch.stack([verts[:, :, c].matmul(j_reg_t) for c in range(3)], dim=2)   # [B,16,3]
    out = [None] * 21
    for mano_id, slot in XYZ_MAP.items():
        out[slot] = joints[:, mano_id, :]
    for slot, vid in XYZ_TIPS.items():
        out[slot] = verts[:, vid, :]
    return torch.stack(out, dim=1)


def root_relative(joints, mano_verts, root_id=9):
    """models_res_nimble.py:160-166 (training branch): subtract the predicted root joint."""
    pred_root = joints[:, root_id, :].unsqueeze(1)
    return joints - pred_root, mano_verts - pred_root, pred_root
