"""ORACLE (test infrastructure, not product code): the whole training step on the CPU.

Composes the oracle pieces -- oracle/torch_modules.py (encoder / heads / VGG restated in plain torch), oracle/mano_oracle.py,
oracle/render_oracle.py + oracle/raster_oracle.c, oracle/loss_oracle.py -- following reference models_res_nimble.py:102-225
(Model.forward), train_hrnet.py:50-113 (the step) and losses.py:234-453.  Independent of the product: nothing under oracle/ imports
`hifihr_amd` (tests/test_host_logic.py::test_oracle_is_independent_of_the_product checks it).  Used by tests/ (end-to-end loss
parity of the HIP path), by __graft_entry__.smoke() and as bench.py's cpu_baseline ("port").
"""
from __future__ import annotations

import torch
import torch.nn as nn

from oracle import mano_oracle as mo
from oracle import render_oracle as ro
from oracle.loss_oracle import LossFunctionRef, trans_proj_j2d
from oracle.torch_modules import EffiEncoderRef, HandEncoderRef, LightEstimatorRef, ResEncoderRef

SKIN_TONE = (0.78, 0.60, 0.50)


def texture_stand_in_basis(ncomp: int) -> torch.Tensor:
    """The fixed seeded map texture_params -> per-vertex colour offsets of the declared NIMBLE texture stand-in (DESIGN.md section 2,
    A9): 0.05 * randn(ncomp, 778 * 3) from generator seed 7.  A data definition, restated here so that the oracle does not import
    the product; tests check that both sides hold the same numbers."""
    gen = torch.Generator().manual_seed(7)
    return 0.05 * torch.randn(int(ncomp), 778 * 3, generator=gen)


class OracleModel(nn.Module):
    def __init__(self, tables, pretrain="res18", image_size=224, aa=3, root_id=9, texture_stand_in=0):
        super().__init__()
        self.tables, self.image_size, self.aa, self.root_id = tables, image_size, aa, root_id
        if pretrain in ("res18", "res50", "res101"):
            self.base_encoder = ResEncoderRef(pretrain=pretrain)
            feat_dim, low_dim = (512, 128) if pretrain == "res18" else (2048, 512)
        else:
            self.base_encoder, feat_dim, low_dim = EffiEncoderRef(pretrain), 1536, 32
        self.hand_encoder = HandEncoderRef(hand_model="mano", ncomps=[10, 48, int(texture_stand_in) or None], in_dim=feat_dim,
                                           ifRender=True, use_mean_shape=False)
        if texture_stand_in:                   # vertex colours = skin tone + basis . texture_params
            self.register_buffer("texture_basis", texture_stand_in_basis(texture_stand_in), persistent=False)
        self.texture_stand_in = int(texture_stand_in)
        self.light_estimator = LightEstimatorRef(low_dim)
        self.faces = torch.as_tensor(tables.faces).long()

    def forward(self, dat_name, mode_train, images, Ks=None, root_xyz=None):
        low, feat = self.base_encoder(images)
        return self.forward_from_features(dat_name, mode_train, images, low, feat, Ks=Ks, root_xyz=root_xyz)

    def forward_from_features(self, dat_name, mode_train, images, low, feat, Ks=None, root_xyz=None):
        light = self.light_estimator(low)
        hp = self.hand_encoder(feat)
        verts, _, _ = mo.mano_forward(self.tables, hp["pose_params"], hp["shape_params"])
        outputs = {"mano_verts": verts}
        outputs.update(hp)
        joints = mo.xyz_from_vertice(self.tables, verts)
        joints, mano_verts, pred_root = mo.root_relative(joints, verts, 0 if (dat_name == "HO3D" and not mode_train) else self.root_id)
        outputs["joints"], outputs["mano_verts"] = joints, mano_verts
        cam = ro.ndc_camera_from_K(Ks, float(self.image_size))
        verts_cam = mano_verts + root_xyz
        vcol = torch.tensor(SKIN_TONE).repeat(images.shape[0], 778, 1)
        if self.texture_stand_in:
            vcol = vcol + (hp["texture_params"] @ self.texture_basis).view(-1, 778, 3)
        rgba, p2f = ro.render(verts_cam, vcol, cam, light["colors"], light["directions"], self.faces,
                              image_size=self.image_size, aa=self.aa)
        re_img, re_sil, mask_rgbs = ro.model_render_outputs(rgba, images)
        outputs.update({"re_img": re_img, "re_sil": re_sil, "maskRGBs": mask_rgbs, "face_id": torch.from_numpy(p2f)})
        outputs["mano_faces"] = torch.as_tensor(self.tables.faces, dtype=torch.int16).unsqueeze(0).repeat(images.shape[0], 1, 1)
        return outputs


def oracle_step(model: OracleModel, examples_cpu: dict, args, optimizer=None, features=None, dat_name="FreiHand", perceptual=None):
    """train_hrnet.py:50-113 on the CPU.  Returns (loss, loss_dic, outputs).  `features=(low, feat)` skips the
    image encoder (tests isolate the HIP kernels from conv back-end rounding that way)."""
    root_xyz = examples_cpu["joints"][:, args.ROOT, :].unsqueeze(1)
    if features is None:
        outputs = model(dat_name, True, examples_cpu["imgs"], Ks=examples_cpu["Ps"], root_xyz=root_xyz)
    else:
        outputs = model.forward_from_features(dat_name, True, examples_cpu["imgs"], features[0], features[1],
                                              Ks=examples_cpu["Ps"], root_xyz=root_xyz)
    ex = dict(examples_cpu)
    if dat_name != "HO3D":                 # train_hrnet.py:64-68
        ex["joints"] = examples_cpu["joints"] - root_xyz
        if "verts" in examples_cpu:
            ex["verts"] = examples_cpu["verts"] - root_xyz
    outputs["j2d"] = trans_proj_j2d(outputs, examples_cpu["Ks"], root_xyz=root_xyz)
    loss_dic = LossFunctionRef(perceptual)(ex, outputs, args.losses, dat_name, args)
    loss = sum(loss_dic[k] for k in args.losses)
    if optimizer is not None:
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    return loss, loss_dic, outputs


_MANO_TO_FREI = {0: 0, 1: 5, 2: 6, 3: 7, 4: 8, 5: 9, 6: 10, 7: 11, 8: 12, 9: 17, 10: 18, 11: 19, 12: 20, 13: 13, 14: 14, 15: 15, 16: 16,
                 17: 1, 18: 2, 19: 3, 20: 4}


def mano2frei(mano_joints):
    """reference utils/fh_utils.py:542-556."""
    out = torch.zeros_like(mano_joints)
    for mano_id, frei_id in _MANO_TO_FREI.items():
        out[:, frei_id] = mano_joints[:, mano_id]
    return out


def nimble_forward_tail(t, hand_params, images, Ks, root_xyz, light, dat_name="FreiHand", mode_train=True, root_id=9, root_id_nimble=11,
                        image_size=224, aa=3, point_lights=False):
    """Model.forward after the heads for hand_model == 'nimble' (reference models_res_nimble.py:133-225) on the NIMBLE-shaped layer of
    oracle/lbs_oracle.py: Mano2Frei, root-relative outputs, the skin mesh rendered with its per-vertex texture."""
    from oracle import lbs_oracle as lo
    out = lo.nimble_layer(t, hand_params["pose_params"], hand_params["shape_params"], hand_params.get("texture_params"))
    out.update(hand_params)
    out["joints"] = mano2frei(out["joints"])
    eval_ho3d = dat_name == "HO3D" and not mode_train
    pred_root = out["joints"][:, 0 if eval_ho3d else root_id].unsqueeze(1)
    out["joints"], out["mano_verts"] = out["joints"] - pred_root, out["mano_verts"] - pred_root
    nroot = out["nimble_joints"][:, 0 if eval_ho3d else root_id_nimble].unsqueeze(1)
    out["nimble_joints"] = out["nimble_joints"] - nroot
    cam = ro.ndc_camera_from_K(Ks, float(image_size))
    verts_cam = out["verts"] - pred_root + root_xyz
    uv = None
    if getattr(t, "faces_uvs", None) is not None and getattr(t, "tex_img_basis", None) is not None:
        # TexturesUV (models_res_nimble.py:203-208): the texture IMAGE of the layer's texture PCA, sampled through per-face uvs
        TH, TW = t.tex_hw
        maps = (torch.as_tensor(t.tex_img_mean) + hand_params["texture_params"] @ torch.as_tensor(t.tex_img_basis)).reshape(-1, TH, TW, 3)
        uv = (maps, torch.as_tensor(t.faces_uvs).long(), torch.as_tensor(t.verts_uvs))
        out["texture_maps"] = maps
    rgba, p2f = ro.render(verts_cam, out["textures"], cam, light["colors"], light["directions"], torch.as_tensor(t.faces).long(),
                          image_size=image_size, aa=aa, point_lights=point_lights, textures_uv=uv)
    out["re_img"], out["re_sil"], out["maskRGBs"] = ro.model_render_outputs(rgba, images)
    out["face_id"], out["skin_verts"] = torch.from_numpy(p2f), verts_cam
    return out
