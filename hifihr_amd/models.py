"""`Model` with the reference's constructor signature, forward signature and output-dict keys
(reference models_res_nimble.py:32-225), assembled from this package's HIP ops.

hand_model == 'mano' with render=True is a composition the reference cannot run (MyMANOLayer
builds a Meshes without textures); this build defines it with a constant per-vertex skin colour as the
TexturesVertex stand-in.  hand_model == 'nimble' needs the un-vendored NIMBLE submodule + assets and is
not built ("parity unpinned").  The NIMBLE configurations (full_rhd_freihand.json,
weak_rhd_ho3d.json) run here as "MANO + a texture stand-in" (SURVEY.md section 8(c)): `texture_stand_in=T` adds the
T-component texture head of the NIMBLE HandEncoder and turns it into per-vertex colours through a fixed seeded linear
basis (skin tone + basis . texture_params), so `texture_params`, the `mtex` term and the texture gradient exist.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .mano_tables import ManoTables, synthetic_mano_tables
from .network import HandEncoder, LightEstimator, ResEncoder

SKIN_TONE = (0.78, 0.60, 0.50)


def texture_stand_in_basis(ncomp: int) -> torch.Tensor:
    """[ncomp, 778*3] fixed linear map texture_params -> per-vertex colour offsets of the NIMBLE texture stand-in."""
    gen = torch.Generator().manual_seed(7)
    return 0.05 * torch.randn(int(ncomp), 778 * 3, generator=gen)


class MyMANOLayer(nn.Module):
    """reference utils/my_mano.py:22-54.  Returns mano_verts (and the face table instead of a pytorch3d Meshes)."""

    def __init__(self, ifRender, device, shape_ncomp=10, pose_ncomp=48, tex_ncomp=None, tables: ManoTables | None = None):
        super().__init__()
        self.tables = tables if tables is not None else synthetic_mano_tables(0)
        self.handle = ops.ManoLayerHandle(self.tables)
        self.register_buffer("mesh_face", torch.as_tensor(self.tables.faces, dtype=torch.int32).unsqueeze(0), persistent=False)

    def forward(self, hand_params, handle_collision=True):
        verts, _ = ops.mano_lbs(self.handle, hand_params["pose_params"], hand_params["shape_params"])
        return {"mano_verts": verts, "skin_verts": verts}


class Model(nn.Module):
    def __init__(self, ifRender, device, if_4c, hand_model, use_mean_shape, pretrain, root_id=9, root_id_nimble=11,
                 ifLight=True, mano_tables: ManoTables | None = None, image_size=224, aa_factor=3, texture_stand_in=0):
        super().__init__()
        if hand_model != "mano":
            raise NotImplementedError(f"hand_model='{hand_model}': only 'mano' is built (NIMBLE assets are not available)")
        self.hand_model, self.root_id, self.root_id_nimble = hand_model, root_id, root_id_nimble
        if pretrain in ("res18", "res50", "res101"):
            # res18: SURVEY.md F6 (the reference hard-codes 2048 / 512, the ResNet-50 / -101 widths, and is broken for res18)
            self.features_dim, self.low_feat_dim = (512, 128) if pretrain == "res18" else (2048, 512)
            self.base_encoder = ResEncoder(pretrain=pretrain, if_4c=if_4c)
        elif pretrain == "effb3":                                    # models_res_nimble.py:50-53
            from .effnet import EffiEncoder
            self.features_dim, self.low_feat_dim = 1536, 32
            self.base_encoder = EffiEncoder(pretrain=pretrain)
        else:
            raise NotImplementedError(f"pretrain='{pretrain}' is not built yet")
        self.ncomps = [10, 48, int(texture_stand_in) if texture_stand_in else None]
        self.hand_layer = MyMANOLayer(ifRender, device, shape_ncomp=10, pose_ncomp=48, tables=mano_tables)
        self.hand_encoder = HandEncoder(hand_model=hand_model, ncomps=self.ncomps, in_dim=self.features_dim,
                                        ifRender=ifRender, use_mean_shape=use_mean_shape)
        self.register_buffer("mano_face", self.hand_layer.mesh_face.clone().to(torch.int16), persistent=False)
        self.ifRender, self.ifLight, self.aa_factor, self.image_size = ifRender, ifLight, aa_factor, image_size
        if ifRender:
            # Materials(diffuse .8, specular .2, shininess 30) + DirectionalLights defaults (ambient .5, specular .2)
            if not ifLight:
                self.register_buffer("_pl_color", torch.tensor([[0.3, 0.3, 0.3]]), persistent=False)
                self.register_buffer("_pl_location", torch.tensor([[0.0, 1.0, 0.0]]), persistent=False)
            self.renderer_p3d = ops.RendererHandle(self.hand_layer.tables.faces, 778, image_size=image_size, aa=aa_factor, point_lights=not ifLight,
                                                   ambient=(0.5,) * 3, mat_diffuse=(0.8,) * 3, specular=(0.04,) * 3,
                                                   shininess=30.0, background=(1.0,) * 3)
            self.register_buffer("vertex_colors", torch.tensor(SKIN_TONE).repeat(778, 1), persistent=False)
            if texture_stand_in:
                self.register_buffer("texture_basis", texture_stand_in_basis(texture_stand_in), persistent=False)
                pad = (-778 * 3) % 4                                                # the decode kernel moves float4
                self.register_buffer("texture_basis_pad", torch.nn.functional.pad(self.texture_basis, (0, pad)).contiguous(), persistent=False)
                self.register_buffer("texture_mean_pad", torch.nn.functional.pad(self.vertex_colors.reshape(-1), (0, pad)).contiguous(), persistent=False)
        if ifLight:
            self.light_estimator = LightEstimator(self.low_feat_dim)

    def get_ndc_fx_fy_cx_cy(self, Ks):
        s = float(self.image_size)
        focal = torch.stack([Ks[:, 0, 0] * 2 / s, Ks[:, 1, 1] * 2 / s], dim=-1)
        prp = torch.stack([-(Ks[:, 0, 2] - s / 2) * 2 / s, -(Ks[:, 1, 2] - s / 2) * 2 / s], dim=-1)
        return focal, prp

    def camera_from_K(self, Ks):
        """cat([-fcl, prp]) of get_ndc_fx_fy_cx_cy as one gather + one multiply-add of the flattened intrinsics (instead of ten
        elementwise launches; not a GEMM call either): (fx, fy, px, py) = (-2 K00 / s, -2 K11 / s, 1 - 2 K02 / s, 1 - 2 K12 / s)."""
        B, cols = Ks.shape[0], Ks.shape[2]
        key = (cols, Ks.device)
        if getattr(self, "_cam_key", None) != key:
            s = float(self.image_size)
            self._cam_idx = torch.tensor([0, cols + 1, 2, cols + 2], device=Ks.device)
            self._cam_scale = torch.full((4,), -2.0 / s, device=Ks.device)
            self._cam_b, self._cam_key = torch.tensor([0.0, 0.0, 1.0, 1.0], device=Ks.device), key
        return torch.addcmul(self._cam_b, Ks.reshape(B, -1).index_select(1, self._cam_idx), self._cam_scale)

    ENCODER_SIZE = 224          # the encoders' input resolution (the LightEstimator's flatten fixes it: 256 = 64 x 2 x 2 from 28 x 28 / 56 x 56)

    def encode(self, images):
        """base_encoder on the images; a render resolution other than 224 (BASELINE configs[4]) is resized first (nearest, as data_dic's
        HO-3D branch resizes the crop, utils/traineval_util.py:157)."""
        if images.shape[-1] != self.ENCODER_SIZE:
            images = torch.nn.functional.interpolate(images, (self.ENCODER_SIZE, self.ENCODER_SIZE))
        return self.base_encoder(images)

    def forward(self, dat_name, mode_train, images, Ks=None, root_xyz=None):
        low_features, features = self.encode(images)
        return self.forward_from_features(dat_name, mode_train, images, low_features, features, Ks=Ks, root_xyz=root_xyz)

    def forward_from_features(self, dat_name, mode_train, images, low_features, features, Ks=None, root_xyz=None):
        """Everything after the image encoder (models_res_nimble.py:118-225)."""
        if self.ifLight:
            light_params = self.light_estimator(low_features)
        hand_params = self.hand_encoder(features)
        outputs = self.hand_layer(hand_params, handle_collision=False)
        outputs.update(hand_params)
        # joints regressed from the posed verts + root-relative (models_res_nimble.py:150-166), one HIP launch
        root_id = 0 if (dat_name == "HO3D" and not mode_train) else self.root_id
        joints, mano_verts, pred_root = ops.mano_joints_root_relative(self.hand_layer.handle, outputs["mano_verts"], root_id)
        outputs["joints"], outputs["mano_verts"] = joints, mano_verts
        if self.ifRender:
            cam = self.camera_from_K(Ks)                                  # PerspectiveCameras(focal_length=-fcl, principal_point=prp)
            if self.ifLight:
                colors, directions = light_params["colors"], light_params["directions"]
            else:
                # PointLights() defaults (models_res_nimble.py:191-198; PyTorch3D [recalled]: ambient .5, diffuse .3, specular .2,
                # location (0, 1, 0)): constant, the renderer was created in point-light mode and reads `directions` as the location
                colors, directions = self._pl_color.expand(images.shape[0], -1), self._pl_location.expand(images.shape[0], -1)
            # skin_meshes.offset_verts_(-pred_root); .offset_verts_(+root_xyz)   (models_res_nimble.py:203-205)
            verts_cam = mano_verts + root_xyz
            vcolors = self.vertex_colors
            if self.ncomps[2]:                   # texture stand-in: per-sample vertex colours = skin tone + basis . texture_params
                # the texture-PCA decode kernel (csrc/texpca.hip): skin tone + texture_params . basis, [B, T] x [T, 778 * 3 (+ 2 pad)]
                tex = ops.texture_pca_decode(outputs["texture_params"], self.texture_basis_pad, self.texture_mean_pad)
                vcolors = tex[:, :778 * 3].reshape(-1, 778, 3)
            rgba, face_id = ops.render(self.renderer_p3d, verts_cam, vcolors, cam, colors, directions)
            outputs["re_img"] = rgba[:, :3]
            outputs["_rgba"] = rgba                                                      # for the fused photometric losses
            outputs["re_sil"], outputs["maskRGBs"] = ops.sil_post(rgba, images)          # :219-220, one launch
            outputs["face_id"] = face_id
            outputs["skin_verts"] = verts_cam
        outputs["mano_faces"] = self.mano_face.expand(images.shape[0], -1, -1)           # a view (the reference repeats)
        outputs["_faces_i32"] = self.hand_layer.mesh_face[0]
        return outputs
