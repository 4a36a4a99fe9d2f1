"""`Model` with the reference's constructor signature, forward signature and output-dict keys
(reference models_res_nimble.py:32-225), assembled from this package's HIP ops.

hand_model == 'mano' with render=True is a composition the reference cannot run (MyMANOLayer
builds a Meshes without textures); this build defines it with a constant per-vertex skin colour as the
TexturesVertex stand-in.  hand_model == 'nimble': the reference's MyNIMBLELayer is an un-vendored submodule whose source and assets
are absent, so `MyNIMBLELayer` here is a layer of NIMBLE's SHAPE (20 / 30 / 10 components, 25 joints, 5 990
skin vertices, a 778-vertex MANO-topology regression, per-vertex colours from the texture PCA) on caller-supplied `NimbleTables`
(hifihr_amd/nimble_tables.py; seeded synthetic ones by default) -- "parity unpinned" for NIMBLE's own numbers, checked against
oracle/lbs_oracle.py.  The NIMBLE configurations (full_rhd_freihand.json, weak_rhd_ho3d.json) also run as "MANO + a texture stand-in"
(SURVEY.md section 8(c)): `texture_stand_in=T` adds the T-component texture head of the NIMBLE HandEncoder and turns it into per-vertex
colours through a fixed seeded linear basis (skin tone + basis . texture_params), so `texture_params`, the `mtex` term and the texture
gradient exist.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .mano_tables import ManoTables, synthetic_mano_tables
from .nimble_tables import NimbleTables, synthetic_nimble_tables
from .network import HandEncoder, LightEstimator, ResEncoder

SKIN_TONE = (0.78, 0.60, 0.50)


def texture_stand_in_basis(ncomp: int) -> torch.Tensor:
    """[ncomp, 778*3] fixed linear map texture_params -> per-vertex colour offsets of the NIMBLE texture stand-in."""
    gen = torch.Generator().manual_seed(7)
    return 0.05 * torch.randn(int(ncomp), 778 * 3, generator=gen)


_MANO_FUSED = os.environ.get("HIFIHR_MANO_FUSED", "1") != "0"
_LIGHT_BRANCH = os.environ.get("HIFIHR_LIGHT_BRANCH", "1") != "0"


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


# Mano2Frei as a gather: FreiHAND joint i = MANO-ordered joint _FREI_FROM_MANO[i] (reference utils/fh_utils.py:542-556)
_MANO_TO_FREI = {0: 0, 1: 5, 2: 6, 3: 7, 4: 8, 5: 9, 6: 10, 7: 11, 8: 12, 9: 17, 10: 18, 11: 19, 12: 20, 13: 13, 14: 14, 15: 15, 16: 16,
                 17: 1, 18: 2, 19: 3, 20: 4}
_FREI_FROM_MANO = [k for k, _ in sorted(_MANO_TO_FREI.items(), key=lambda kv: kv[1])]


class MyNIMBLELayer(nn.Module):
    """A hand layer of NIMBLE's shape (what reference models_res_nimble.py:133-142 consumes): pose PCA decode (csrc/texpca.hip) ->
    generic LBS over the 25-joint tree (csrc/lbs.hip) -> the 778-vertex MANO-topology regression (barycentric on skin faces), the 21
    MANO-ordered joints, per-vertex colours from the texture PCA (csrc/texpca.hip)."""

    def __init__(self, ifRender, device, shape_ncomp=20, pose_ncomp=30, tex_ncomp=10, tables: NimbleTables | None = None):
        super().__init__()
        t = self.tables = tables if tables is not None else synthetic_nimble_tables(0)
        assert t.shapedirs.shape[2] == shape_ncomp and t.pose_basis.shape[0] == pose_ncomp and t.tex_basis.shape[0] == tex_ncomp
        self.handle = ops.LbsHandle(t.v_template, t.shapedirs, t.J_regressor, t.weights, t.parents)
        self.V, self.J, self.ifRender = t.v_template.shape[0], t.weights.shape[1], ifRender
        pad4 = lambda a: torch.nn.functional.pad(torch.as_tensor(a, dtype=torch.float32), (0, (-a.shape[-1]) % 4)).contiguous()
        self.register_buffer("pose_basis", pad4(t.pose_basis), persistent=False)        # the decode kernel moves float4
        self.register_buffer("pose_mean", pad4(t.pose_mean), persistent=False)
        self.register_buffer("tex_basis", pad4(t.tex_basis), persistent=False)
        self.register_buffer("tex_mean", pad4(t.tex_mean), persistent=False)
        self.register_buffer("mesh_face", torch.as_tensor(t.faces, dtype=torch.int32).unsqueeze(0), persistent=False)
        corner = torch.as_tensor(t.faces, dtype=torch.long)[torch.as_tensor(t.mano_vreg_fidx, dtype=torch.long)]          # [778,3]
        self.register_buffer("vreg_corner", corner.reshape(-1), persistent=False)
        self.register_buffer("vreg_bc", torch.as_tensor(t.mano_vreg_bc, dtype=torch.float32).view(1, 778, 3, 1), persistent=False)
        self.register_buffer("joint21", torch.as_tensor(t.joint21, dtype=torch.long), persistent=False)
        self.uv_texture = t.faces_uvs is not None and t.tex_img_basis is not None
        if self.uv_texture:                           # the texture is an IMAGE sampled through per-face UVs (TexturesUV), not vertex colours
            self.register_buffer("tex_img_basis", pad4(t.tex_img_basis), persistent=False)
            self.register_buffer("tex_img_mean", pad4(t.tex_img_mean), persistent=False)

    def forward(self, hand_params, handle_collision=True):
        B = hand_params["pose_params"].shape[0]
        theta = ops.texture_pca_decode(hand_params["pose_params"], self.pose_basis, self.pose_mean)[:, :self.J * 3]
        verts, joints = ops.lbs(self.handle, theta.reshape(B, self.J, 3), hand_params["shape_params"])
        mano_verts = (verts.index_select(1, self.vreg_corner).view(B, 778, 3, 3) * self.vreg_bc).sum(2)
        out = {"nimble_joints": joints, "verts": verts, "faces": None, "mano_verts": mano_verts,
               "joints": joints.index_select(1, self.joint21), "rot": None}
        if self.ifRender and hand_params.get("texture_params") is not None and self.uv_texture:
            TH, TW = self.tables.tex_hw
            out["texture_maps"] = ops.texture_pca_decode(hand_params["texture_params"], self.tex_img_basis, self.tex_img_mean)[:, :TH * TW * 3].reshape(B, TH, TW, 3)
        elif self.ifRender and hand_params.get("texture_params") is not None:
            out["textures"] = ops.texture_pca_decode(hand_params["texture_params"], self.tex_basis, self.tex_mean)[:, :self.V * 3].reshape(B, self.V, 3)
        return out


class Model(nn.Module):
    def __init__(self, ifRender, device, if_4c, hand_model, use_mean_shape, pretrain, root_id=9, root_id_nimble=11,
                 ifLight=True, mano_tables: ManoTables | None = None, image_size=224, aa_factor=3, texture_stand_in=0,
                 nimble_tables: NimbleTables | None = None, conv_precision="fast"):
        """conv_precision: "fast" (default) -- the encoder's stride-1 3x3 convolutions run as Winograd F(4x4, 3x3) / F(2x2, 3x3); "reference"
        -- on the direct kernels, whose outputs round like a plain fp32 convolution: features within 5e-6 of the reference's instead of
        1.3e-5, for ~50 % more time per step.  The trunk's GRADIENT error against the reference (~1e-2 of a gradient's maximum on the
        batch-of-8 fixture: ReLU sign flips) does not shrink with it (README "Precision of the default dispatch").  Per model, not per process."""
        super().__init__()
        ops.conv_precision(conv_precision)               # (validates the name)
        self.conv_precision = conv_precision
        if hand_model not in ("mano", "nimble"):
            raise NotImplementedError(f"hand_model='{hand_model}': 'mano' and 'nimble' are built")
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
        if hand_model == "nimble":                                   # models_res_nimble.py:55-57
            assert not texture_stand_in, "texture_stand_in is the MANO-topology stand-in; hand_model='nimble' has its own texture PCA"
            self.ncomps = [20, 30, 10]
            self.hand_layer = MyNIMBLELayer(ifRender, device, shape_ncomp=20, pose_ncomp=30, tex_ncomp=10, tables=nimble_tables)
            mano_faces = (mano_tables if mano_tables is not None else synthetic_mano_tables(0)).faces       # :62-64 (MANO_RIGHT.pkl 'f')
            self.register_buffer("_mano_face_i32", torch.as_tensor(mano_faces, dtype=torch.int32), persistent=False)
            self.register_buffer("_frei_from_mano", torch.tensor(_FREI_FROM_MANO), persistent=False)
        else:
            self.ncomps = [10, 48, int(texture_stand_in) if texture_stand_in else None]
            self.hand_layer = MyMANOLayer(ifRender, device, shape_ncomp=10, pose_ncomp=48, tables=mano_tables)
            self.register_buffer("_mano_face_i32", self.hand_layer.mesh_face[0].clone(), persistent=False)
        self.hand_encoder = HandEncoder(hand_model=hand_model, ncomps=self.ncomps, in_dim=self.features_dim,
                                        ifRender=ifRender, use_mean_shape=use_mean_shape)
        self.register_buffer("mano_face", self._mano_face_i32.unsqueeze(0).to(torch.int16), persistent=False)
        self.ifRender, self.ifLight, self.aa_factor, self.image_size = ifRender, ifLight, aa_factor, image_size
        if ifRender:
            # Materials(diffuse .8, specular .2, shininess 30) + DirectionalLights defaults (ambient .5, specular .2)
            if not ifLight:
                self.register_buffer("_pl_color", torch.tensor([[0.3, 0.3, 0.3]]), persistent=False)
                self.register_buffer("_pl_location", torch.tensor([[0.0, 1.0, 0.0]]), persistent=False)
            self.renderer_p3d = ops.RendererHandle(self.hand_layer.tables.faces, int(self.hand_layer.tables.v_template.shape[0]), image_size=image_size, aa=aa_factor, point_lights=not ifLight,
                                                   ambient=(0.5,) * 3, mat_diffuse=(0.8,) * 3, specular=(0.04,) * 3,
                                                   shininess=30.0, background=(1.0,) * 3)
            if hand_model == "nimble" and self.hand_layer.uv_texture:
                self.renderer_p3d.set_uv(self.hand_layer.tables.faces_uvs, self.hand_layer.tables.verts_uvs)
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
        with ops.conv_precision(self.conv_precision):
            return self.base_encoder(images)

    accepts_cam_ndc = True      # forward(..., cam_ndc=): the NDC camera terms precomputed by the batch kernel (superset of the reference's signature)

    def forward(self, dat_name, mode_train, images, Ks=None, root_xyz=None, cam_ndc=None):
        low_features, features = self.encode(images)
        return self.forward_from_features(dat_name, mode_train, images, low_features, features, Ks=Ks, root_xyz=root_xyz, cam_ndc=cam_ndc)

    def forward_from_features(self, dat_name, mode_train, images, low_features, features, Ks=None, root_xyz=None, cam_ndc=None):
        """Everything after the image encoder (models_res_nimble.py:118-225).  cam_ndc [B,4]: cat([-fcl, prp]) of get_ndc_fx_fy_cx_cy
        (Ks) when the caller already holds it (data.FreiHandDeviceCache.batch_examples emits it with the batch)."""
        br = None
        if self.ifLight:
            # The light estimator (3 small convolutions, 2 pools, 2 linears: ~10 launches forward, ~18 backward, each a few microseconds
            # of work on a handful of CUs) and the chain hand encoder -> MANO -> joints are independent until the renderer: on a side
            # stream the two latency chains overlap (ops.side_branch).  HIFIHR_LIGHT_BRANCH=0: one stream.
            # MEASURED: ResNet-18 step (B = 32) 5.38 -> 5.32 ms; EfficientNet-b3 config 3 (B = 48) 34.11 -> 34.49 ms -- there the branch
            # runs beside the perceptual loss's large kernels and only delays them; ResNet-50 (512-channel low features: the branch's first
            # convolution is no longer small) 15.73 -> 15.82 ms.  The ResNet-18 trunk (128-channel low features) only.
            with ops.side_branch(low_features, "light", enabled=_LIGHT_BRANCH and self.ifRender and self.low_feat_dim == 128) as br:
                light_params = self.light_estimator(low_features)
        hand_params = self.hand_encoder(features)
        root_id = 0 if (dat_name == "HO3D" and not mode_train) else self.root_id
        verts_cam = None
        if self.hand_model == "mano" and _MANO_FUSED and features.is_cuda:
            # ManoLayer.forward, the joint regression, the root-relative step and the camera-space offset of the mesh: ONE launch
            # (ops.mano_full; HIFIHR_MANO_FUSED=0: the layer, then ops.mano_joints_root_relative, then an elementwise add)
            joints, mano_verts, verts_cam, pred_root, pose_o, shape_o = ops.mano_full(
                self.hand_layer.handle, hand_params["pose_params"], hand_params["shape_params"], root_id, root_xyz if self.ifRender else None)
            # (the layer's dict holds the ABSOLUTE posed vertices under 'skin_verts'; with the renderer on it is overwritten below)
            outputs = {"skin_verts": mano_verts if self.ifRender else mano_verts + pred_root.reshape(-1, 1, 3)}
            outputs.update(hand_params)
            outputs["pose_params"], outputs["shape_params"] = pose_o, shape_o      # (aliases: the regularisers' gradient joins the layer's backward)
        else:
            outputs = self.hand_layer(hand_params, handle_collision=False)
            outputs.update(hand_params)
        if self.hand_model == "nimble":
            if br is not None:
                br.join(*light_params.values())
            return self._nimble_tail(dat_name, mode_train, images, outputs, light_params if self.ifLight else None, Ks, root_xyz)
        if verts_cam is None:
            # joints regressed from the posed verts + root-relative (models_res_nimble.py:150-166), one HIP launch
            joints, mano_verts, pred_root = ops.mano_joints_root_relative(self.hand_layer.handle, outputs["mano_verts"], root_id)
            verts_cam = (mano_verts + root_xyz) if self.ifRender else None   # skin_meshes.offset_verts_(-pred_root); .offset_verts_(+root_xyz)  (:203-205)
        outputs["joints"], outputs["mano_verts"] = joints, mano_verts
        if self.ifRender:
            cam = cam_ndc if cam_ndc is not None else self.camera_from_K(Ks)   # PerspectiveCameras(focal_length=-fcl, principal_point=prp)
            if br is not None:
                br.join(*light_params.values())                           # the light branch meets the main chain at the renderer
            if self.ifLight:
                colors, directions = light_params["colors"], light_params["directions"]
            else:
                # PointLights() defaults (models_res_nimble.py:191-198; PyTorch3D [recalled]: ambient .5, diffuse .3, specular .2,
                # location (0, 1, 0)): constant, the renderer was created in point-light mode and reads `directions` as the location
                colors, directions = self._pl_color.expand(images.shape[0], -1), self._pl_location.expand(images.shape[0], -1)
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
        outputs["_faces_i32"] = self._mano_face_i32
        return outputs

    def _nimble_tail(self, dat_name, mode_train, images, outputs, light_params, Ks, root_xyz):
        """models_res_nimble.py:156-225 for hand_model == 'nimble': Mano2Frei on the layer's 21 joints, root-relative joints / MANO-topology
        verts / the 25 bone joints, the SKIN mesh rendered with its texture."""
        B = images.shape[0]
        joints = outputs["joints"].index_select(1, self._frei_from_mano)                 # Mano2Frei (:157)
        eval_ho3d = dat_name == "HO3D" and not mode_train
        pred_root = joints[:, 0 if eval_ho3d else self.root_id].unsqueeze(1)             # :161-166
        outputs["joints"], outputs["mano_verts"] = joints - pred_root, outputs["mano_verts"] - pred_root
        nroot = outputs["nimble_joints"][:, 0 if eval_ho3d else self.root_id_nimble].unsqueeze(1)      # :167-172
        outputs["nimble_joints"] = outputs["nimble_joints"] - nroot
        if self.ifRender:
            cam = self.camera_from_K(Ks)
            if self.ifLight:
                colors, directions = light_params["colors"], light_params["directions"]
            else:
                colors, directions = self._pl_color.expand(B, -1), self._pl_location.expand(B, -1)
            verts_cam = outputs["verts"] - pred_root + root_xyz                          # :203-205
            if "texture_maps" in outputs:             # TexturesUV (models_res_nimble.py:203-208): the texture image sampled per sample
                rgba, face_id = ops.render_uv(self.renderer_p3d, verts_cam, outputs["texture_maps"], cam, colors, directions)
            else:
                rgba, face_id = ops.render(self.renderer_p3d, verts_cam, outputs["textures"], cam, colors, directions)
            outputs["re_img"], outputs["_rgba"] = rgba[:, :3], rgba
            outputs["re_sil"], outputs["maskRGBs"] = ops.sil_post(rgba, images)
            outputs["face_id"], outputs["skin_verts"] = face_id, verts_cam
        outputs["mano_faces"] = self.mano_face.expand(B, -1, -1)
        outputs["_faces_i32"] = self._mano_face_i32
        return outputs
