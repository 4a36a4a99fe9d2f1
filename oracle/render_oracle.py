"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's render path.

Restates, in torch-CPU ops (autograd supplies reference gradients), what `Model.forward` does between
reference models_res_nimble.py:176-220 through PyTorch3D:

  get_ndc_fx_fy_cx_cy                      models_res_nimble.py:228-235
  PerspectiveCameras(focal_length=-fcl, principal_point=prp) -> MeshRasterizer.transform   [recalled]
  rasterize_meshes (hard, K=1)             oracle/raster_oracle.c                          [recalled]
  Meshes.verts_normals_packed              [recalled]
  HardPhongShader / phong_shading / DirectionalLights / Materials / hard_rgb_blend          [recalled]
  F.avg_pool2d(aa) and the re_img / re_sil / maskRGBs outputs   models_res_nimble.py:210-220

PyTorch3D is a third-party dependency of the reference that is absent from /root/reference (README.md:70-71,
unpinned git HEAD): every [recalled] item restates its published algorithm from memory and the reference
holds no tests or vectors at that boundary, so PARITY IS UNPINNED there.  The HIP renderer is checked
against THIS file (pixels to 1e-4, face indices bit-exact against raster_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
K_EPS = 1e-8


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libraster_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _LIB = ctypes.CDLL(path)
    return _LIB


@dataclass
class ShadeConsts:
    """Materials(diffuse .8, specular .2, shininess 30) (models_res_nimble.py:79-85) and the DirectionalLights
    defaults the reference leaves untouched (ambient .5, specular .2) [recalled]; Materials ambient default 1."""
    ambient: tuple = (0.5, 0.5, 0.5)          # materials.ambient_color(1) * lights.ambient_color(.5)
    mat_diffuse: tuple = (0.8, 0.8, 0.8)
    specular: tuple = (0.04, 0.04, 0.04)      # materials.specular_color(.2) * lights.specular_color(.2)
    shininess: float = 30.0
    background: tuple = (1.0, 1.0, 1.0)


def ndc_camera_from_K(Ks, image_size=224.0):
    """get_ndc_fx_fy_cx_cy + the sign flip of `focal_length=-fcl` (models_res_nimble.py:184-186,228-235).
    Ks [B,3,3] or [B,3,4] -> cam [B,4] = (fx, fy, px, py) as handed to PerspectiveCameras."""
    fx = Ks[:, 0, 0] * 2 / image_size
    fy = Ks[:, 1, 1] * 2 / image_size
    px = -(Ks[:, 0, 2] - image_size / 2) * 2 / image_size
    py = -(Ks[:, 1, 2] - image_size / 2) * 2 / image_size
    return torch.stack([-fx, -fy, px, py], dim=-1)


def project_ndc(verts, cam):
    """PerspectiveCameras projection (R=I, T=0) as MeshRasterizer.transform applies it [recalled]:
    x_ndc = (fx X + px Z) / Z, y_ndc = (fy Y + py Z) / Z, z = view-space Z."""
    X, Y, Z = verts[..., 0], verts[..., 1], verts[..., 2]
    fx, fy, px, py = (cam[:, k].unsqueeze(1) for k in range(4))
    return torch.stack([(X * fx + Z * px) / Z, (Y * fy + Z * py) / Z, Z], dim=-1)


def vertex_normals(verts, faces):
    """Meshes.verts_normals_packed [recalled]: area-weighted sum of cross(v2-v1, v0-v1) over incident faces,
    then F.normalize(eps=1e-6).  verts [B,V,3], faces LongTensor [F,3]."""
    v0, v1, v2 = verts[:, faces[:, 0]], verts[:, faces[:, 1]], verts[:, faces[:, 2]]
    fn = torch.cross(v2 - v1, v0 - v1, dim=-1)
    n = torch.zeros_like(verts)
    for k in range(3):
        n = n.index_add(1, faces[:, k], fn)
    return F.normalize(n, eps=1e-6, dim=-1)


def rasterize(verts_ndc, faces, S):
    """-> pix_to_face int32 [B,S,S] (local face index, -1 miss), zbuf, bary [B,S,S,3] (numpy, via the C oracle)."""
    vn = np.ascontiguousarray(verts_ndc.detach().cpu().numpy(), dtype=np.float32)
    fc = np.ascontiguousarray(faces.cpu().numpy() if torch.is_tensor(faces) else faces, dtype=np.int32)
    B, V, _ = vn.shape
    p2f = np.empty((B, S, S), np.int32)
    zbuf = np.empty((B, S, S), np.float32)
    bary = np.empty((B, S, S, 3), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _lib().raster_oracle(vn.ctypes.data_as(fp), fc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                         ctypes.c_int(B), ctypes.c_int(V), ctypes.c_int(fc.shape[0]), ctypes.c_int(S),
                         p2f.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), zbuf.ctypes.data_as(fp), bary.ctypes.data_as(fp))
    return p2f, zbuf, bary


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def pix_centres(S, dtype=torch.float32):
    i = torch.arange(S, dtype=dtype)
    c = -1.0 + (2.0 * (S - 1 - i) + 1.0) / S       # index 0 <-> NDC +1 (x: left, y: top)
    return c


def differentiable_bary(verts_ndc, faces, p2f):
    """Perspective-corrected barycentrics of the winning face at every sample, as a differentiable function of
    verts_ndc (PyTorch3D BarycentricCoordsForward + BarycentricPerspectiveCorrectionForward [recalled])."""
    B, S = p2f.shape[0], p2f.shape[1]
    hit = p2f >= 0
    idx = p2f.clamp(min=0).long()
    fv = verts_ndc[:, faces]                                           # [B,F,3,3]
    g = torch.gather(fv.reshape(B, -1, 9), 1, idx.reshape(B, -1, 1).expand(-1, -1, 9)).reshape(B, S, S, 3, 3)
    c = pix_centres(S, verts_ndc.dtype)
    py = c.view(1, S, 1).expand(B, S, S)
    px = c.view(1, 1, S).expand(B, S, S)
    x0, y0, z0 = g[..., 0, 0], g[..., 0, 1], g[..., 0, 2]
    x1, y1, z1 = g[..., 1, 0], g[..., 1, 1], g[..., 1, 2]
    x2, y2, z2 = g[..., 2, 0], g[..., 2, 1], g[..., 2, 2]
    area = _edge(x2, y2, x0, y0, x1, y1) + K_EPS
    w0 = _edge(px, py, x1, y1, x2, y2) / area
    w1 = _edge(px, py, x2, y2, x0, y0) / area
    w2 = _edge(px, py, x0, y0, x1, y1) / area
    t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
    denom = (t0 + t1 + t2).clamp(min=K_EPS)
    bary = torch.stack([t0 / denom, t1 / denom, t2 / denom], dim=-1)
    return torch.where(hit.unsqueeze(-1), bary, torch.zeros_like(bary)), hit, idx


def _interp(attr, faces, idx, bary):
    """interpolate_face_attributes: attr [B,V,D] -> [B,S,S,D]."""
    B, S = idx.shape[0], idx.shape[1]
    fa = attr[:, faces]                                                # [B,F,3,D]
    D = attr.shape[-1]
    g = torch.gather(fa.reshape(B, -1, 3 * D), 1, idx.reshape(B, -1, 1).expand(-1, -1, 3 * D)).reshape(B, S, S, 3, D)
    return (bary.unsqueeze(-1) * g).sum(-2)


def sample_textures_uv(maps, faces_uvs, verts_uvs, idx, bary):
    """TexturesUV.sample_textures [recalled, PyTorch3D renderer/mesh/textures.py]: pixel uv = barycentric interpolation of the face's three
    uv coordinates; the maps are flipped vertically and sampled with F.grid_sample(2 uv - 1, bilinear, align_corners=True, padding border).
    maps [B,TH,TW,3], faces_uvs Long [F,3], verts_uvs [n,2], idx Long [B,S,S] (clamped face index), bary [B,S,S,3] -> [B,S,S,3]."""
    fu = verts_uvs[faces_uvs]                                   # [F,3,2]
    uv = (bary.unsqueeze(-1) * fu[idx]).sum(-2)                  # [B,S,S,2]
    tex = torch.flip(maps.permute(0, 3, 1, 2), [2])
    return F.grid_sample(tex, uv * 2.0 - 1.0, mode="bilinear", align_corners=True, padding_mode="border").permute(0, 2, 3, 1)


def render(verts, vcolors, cam, light_color, light_dir, faces, image_size=224, aa=3, consts=ShadeConsts(), point_lights=False, textures_uv=None):
    """verts [B,V,3] (view space), vcolors [B,V,3] (TexturesVertex stand-in), cam [B,4], light_color/dir [B,3],
    faces LongTensor [F,3].  -> rgba [B,4,H,H] after the aa x aa average pool, pix_to_face [B,S,S] (numpy)."""
    faces = torch.as_tensor(faces).long()
    S = image_size * aa
    vndc = project_ndc(verts, cam)
    p2f_np, _, _ = rasterize(vndc, faces, S)
    p2f = torch.from_numpy(p2f_np)
    bary, hit, idx = differentiable_bary(vndc, faces, p2f)
    normals = vertex_normals(verts, faces)
    P = _interp(verts, faces, idx, bary)
    N = _interp(normals, faces, idx, bary)
    if textures_uv is not None:                                  # (maps [B,TH,TW,3], faces_uvs [F,3], verts_uvs [n,2]); vcolors unused
        maps, faces_uvs, verts_uvs = textures_uv
        T = sample_textures_uv(maps, torch.as_tensor(faces_uvs).long(), verts_uvs, idx, bary)
    else:
        T = _interp(vcolors, faces, idx, bary)
    dt = verts.dtype
    amb = torch.tensor(consts.ambient, dtype=dt)
    md = torch.tensor(consts.mat_diffuse, dtype=dt)
    sp = torch.tensor(consts.specular, dtype=dt)
    lc = light_color.view(-1, 1, 1, 3)
    if point_lights:       # PointLights [recalled]: direction = location - point, normalised per sample (lighting.py PointLights.diffuse)
        ld = F.normalize(light_dir.view(-1, 1, 1, 3) - P, p=2, dim=-1, eps=1e-6)
    else:
        ld = F.normalize(light_dir, p=2, dim=-1, eps=1e-6).view(-1, 1, 1, 3)
    nh = F.normalize(N, p=2, dim=-1, eps=1e-6)
    cosang = (nh * ld).sum(-1)
    diffuse = lc * F.relu(cosang)[..., None]
    mask = (cosang > 0).to(dt)
    vh = F.normalize(-P, p=2, dim=-1, eps=1e-6)                         # camera centre = origin
    refl = -ld + 2 * (cosang[..., None] * nh)
    alpha = F.relu((vh * refl).sum(-1)) * mask
    spec = sp * torch.pow(alpha, consts.shininess)[..., None]
    colors = (amb + md * diffuse) * T + spec
    bg = torch.tensor(consts.background, dtype=dt).view(1, 1, 1, 3)
    rgb = torch.where(hit.unsqueeze(-1), colors, bg.expand_as(colors))   # hard_rgb_blend
    rgba = torch.cat([rgb, hit.to(dt).unsqueeze(-1)], dim=-1).permute(0, 3, 1, 2)
    rgba = F.avg_pool2d(rgba, kernel_size=aa, stride=aa)                # models_res_nimble.py:210-211
    return rgba, p2f_np


def model_render_outputs(rgba, images):
    """models_res_nimble.py:217-220."""
    re_img = rgba[:, :3]
    re_sil = rgba[:, 3:4].detach().clone()
    re_sil[re_sil > 0] = 255
    mask_rgbs = images * (re_sil > 0).float().repeat(1, 3, 1, 1)
    return re_img, re_sil, mask_rgbs
