"""Kernel-level parity cases shared by the hostsim (CPU, `not gpu`) and HIP (`gpu`) test modules.

Each case drives the C ABI through hifihr_amd._lib.HifihrLib with tensors on `device` and compares with the
oracle.  `lib` is either the product library (device='cuda') or tests/hostsim's emulator build (device='cpu').
"""
import os
import subprocess

import numpy as np
import torch

from hifihr_amd._lib import HifihrLib
from oracle import mano_oracle as mo

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM_DIR = os.path.join(REPO, "tests", "hostsim")
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, "libhifihr_hostsim.so")


def bn_slots(stats, C, slots=32):
    """[slots][2][C] partial (sum, sum of squares) view of a batch-norm FORWARD statistics buffer: float64 since round 3
    (include/hifihr.h: hifihr_bn_stats_floats; csrc/hifihr_internal.h "FORWARD statistics")."""
    return stats[:slots * 2 * C * 2].view(torch.float64).view(slots, 2, C)


def build_hostsim() -> HifihrLib:
    subprocess.run(["make", "-s", "-C", HOSTSIM_DIR, "-j8"], check=True)
    return HifihrLib(HOSTSIM_LIB)


def _dev(x, device):
    return torch.as_tensor(x).to(device).contiguous()


def mano_fwd_bwd_case(lib, tables, g, device, atol_v=5e-6, gtol=3e-4):
    """g: golden dict with pose, beta, wv, wj, verts, jtr, gpose, gbeta (reference ManoLayer outputs)."""
    h = lib.mano_create(tables)
    try:
        pose, beta = _dev(g["pose"], device), _dev(g["beta"], device)
        B = pose.shape[0]
        verts = torch.empty(B, 778, 3, device=device)
        jtr = torch.empty(B, 21, 3, device=device)
        saved = torch.empty(B, 778, 3, device=device)
        lib.mano_lbs_fwd(h, pose, beta, verts, jtr, saved)
        np.testing.assert_allclose(verts.cpu().numpy(), g["verts"], atol=atol_v, rtol=0)
        np.testing.assert_allclose(jtr.cpu().numpy(), g["jtr"], atol=atol_v, rtol=0)
        gpose = torch.empty(B, 48, device=device)
        gbeta = torch.empty(B, 10, device=device)
        lib.mano_lbs_bwd(h, pose, beta, saved, _dev(g["wv"], device), _dev(g["wj"], device), gpose, gbeta)
        scale_p = np.abs(g["gpose"]).max()
        scale_b = np.abs(g["gbeta"]).max()
        np.testing.assert_allclose(gpose.cpu().numpy(), g["gpose"], atol=gtol * scale_p, rtol=1e-4)
        np.testing.assert_allclose(gbeta.cpu().numpy(), g["gbeta"], atol=gtol * scale_b, rtol=1e-4)
    finally:
        lib.mano_destroy(h)


def mano_random_vs_oracle_case(lib, tables, device, B, seed):
    gen = torch.Generator().manual_seed(seed)
    pose = (0.6 * torch.randn(B, 48, generator=gen)).requires_grad_(True)
    beta = (0.7 * torch.randn(B, 10, generator=gen)).requires_grad_(True)
    wv = torch.randn(B, 778, 3, generator=gen)
    wj = torch.randn(B, 21, 3, generator=gen)
    verts, jtr, _ = mo.mano_forward(tables, pose, beta)
    ((verts * wv).sum() + (jtr * wj).sum()).backward()
    g = dict(pose=pose.detach().numpy(), beta=beta.detach().numpy(), wv=wv.numpy(), wj=wj.numpy(),
             verts=verts.detach().numpy(), jtr=jtr.detach().numpy(), gpose=pose.grad.numpy(), gbeta=beta.grad.numpy())
    mano_fwd_bwd_case(lib, tables, g, device)


def mano_joints_case(lib, tables, device, B, seed, root_id=9):
    gen = torch.Generator().manual_seed(seed)
    verts = (0.1 * torch.randn(B, 778, 3, generator=gen)).requires_grad_(True)
    j = mo.xyz_from_vertice(tables, verts)
    if root_id >= 0:
        jr, vr, root = mo.root_relative(j, verts, root_id)
    else:
        jr, vr, root = j, verts, torch.zeros(B, 1, 3)
    wj = torch.randn(B, 21, 3, generator=gen)
    wv = torch.randn(B, 778, 3, generator=gen)
    wr = torch.randn(B, 3, generator=gen)
    ((jr * wj).sum() + (vr * wv).sum() + (root.reshape(B, 3) * wr).sum()).backward()
    h = lib.mano_create(tables)
    try:
        vd = _dev(verts.detach(), device)
        o_j = torch.empty(B, 21, 3, device=device)
        o_v = torch.empty(B, 778, 3, device=device)
        o_r = torch.empty(B, 3, device=device)
        lib.mano_joints_fwd(h, vd, root_id, o_j, o_v, o_r)
        np.testing.assert_allclose(o_j.cpu().numpy(), jr.detach().numpy(), atol=2e-6)
        np.testing.assert_allclose(o_v.cpu().numpy(), vr.detach().numpy(), atol=2e-6)
        np.testing.assert_allclose(o_r.cpu().numpy(), root.detach().reshape(B, 3).numpy(), atol=2e-6)
        gv = torch.empty(B, 778, 3, device=device)
        lib.mano_joints_bwd(h, _dev(wj, device), _dev(wv, device), _dev(wr, device) if root_id >= 0 else None, root_id, gv)
        np.testing.assert_allclose(gv.cpu().numpy(), verts.grad.numpy(), atol=2e-4, rtol=1e-4)
    finally:
        lib.mano_destroy(h)


def mano_full_case(lib, tables, device, B, seed, root_id=9, with_cam=True):
    """hifihr_mano_full_fwd / _bwd (layer + joint regression + root-relative step + camera-space offset; the backward in one launch)
    against the ORACLE chain (oracle/mano_oracle.py: mano_forward -> xyz_from_vertice -> root_relative -> + root_xyz) and, bit for bit in
    the forward direction, against the two-call form of the same library."""
    gen = torch.Generator().manual_seed(seed)
    pose = (0.6 * torch.randn(B, 48, generator=gen)).requires_grad_(True)
    beta = (0.7 * torch.randn(B, 10, generator=gen)).requires_grad_(True)
    root_xyz = torch.randn(B, 3, generator=gen) * 0.3
    verts, _, _ = mo.mano_forward(tables, pose, beta)
    j = mo.xyz_from_vertice(tables, verts)
    if root_id >= 0:
        jr, vr, root = mo.root_relative(j, verts, root_id)
    else:
        jr, vr, root = j, verts, torch.zeros(B, 1, 3)
    vc = vr + root_xyz.unsqueeze(1)
    wj, wv, wc, wr = (torch.randn(B, 21, 3, generator=gen), torch.randn(B, 778, 3, generator=gen), torch.randn(B, 778, 3, generator=gen),
                      torch.randn(B, 3, generator=gen))
    tot = (jr * wj).sum() + (vr * wv).sum() + (root.reshape(B, 3) * wr).sum()
    if with_cam:
        tot = tot + (vc * wc).sum()
    tot.backward()
    h = lib.mano_create(tables)
    try:
        f = lambda *shape: torch.empty(*shape, device=device)
        pd, bd = _dev(pose.detach(), device), _dev(beta.detach(), device)
        o_verts, o_j, o_v, o_c, o_r, saved = f(B, 778, 3), f(B, 21, 3), f(B, 778, 3), f(B, 778, 3), f(B, 3), f(B, 778, 3)
        lib.mano_full_fwd(h, pd, bd, root_id, _dev(root_xyz, device), o_verts, o_j, o_v, o_c, o_r, saved)
        np.testing.assert_allclose(o_j.cpu().numpy(), jr.detach().numpy(), atol=5e-6)
        np.testing.assert_allclose(o_v.cpu().numpy(), vr.detach().numpy(), atol=5e-6)
        np.testing.assert_allclose(o_c.cpu().numpy(), vc.detach().numpy(), atol=5e-6)
        np.testing.assert_allclose(o_r.cpu().numpy(), root.detach().reshape(B, 3).numpy(), atol=5e-6)
        # the two-call form of the same library: identical bits
        t_verts, t_jtr, t_saved, t_j, t_v, t_r = f(B, 778, 3), f(B, 21, 3), f(B, 778, 3), f(B, 21, 3), f(B, 778, 3), f(B, 3)
        lib.mano_lbs_fwd(h, pd, bd, t_verts, t_jtr, t_saved)
        lib.mano_joints_fwd(h, t_verts, root_id, t_j, t_v, t_r)
        assert torch.equal(t_verts, o_verts) and torch.equal(t_saved, saved) and torch.equal(t_j, o_j) and torch.equal(t_v, o_v) and torch.equal(t_r, o_r)
        gp, gb = f(B, 48), f(B, 10)
        lib.mano_full_bwd(h, pd, bd, saved, _dev(wj, device), _dev(wv, device), _dev(wc, device) if with_cam else None,
                          _dev(wr, device) if root_id >= 0 else None, root_id, gp, gb)
        eg = float(np.abs(gp.cpu().numpy() - pose.grad.numpy()).max() / np.abs(pose.grad.numpy()).max())
        eb = float(np.abs(gb.cpu().numpy() - beta.grad.numpy()).max() / np.abs(beta.grad.numpy()).max())
        assert eg <= 3e-4 and eb <= 3e-4, (eg, eb)
        gp2, gb2 = f(B, 48), f(B, 10)                         # deterministic: no float atomics
        lib.mano_full_bwd(h, pd, bd, saved, _dev(wj, device), _dev(wv, device), _dev(wc, device) if with_cam else None,
                          _dev(wr, device) if root_id >= 0 else None, root_id, gp2, gb2)
        assert torch.equal(gp, gp2) and torch.equal(gb, gb2)
        # gradients that reach pose / beta through their other consumer are added inside the launch
        ap, ab = torch.randn(B, 48, generator=gen), torch.randn(B, 10, generator=gen)
        lib.mano_full_bwd(h, pd, bd, saved, _dev(wj, device), _dev(wv, device), _dev(wc, device) if with_cam else None,
                          _dev(wr, device) if root_id >= 0 else None, root_id, gp2, gb2, gpose_add=_dev(ap, device), gbeta_add=_dev(ab, device))
        assert torch.equal(gp2.cpu(), gp.cpu() + ap) and torch.equal(gb2.cpu(), gb.cpu() + ab)
    finally:
        lib.mano_destroy(h)


# ------------------------------------------------------------------------------------------------
# renderer
# ------------------------------------------------------------------------------------------------
def make_render_inputs(tables, B, seed, image_size, z=0.6):
    """Posed synthetic hands in front of a FreiHAND-like camera (scaled to `image_size`)."""
    from oracle import render_oracle as ro
    gen = torch.Generator().manual_seed(seed)
    pose = 0.4 * torch.randn(B, 48, generator=gen)
    pose[:, :3] = torch.randn(B, 3, generator=gen)             # arbitrary global orientation
    beta = 0.5 * torch.randn(B, 10, generator=gen)
    verts, _, _ = mo.mano_forward(tables, pose, beta)
    root = torch.stack([0.05 * (torch.rand(B, generator=gen) - 0.5), 0.05 * (torch.rand(B, generator=gen) - 0.5),
                        z + 0.2 * torch.rand(B, generator=gen)], dim=1)
    verts = (verts + root.unsqueeze(1)).detach()
    f = 450.0 + 200.0 * torch.rand(B, generator=gen)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = f; K[:, 1, 1] = f; K[:, 2, 2] = 1
    K[:, 0, 2] = 112 + 20 * (torch.rand(B, generator=gen) - 0.5)
    K[:, 1, 2] = 112 + 20 * (torch.rand(B, generator=gen) - 0.5)
    cam = ro.ndc_camera_from_K(K)                             # NDC camera: independent of the raster size
    vcol = 0.3 + 0.6 * torch.rand(B, 778, 3, generator=gen)
    lc = torch.rand(B, 3, generator=gen) * 1.6 - 0.6           # hardtanh range [-1,1]
    ld = torch.randn(B, 3, generator=gen)
    return verts, vcol, cam, lc, ld


def texture_pca_case(lib, device, B, K, n, seed=0, with_mean=True):
    """csrc/texpca.hip vs torch: tex = mean + coef . basis and d(sum(tex * w))/dcoef."""
    gen = torch.Generator().manual_seed(seed)
    coef = torch.randn(B, K, generator=gen); basis = torch.randn(K, n, generator=gen) * 0.1
    mean = torch.rand(n, generator=gen) if with_mean else None
    w = torch.randn(B, n, generator=gen)
    cr = coef.clone().requires_grad_(True)
    ref = cr @ basis + (mean if with_mean else 0.0)
    (ref * w).sum().backward()
    d = lambda t: t.to(device).contiguous() if t is not None else None
    out = torch.full((B, n), 7.0, device=device)
    lib.texture_pca_fwd(d(coef), d(basis), d(mean), out)
    assert float((out.cpu() - ref.detach()).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    dc = torch.zeros(B, K, device=device)
    lib.texture_pca_bwd(d(w), d(basis), dc)
    assert float((dc.cpu() - cr.grad).abs().max()) <= 2e-5 * float(cr.grad.abs().max()) * max(1.0, (n / 4096) ** 0.5)


def nimble_sized_mesh(B, seed, V=5990):
    """A closed genus-0 mesh with the NIMBLE skin's counts (V = 5990 vertices, F = 2 V - 4 = 11976 faces; reference
    models_res_nimble.py:135-136): a UV sphere of 499 rings x 12 segments + 2 poles, hand-sized (about 0.1 x 0.06 x 0.18 m), with a
    seeded bumpy radial displacement per batch item so that layers of surface overlap in the image like fingers do.  Synthetic: the
    real NIMBLE assets are not available."""
    rings, segs = (V - 2) // 12, 12
    assert rings * segs + 2 == V
    gen = torch.Generator().manual_seed(seed)
    th = (torch.arange(rings, dtype=torch.float32) + 1) / (rings + 1) * np.pi
    ph = torch.arange(segs, dtype=torch.float32) / segs * 2 * np.pi
    T, P = torch.meshgrid(th, ph, indexing="ij")
    base = torch.stack([torch.sin(T) * torch.cos(P), torch.sin(T) * torch.sin(P), torch.cos(T)], -1).reshape(-1, 3)
    base = torch.cat([torch.tensor([[0.0, 0.0, 1.0]]), base, torch.tensor([[0.0, 0.0, -1.0]])], 0)                       # [V,3]
    faces = []
    for j in range(segs):
        faces.append((0, 1 + j, 1 + (j + 1) % segs))
        last = 1 + (rings - 1) * segs
        faces.append((V - 1, last + (j + 1) % segs, last + j))
    for i in range(rings - 1):
        for j in range(segs):
            a, b = 1 + i * segs + j, 1 + i * segs + (j + 1) % segs
            c, d = a + segs, b + segs
            faces.append((a, c, b)); faces.append((b, c, d))
    faces = np.asarray(faces, dtype=np.int32)
    assert faces.shape[0] == 2 * V - 4
    k = torch.randn(B, 6, 3, generator=gen) * 3.0
    amp = 0.25 * torch.rand(B, 6, generator=gen)
    bump = 1.0 + (amp.unsqueeze(1) * torch.sin(torch.einsum("vc,bkc->bvk", base, k))).sum(-1)                                # [B,V]
    verts = base.unsqueeze(0) * bump.unsqueeze(-1) * torch.tensor([0.05, 0.03, 0.09])
    return verts, faces


def render_case(lib, tables, device, B, seed, image_size, aa, check_grad=True, rgb_atol=2e-5, gtol=2e-3, mesh=None, point_lights=False):
    from oracle import render_oracle as ro
    verts, vcol, cam, lc, ld = make_render_inputs(tables, B, seed, image_size)
    if point_lights:                            # PointLights defaults: diffuse .3, location (0, 1, 0) (+ a second, closer location)
        lc = torch.full((B, 3), 0.3)
        ld = torch.tensor([[0.0, 1.0, 0.0]]).repeat(B, 1)
        if B > 1:
            ld[1] = torch.tensor([0.1, -0.2, 0.3])
    faces_np, V = tables.faces, 778
    if mesh is not None:                        # another mesh in place of the hand: (verts [B,V,3] around the origin, faces [F,3])
        mv, faces_np = mesh
        V = mv.shape[1]
        gen = torch.Generator().manual_seed(seed + 5)
        verts = mv + verts.mean(1, keepdim=True)                    # same placements in front of the camera
        vcol = 0.3 + 0.6 * torch.rand(B, V, 3, generator=gen)
    faces = torch.as_tensor(faces_np).long()
    vr, cr, lcr, ldr = (t.clone().requires_grad_(True) for t in (verts, vcol, lc, ld))
    rgba_ref, p2f_ref = ro.render(vr, cr, cam, lcr, ldr, faces, image_size=image_size, aa=aa, point_lights=point_lights)
    S = image_size * aa
    h = lib.renderer_create(faces_np, V, image_size=image_size, aa=aa)
    if point_lights:
        lib.renderer_set_light_mode(h, True)
    try:
        ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device=device)
        d = lambda t: t.to(device).contiguous()
        rgba = torch.empty(B, 4, image_size, image_size, device=device)
        fid = torch.empty(B, S, S, dtype=torch.int32, device=device)
        dv, dc, dcam, dlc, dld = d(verts), d(vcol), d(cam), d(lc), d(ld)
        lib.render_fwd(h, dv, dc, dcam, dlc, dld, rgba, fid, ws)
        fid_np = fid.cpu().numpy()
        assert (p2f_ref >= 0).mean() > 0.02, "test mesh barely visible"
        np.testing.assert_array_equal(fid_np, p2f_ref)             # bit-exact face indices
        np.testing.assert_allclose(rgba.cpu().numpy(), rgba_ref.detach().numpy(), atol=rgb_atol, rtol=0)
        if not check_grad:
            return
        gen = torch.Generator().manual_seed(seed + 1)
        w = torch.randn(B, 4, image_size, image_size, generator=gen)
        (rgba_ref * w).sum().backward()
        gv = torch.empty(B, V, 3, device=device); gc = torch.empty(B, V, 3, device=device)
        glc = torch.empty(B, 3, device=device); gld = torch.empty(B, 3, device=device)
        lib.render_bwd(h, dv, dcam, dlc, dld, fid, d(w), gv, gc, glc, gld, ws)
        checks = [("verts", gv, vr.grad), ("vcolors", gc, cr.grad), ("light_color", glc, lcr.grad)]
        if point_lights:
            assert float(gld.abs().max()) == 0.0           # the location is a constant of the default-lighting branch
        else:
            checks.append(("light_dir", gld, ldr.grad))
        for name, got, ref in checks:
            ref = ref.numpy()
            scale = np.abs(ref).max() + 1e-12
            err = np.abs(got.cpu().numpy() - ref).max() / scale
            assert err < gtol, f"grad {name}: max err / max |ref| = {err:.3e} (scale {scale:.3e})"
        # a SECOND backward after the one forward: the accumulators the forward's vertex kernel zeroed are dirty now, the launcher clears them
        gv2 = torch.full((B, V, 3), 7.0, device=device); gc2 = torch.full((B, V, 3), 7.0, device=device)
        glc2 = torch.full((B, 3), 7.0, device=device); gld2 = torch.full((B, 3), 7.0, device=device)
        lib.render_bwd(h, dv, dcam, dlc, dld, fid, d(w), gv2, gc2, glc2, gld2, ws)
        for name, a, b2 in (("verts", gv, gv2), ("vcolors", gc, gc2), ("light_color", glc, glc2), ("light_dir", gld, gld2)):
            scale = float(a.abs().max()) + 1e-12
            assert float((a - b2).abs().max()) <= 1e-5 * scale, f"second backward on one forward's workspace: grad {name} differs"
    finally:
        lib.renderer_destroy(h)


def synthetic_uv_tables(faces_np, V, seed=0):
    """A UV layout for any mesh: one uv per (face, corner) -- faces_uvs = arange(3 F) -- placed by a seeded planar map of the vertex index
    plus a per-face jitter, all inside [0.05, 0.95] (texture seams everywhere: the general TexturesUV case)."""
    rng = np.random.default_rng(seed)
    F_ = len(faces_np)
    base = rng.random((V, 2)).astype(np.float32) * 0.8 + 0.1
    uv = base[np.asarray(faces_np).reshape(-1)] + (rng.random((3 * F_, 2)).astype(np.float32) - 0.5) * 0.1
    return np.arange(3 * F_, dtype=np.int32).reshape(F_, 3), np.clip(uv, 0.05, 0.95).astype(np.float32)


def render_uv_case(lib, tables, device, B, seed, image_size, aa, TH=24, TW=40, rgb_atol=3e-5, gtol=3e-3, uv_scale=1.0):
    """hifihr_render_fwd_uv / _bwd_uv vs oracle/render_oracle.render(textures_uv=...) ([recalled] PyTorch3D TexturesUV semantics through
    torch's own grid_sample): face ids exact, pixels, and the gradients w.r.t. vertices (incl. the path through uv), texture maps, light."""
    from oracle import render_oracle as ro
    verts, _, cam, lc, ld = make_render_inputs(tables, B, seed, image_size)
    faces_np, V = tables.faces, 778
    faces = torch.as_tensor(faces_np).long()
    fu, vu = synthetic_uv_tables(faces_np, V, seed)
    if uv_scale != 1.0:                         # uvs outside [0, 1]: grid_sample's border padding (clamped coordinate, zero uv gradient there)
        vu = ((vu - 0.5) * uv_scale + 0.5).astype(np.float32)
        assert (vu < 0).any() and (vu > 1).any()
    gen = torch.Generator().manual_seed(seed + 3)
    maps = torch.rand(B, TH, TW, 3, generator=gen)
    vr, mr, lcr, ldr = (t.clone().requires_grad_(True) for t in (verts, maps, lc, ld))
    rgba_ref, p2f_ref = ro.render(vr, None, cam, lcr, ldr, faces, image_size=image_size, aa=aa,
                                  textures_uv=(mr, torch.from_numpy(fu).long(), torch.from_numpy(vu)))
    S = image_size * aa
    h = lib.renderer_create(faces_np, V, image_size=image_size, aa=aa)
    try:
        lib.renderer_set_uv(h, fu, vu)
        ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device=device)
        d = lambda t: t.to(device).contiguous()
        rgba = torch.empty(B, 4, image_size, image_size, device=device)
        fid = torch.empty(B, S, S, dtype=torch.int32, device=device)
        texels = gtexels = None                                     # (reserved arguments: the texture is sampled inside the tile kernels)
        assert lib.render_uv_scratch_bytes(h, B) == 0
        dv, dm, dcam, dlc, dld = d(verts), d(maps), d(cam), d(lc), d(ld)
        lib.render_fwd_uv(h, dv, dm, dcam, dlc, dld, rgba, fid, texels, ws)
        assert (p2f_ref >= 0).mean() > 0.02, "test mesh barely visible"
        np.testing.assert_array_equal(fid.cpu().numpy(), p2f_ref)
        np.testing.assert_allclose(rgba.cpu().numpy(), rgba_ref.detach().numpy(), atol=rgb_atol, rtol=0)
        w = torch.randn(B, 4, image_size, image_size, generator=torch.Generator().manual_seed(seed + 1))
        (rgba_ref * w).sum().backward()
        gv = torch.empty(B, V, 3, device=device); gm = torch.zeros(B, TH, TW, 3, device=device)
        glc = torch.empty(B, 3, device=device); gld = torch.empty(B, 3, device=device)
        lib.render_bwd_uv(h, dv, dm, dcam, dlc, dld, fid, d(w), texels, gtexels, gv, gm, glc, gld, ws)
        for name, got, ref in (("verts", gv, vr.grad), ("maps", gm, mr.grad), ("light_color", glc, lcr.grad), ("light_dir", gld, ldr.grad)):
            ref = ref.numpy()
            scale = np.abs(ref).max() + 1e-12
            err = np.abs(got.cpu().numpy() - ref).max() / scale
            assert err < gtol, f"grad {name}: max err / max |ref| = {err:.3e} (scale {scale:.3e})"
    finally:
        lib.renderer_destroy(h)


# ------------------------------------------------------------------------------------------------
# fused Adam
# ------------------------------------------------------------------------------------------------
def adam_case(lib, device, n, wd, steps, grad_scale=0.5, lr=1e-3):
    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    p = p0.clone().to(device); m = torch.zeros(n, device=device); v = torch.zeros(n, device=device)
    for s in range(1, steps + 1):
        g = torch.randn(n, generator=gen)
        ref.grad = (g * grad_scale).clone()
        opt.step()
        if s % 2:
            lib.adam_step(p, g.to(device), m, v, grad_scale, lr, 0.9, 0.999, 1e-8, wd, s)
        else:           # the graph-replayable variant: scalars from device memory
            dyn = torch.tensor([lr / (1 - 0.9 ** s), 1.0 / (1 - 0.999 ** s) ** 0.5], dtype=torch.float32).to(device)
            lib.adam_step_dyn(p, g.to(device), m, v, grad_scale, 0.9, 0.999, 1e-8, wd, dyn)
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    # the counted variant (step counter and lr in device memory, bias corrections derived by the kernel): the same trajectory, and the counter
    # the last workgroup advances ends at `steps`; started from a state image at step 1 to cover a restored counter
    gen = torch.Generator().manual_seed(n)
    torch.randn(n, generator=gen)
    pc = p0.clone().to(device); mc = torch.zeros(n, device=device); vc = torch.zeros(n, device=device)
    state = None
    for s in range(1, steps + 1):
        g = torch.randn(n, generator=gen)
        if s == 1:
            lib.adam_step(pc, g.to(device), mc, vc, grad_scale, lr, 0.9, 0.999, 1e-8, wd, 1)
            state = lib.adam_state_image(lr, 0.9, 0.999, 1).to(device)
        else:
            lib.adam_step_counted(pc, g.to(device), mc, vc, grad_scale, 1e-8, wd, state)
    np.testing.assert_allclose(pc.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    import struct
    lr_d, b1_d, b2_d, p1_d, p2_d, step_d, done_d = struct.unpack("<dddddii", bytes(state.cpu().numpy().tobytes()))
    assert (lr_d, step_d, done_d) == (lr, steps, 0), (lr_d, step_d, done_d)
    assert abs(p1_d - 0.9 ** steps) <= 1e-14 and abs(p2_d - 0.999 ** steps) <= 1e-14


# ------------------------------------------------------------------------------------------------
# convolution (NHWC implicit GEMM on the f32 matrix cores) vs plain PyTorch fp32 conv2d
# ------------------------------------------------------------------------------------------------
def conv_case(lib, device, N, H, W, C, K, R, stride, pad, seed=0, bias=False, rtol=2e-5, zero_last_channel=False):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen)
    w = torch.randn(K, C, R, R, generator=gen) / (C * R * R) ** 0.5
    if zero_last_channel:                       # the encoder's NHWC4 stem: channel 3 of the image and of the filter is padding
        x[:, C - 1] = 0.0; w[:, C - 1] = 0.0
    b = torch.randn(K, generator=gen) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, b, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    d = lambda t: t.to(device).contiguous()
    x_nhwc = d(x.permute(0, 2, 3, 1)); w_krsc = d(w.permute(0, 2, 3, 1)); gy_nhwc = d(gy.permute(0, 2, 3, 1))
    out = torch.empty(N, OH, OW, K, device=device)
    lib.conv2d_fwd(x_nhwc, w_krsc, d(b) if bias else None, out, N, H, W, C, K, R, R, stride, pad)
    ref = y.detach().permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((out.cpu() - ref).abs().max()) <= rtol * scale + 1e-6, "conv fwd"
    dx = torch.empty(N, H, W, C, device=device)
    scratch = torch.empty(K * R * R * C, device=device)
    lib.conv2d_bwd_data(gy_nhwc, w_krsc, dx, scratch, N, H, W, C, K, R, R, stride, pad)
    refx = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu() - refx).abs().max()) <= rtol * float(refx.abs().max()) + 1e-6, "conv bwd data"
    dw = torch.zeros(K, R, R, C, device=device)
    lib.conv2d_bwd_weight(x_nhwc, gy_nhwc, dw, N, H, W, C, K, R, R, stride, pad)
    refw = wr.grad.permute(0, 2, 3, 1)
    assert float((dw.cpu() - refw).abs().max()) <= 5 * rtol * float(refw.abs().max()) + 1e-6, "conv bwd weight"
    # accumulation semantics: a second call adds
    lib.conv2d_bwd_weight(x_nhwc, gy_nhwc, dw, N, H, W, C, K, R, R, stride, pad)
    assert float((dw.cpu() - 2 * refw).abs().max()) <= 1e-4 * float(refw.abs().max()) + 1e-6, "conv bwd weight accumulate"
    # slab forms of the weight gradient (caller's scratch, any contents): same gradient, bit-reproducible
    nws = lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, R, stride, pad)
    if nws:
        wsw = torch.full((nws // 4,), 3.0, device=device)
        dws = torch.full((K, R, R, C), 0.5, device=device)
        lib.conv2d_bwd_weight(x_nhwc, gy_nhwc, dws, N, H, W, C, K, R, R, stride, pad, ws=wsw)
        assert float((dws.cpu() - 0.5 - refw).abs().max()) <= 5 * rtol * float(refw.abs().max()) + 1e-6, "conv bwd weight (slabs)"
        dws2 = torch.full((K, R, R, C), 0.5, device=device)
        lib.conv2d_bwd_weight(x_nhwc, gy_nhwc, dws2, N, H, W, C, K, R, R, stride, pad, ws=wsw)
        assert torch.equal(dws, dws2), "slab weight gradient must be bit-reproducible"
    # balanced (stream-K) schedule through a zero-initialised, self-cleaning workspace: same results, workspace zero again
    nb_f = lib.conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad, False)
    nb_b = lib.conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad, True)
    used = 0
    if nb_f and not bias:
        ws = torch.zeros(nb_f // 4, device=device); out2 = torch.full_like(out, 7.0)
        for _ in range(2):                                  # twice: the second call must find the workspace clean
            lib.conv2d_fwd(x_nhwc, w_krsc, None, out2, N, H, W, C, K, R, R, stride, pad, ws=ws)
            assert float((out2.cpu() - ref).abs().max()) <= rtol * scale + 1e-6, "conv fwd (balanced schedule)"
            assert float(ws.abs().max()) == 0.0, "workspace not cleaned"
        used += 1
    if nb_b:
        ws = torch.zeros(nb_b // 4, device=device); dx2 = torch.full_like(dx, 7.0)
        lib.conv2d_bwd_data(gy_nhwc, w_krsc, dx2, scratch, N, H, W, C, K, R, R, stride, pad, ws=ws)
        assert float((dx2.cpu() - refx).abs().max()) <= rtol * float(refx.abs().max()) + 1e-6, "conv bwd data (balanced schedule)"
        assert float(ws.abs().max()) == 0.0, "workspace not cleaned"
        used += 1
    return used


def image_to_nhwc4_case(lib, device):
    from oracle.torch_modules import normalize_batch_3C
    img = torch.rand(3, 3, 20, 12)
    out = torch.empty(3, 20, 12, 4, device=device)
    lib.image_to_nhwc4(img.to(device), out)
    ref = normalize_batch_3C(img).permute(0, 2, 3, 1)
    np.testing.assert_allclose(out.cpu()[..., :3].numpy(), ref.numpy(), atol=1e-6)
    assert float(out.cpu()[..., 3].abs().max()) == 0.0
    # padded, un-normalised variant (EfficientNet stem): F.pad(img, (left, right, top, bottom)) as NHWC4
    import torch.nn.functional as F
    pad4 = (0, 1, 0, 1)
    out2 = torch.full((3, 21, 13, 4), 7.0, device=device)
    lib.image_to_nhwc4_padded(img.to(device), out2, pad4, False)
    ref2 = F.pad(img, pad4).permute(0, 2, 3, 1)
    assert torch.equal(out2.cpu()[..., :3], ref2) and float(out2.cpu()[..., 3].abs().max()) == 0.0
    out3 = torch.full((3, 23, 15, 4), 7.0, device=device)
    lib.image_to_nhwc4_padded(img.to(device), out3, (2, 1, 1, 2), True)
    ref3 = F.pad(normalize_batch_3C(img), (2, 1, 1, 2)).permute(0, 2, 3, 1)
    np.testing.assert_allclose(out3.cpu()[..., :3].numpy(), ref3.numpy(), atol=1e-6)


# ------------------------------------------------------------------------------------------------
# fused SSIM vs vectors produced by the reference's utils/pytorch_ssim (tests/golden/ssim.npz) and vs the
# torch restatement in oracle/loss_oracle.py (itself pinned to the same vectors on the CPU)
# ------------------------------------------------------------------------------------------------
def ssim_case(lib, device, a, b, ref_val=None, ref_grad=None):
    from oracle import loss_oracle as L
    from hifihr_amd.ops import _ssim_window
    win = _ssim_window()
    a_t, b_t = torch.as_tensor(a), torch.as_tensor(b)
    if ref_val is None:
        ar = a_t.clone().requires_grad_(True)
        v = L.ssim(ar, b_t)
        v.backward()
        ref_val, ref_grad = float(v), ar.grad.numpy()
    B, C, H, W = a_t.shape
    da, db = a_t.to(device).contiguous(), b_t.to(device).contiguous()
    partial = torch.empty(lib.ssim_partial_count(B * C, H, W), device=device)
    maps = torch.empty(3, B, C, H, W, device=device)
    lib.ssim_fwd(win, da, db, partial, maps[0], maps[1], maps[2])
    val = float(partial.sum()) / (B * C * H * W)
    assert abs(val - float(ref_val)) <= 2e-6 * max(1.0, abs(float(ref_val))), (val, float(ref_val))
    g = torch.empty_like(da)
    lib.ssim_bwd(win, da, db, maps[0], maps[1], maps[2], torch.full((1,), 2.0, device=device), g)
    ref = 2.0 * np.asarray(ref_grad)
    assert np.abs(g.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-10
    # the call site's scalar glue: lambda * (1 - SSIM) in one finishing launch, the gradient scale folded into the backward
    n, lam = B * C * H * W, 0.37
    out = torch.full((), 7.0, device=device)
    lib.ssim_finish(partial, 1.0 / n, 0.0, out)
    assert abs(float(out) - val) <= 2e-6 * max(1.0, abs(val))
    lib.ssim_finish(partial, -lam / n, lam, out)
    assert abs(float(out) - lam * (1.0 - val)) <= 2e-6
    g2 = torch.empty_like(da)
    lib.ssim_bwd_scaled(win, da, db, maps[0], maps[1], maps[2], torch.full((1,), 2.0, device=device), -lam, g2)
    assert float((g2 + lam * g).abs().max()) <= 1e-6 * float(g.abs().max()) + 1e-12


# ------------------------------------------------------------------------------------------------
# fused train-mode BatchNorm (+ residual add + ReLU), NHWC, vs plain PyTorch fp32 (F.batch_norm + add + relu autograd)
# ------------------------------------------------------------------------------------------------
def bn_act_case(lib, device, N, H, W, C, relu, residual, seed=0, from_conv=False):
    """relu: True / False / "swish"."""
    act = {False: 0, True: 1, "swish": 2}[relu]
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    M = N * H * W
    x = torch.randn(N, C, H, W, generator=gen) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=gen); beta = 0.1 * torch.randn(C, generator=gen)
    res = torch.randn(N, C, H, W, generator=gen) if residual else None
    rm0, rv0 = torch.randn(C, generator=gen) * 0.1, 1 + 0.1 * torch.rand(C, generator=gen)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    rm, rv = rm0.clone(), rv0.clone()
    out = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    if residual:
        out = out + rr
    if act == 1:
        out = F.relu(out)
    elif act == 2:
        out = out * torch.sigmoid(out)
    gy = torch.randn(out.shape, generator=gen)
    out.backward(gy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(device)
    dx_in = nhwc(x)
    stats = torch.zeros(lib.bn_stats_floats(C), device=device)      # zero on entry (self-cleaning contract)
    lib.bn_stats(dx_in, M, C, stats)
    np.testing.assert_allclose(bn_slots(stats, C).sum(0)[0].cpu().numpy(), x.permute(1, 0, 2, 3).reshape(C, -1).sum(1).numpy(),
                               rtol=1e-4, atol=1e-3)
    y = torch.empty(N, H, W, C, device=device); sm = torch.empty(C, device=device); si = torch.empty(C, device=device)
    rmd, rvd = rm0.clone().to(device), rv0.clone().to(device)
    lib.bn_act_fwd(dx_in, stats, gamma.to(device), beta.to(device), nhwc(res) if residual else None, act, M, C, 1e-5, 0.1, y, sm, si, rmd, rvd)
    assert float(stats.abs().max()) == 0.0, "bn_act_fwd must leave the slots and arrival counters zeroed"
    ref = out.detach().permute(0, 2, 3, 1)
    assert float((y.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), "bn fwd"
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    red = torch.zeros(lib.bn_stats_floats(C), device=device); dxo = torch.empty_like(y); dres = torch.empty_like(y) if residual else None
    dg = torch.full((C,), 0.5, device=device); db = torch.full((C,), -0.25, device=device)     # accumulate semantics
    lib.bn_act_bwd(nhwc(gy), y if act == 1 else None, dx_in, sm, si, gamma.to(device), beta.to(device), act, M, C, red, dxo, dres, dg, db)
    assert float(red[:32 * 4 * C + 64].abs().max()) == 0.0, "bn_act_bwd must leave the slots and arrival counters zeroed"
    if act == 1 and not residual:       # ReLU mask recomputed from x instead of read from y: the same gradients (the slot sums
        #                                 are float atomics, so two launches agree to rounding, not to the bit)
        dxo2 = torch.full_like(dxo, 7.0); dg2 = torch.zeros_like(dg); db2 = torch.zeros_like(db)
        lib.bn_act_bwd(nhwc(gy), None, dx_in, sm, si, gamma.to(device), beta.to(device), act, M, C, red, dxo2, None, dg2, db2)
        dg1 = torch.zeros_like(dg); db1 = torch.zeros_like(db); dxo1 = torch.full_like(dxo, 7.0)
        lib.bn_act_bwd(nhwc(gy), y, dx_in, sm, si, gamma.to(device), beta.to(device), act, M, C, red, dxo1, None, dg1, db1)
        scale = float(dxo1.abs().max())
        assert float((dxo1 - dxo2).abs().max()) <= 2e-6 * scale, "mask recomputed from x: dx"
        assert float(((dxo1 == 0) != (dxo2 == 0)).float().mean()) <= 1e-5, "mask recomputed from x: same zero pattern"
        assert float((dg1 - dg2).abs().max()) <= 1e-5 * float(dg1.abs().max()) + 1e-6 and float((db1 - db2).abs().max()) <= 1e-5 * float(db1.abs().max()) + 1e-6
    refdx = xr.grad.permute(0, 2, 3, 1)
    assert float((dxo.cpu() - refdx).abs().max()) <= 2e-4 * float(refdx.abs().max()) + 1e-7, "bn bwd dx"
    assert float((dg.cpu() - 0.5 - gr.grad).abs().max()) <= 2e-4 * float(gr.grad.abs().max()) + 1e-5, "bn dgamma"
    assert float((db.cpu() + 0.25 - br.grad).abs().max()) <= 2e-4 * float(br.grad.abs().max()) + 1e-5, "bn dbeta"
    if residual:
        assert float((dres.cpu() - rr.grad.permute(0, 2, 3, 1)).abs().max()) <= 1e-6, "bn dres"


def conv_bnstats_case(lib, device, N, H, W, C, K, R, stride, pad, seed=0, use_ws=False):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen); w = torch.randn(K, C, R, R, generator=gen) / (C * R * R) ** 0.5
    y = F.conv2d(x, w, None, stride, pad)
    OH, OW = y.shape[2], y.shape[3]
    d = lambda t: t.to(device).contiguous()
    out = torch.empty(N, OH, OW, K, device=device); stats = torch.zeros(lib.bn_stats_floats(K), device=device)
    nb = lib.conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad, False) if use_ws else 0
    ws = torch.zeros(nb // 4, device=device) if nb else None
    assert not use_ws or nb > 0, "this case was meant to exercise the balanced schedule"
    lib.conv2d_fwd_bnstats(d(x.permute(0, 2, 3, 1)), d(w.permute(0, 2, 3, 1)), out, stats, N, H, W, C, K, R, R, stride, pad, ws=ws)
    assert ws is None or float(ws.abs().max()) == 0.0
    ref = y.permute(0, 2, 3, 1)
    assert float((out.cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-6
    s_ref = y.permute(1, 0, 2, 3).reshape(K, -1)
    st = bn_slots(stats, K).sum(0)
    np.testing.assert_allclose(st[0].cpu().numpy(), s_ref.sum(1).numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(st[1].cpu().numpy(), (s_ref ** 2).sum(1).numpy(), rtol=1e-4, atol=2e-3)


def conv_fwd_pair_case(lib, device, N, H, W, C, K1, K2, seed=0):
    """hifihr_conv2d_fwd_bnstats_pair: the strided 3x3 convolution of a residual stage's first block and the stride-2 1x1 convolution of its
    downsample branch in ONE launch == the two hifihr_conv2d_fwd_bnstats calls: outputs bit for bit, folded statistics to the f32 rounding of the per-share partial sums."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen); w1 = torch.randn(K1, C, 3, 3, generator=gen) / (9 * C) ** 0.5; w2 = torch.randn(K2, C, 1, 1, generator=gen) / C ** 0.5
    r1, r2 = F.conv2d(x, w1, None, 2, 1), F.conv2d(x, w2, None, 2, 0)
    OH, OW = r1.shape[2], r1.shape[3]
    assert r2.shape[2:] == r1.shape[2:]
    d = lambda t: t.to(device).contiguous()
    xd, w1d, w2d = d(x.permute(0, 2, 3, 1)), d(w1.permute(0, 2, 3, 1)), d(w2.permute(0, 2, 3, 1))
    assert lib.conv2d_fwd_bnstats_pair_supported(N, H, W, C, 2, K1, 3, 1, K2, 1, 0)
    ya, yb = torch.full((N, OH, OW, K1), 7.0, device=device), torch.full((N, OH, OW, K2), 7.0, device=device)
    sa, sb = torch.zeros(lib.bn_stats_floats(K1), device=device), torch.zeros(lib.bn_stats_floats(K2), device=device)
    lib.conv2d_fwd_bnstats_pair(xd, w1d, ya, sa, K1, 3, 1, w2d, yb, sb, K2, 1, 0, N, H, W, C, 2)
    y1, y2 = torch.empty_like(ya), torch.empty_like(yb)
    s1, s2 = torch.zeros_like(sa), torch.zeros_like(sb)
    lib.conv2d_fwd_bnstats(xd, w1d, y1, s1, N, H, W, C, K1, 3, 3, 2, 1)
    lib.conv2d_fwd_bnstats(xd, w2d, y2, s2, N, H, W, C, K2, 1, 1, 2, 0)
    assert torch.equal(ya, y1) and torch.equal(yb, y2), "pair: outputs differ from the separate launches"
    for K, p, q in ((K1, sa, s1), (K2, sb, s2)):
        a, b = bn_slots(p, K).sum(0).double().cpu(), bn_slots(q, K).sum(0).double().cpu()
        # (the shifted partial sums are f32 inside a workgroup's share, and the shares differ between the two forms: equal to f32 rounding of the
        #  per-share partials, observed 3e-8 of the largest sum)
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-9
    assert float((ya.cpu() - r1.permute(0, 2, 3, 1)).abs().max()) <= 3e-5 * float(r1.abs().max()) + 1e-6
    assert float((yb.cpu() - r2.permute(0, 2, 3, 1)).abs().max()) <= 3e-5 * float(r2.abs().max()) + 1e-6


def conv_dgrad_plus1x1_case(lib, device, N, H, W, C, K, stride=2, seed=0):
    """hifihr_conv2d_bwd_data_pre_plus1x1: backward-data of the strided 3x3 convolution with the data gradient of the 1x1 / same stride / pad 0
    convolution of the same input as one more tap of parity class (0, 0) == torch autograd of conv(x, w1) + conv(x, w2) wrt x, and ==
    the two separate launches (hifihr_conv2d_bwd_data_pre on the 1x1, its result as the residual of hifihr_conv2d_bwd_data_pre_res on the
    3x3) to the rounding of one more term per pixel of that class."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen).requires_grad_(True)
    w1 = torch.randn(K, C, 3, 3, generator=gen) / (9 * C) ** 0.5; w2 = torch.randn(K, C, 1, 1, generator=gen) / C ** 0.5
    y1, y2 = F.conv2d(x, w1, None, stride, 1), F.conv2d(x, w2, None, stride, 0)
    assert y1.shape == y2.shape
    g1, g2 = torch.randn(y1.shape, generator=gen), torch.randn(y2.shape, generator=gen)
    (y1 * g1).sum().backward(retain_graph=True)
    ref1 = x.grad.clone()
    x.grad = None
    ((y1 * g1).sum() + (y2 * g2).sum()).backward()
    ref = x.grad.permute(0, 2, 3, 1)
    d = lambda t: t.to(device).contiguous()
    g1d, g2d = d(g1.permute(0, 2, 3, 1)), d(g2.permute(0, 2, 3, 1))
    w1d, w2d = d(w1.permute(0, 2, 3, 1)), d(w2.permute(0, 2, 3, 1))                   # [K][R][S][C]
    wt1, wt2 = torch.empty(w1d.numel(), device=device), torch.empty(w2d.numel(), device=device)
    lib.weight_transpose(w1d, wt1, K, 9, C); lib.weight_transpose(w2d, wt2, K, 1, C)  # [C][R][S][K]
    assert lib.conv2d_bwd_data_pre_plus1x1_supported(N, H, W, C, K, 3, 3, stride, 1)
    dx = torch.full((N, H, W, C), 7.0, device=device)
    lib.conv2d_bwd_data_pre_plus1x1(g1d, wt1, g2d, wt2, dx, N, H, W, C, K, 3, 3, stride, 1)
    scale = float(ref.abs().max())
    assert float((dx.cpu() - ref).abs().max()) <= 3e-5 * scale + 1e-6, float((dx.cpu() - ref).abs().max())
    # the two-launch form it replaces
    dx2 = torch.full((N, H, W, C), 7.0, device=device); dxs = torch.full((N, H, W, C), 7.0, device=device)
    lib.conv2d_bwd_data_pre(g2d, wt2, dx2, N, H, W, C, K, 1, 1, stride, 0)
    lib.conv2d_bwd_data_pre_res(g1d, wt1, dx2, dxs, N, H, W, C, K, 3, 3, stride, 1)
    assert float((dx - dxs).abs().max()) <= 2e-6 * scale + 1e-7, float((dx - dxs).abs().max())
    # the pixels off class (0, 0) carry the 3x3 convolution's gradient alone: bit for bit what the plain launch gives
    dx1 = torch.full((N, H, W, C), 7.0, device=device)
    lib.conv2d_bwd_data_pre(g1d, wt1, dx1, N, H, W, C, K, 3, 3, stride, 1)
    off = torch.ones(H, W, dtype=torch.bool); off[::stride, ::stride] = False
    assert torch.equal(dx.cpu()[:, off], dx1.cpu()[:, off])
    assert float((dx1.cpu() - ref1.permute(0, 2, 3, 1)).abs().max()) <= 3e-5 * scale + 1e-6


def conv_wgrad_plus1x1_case(lib, device, N, H, W, C, K, stride=2, seed=0):
    """hifihr_conv2d_bwd_weight_plus1x1: the weight gradients of the strided 3x3 convolution and of the 1x1 / same stride / pad 0 convolution
    of the same input in one launch == torch autograd, and == hifihr_conv2d_bwd_weight on each (float atomics: to rounding); both ACCUMULATE."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen)
    w1 = (torch.randn(K, C, 3, 3, generator=gen) / (9 * C) ** 0.5).requires_grad_(True); w2 = (torch.randn(K, C, 1, 1, generator=gen) / C ** 0.5).requires_grad_(True)
    y1, y2 = F.conv2d(x, w1, None, stride, 1), F.conv2d(x, w2, None, stride, 0)
    g1, g2 = torch.randn(y1.shape, generator=gen), torch.randn(y2.shape, generator=gen)
    ((y1 * g1).sum() + (y2 * g2).sum()).backward()
    r1, r2 = w1.grad.permute(0, 2, 3, 1), w2.grad.permute(0, 2, 3, 1)
    d = lambda t: t.to(device).contiguous()
    xd, g1d, g2d = d(x.permute(0, 2, 3, 1)), d(g1.permute(0, 2, 3, 1)), d(g2.permute(0, 2, 3, 1))
    assert lib.conv2d_bwd_weight_plus1x1_supported(N, H, W, C, K, 3, 3, stride, 1)
    dw1 = torch.full((K, 3, 3, C), 0.5, device=device); dw2 = torch.full((K, 1, 1, C), -0.25, device=device)     # accumulated into
    lib.conv2d_bwd_weight_plus1x1(xd, g1d, dw1, g2d, dw2, N, H, W, C, K, 3, 3, stride, 1)
    for got, ref, base in ((dw1, r1, 0.5), (dw2, r2, -0.25)):
        sc = float(ref.abs().max())
        assert float((got.cpu() - base - ref).abs().max()) <= 3e-5 * sc + 1e-5, float((got.cpu() - base - ref).abs().max())
    s1 = torch.zeros(K, 3, 3, C, device=device); s2 = torch.zeros(K, 1, 1, C, device=device)
    lib.conv2d_bwd_weight(xd, g1d, s1, N, H, W, C, K, 3, 3, stride, 1)
    lib.conv2d_bwd_weight(xd, g2d, s2, N, H, W, C, K, 1, 1, stride, 0)
    assert float((dw1 - 0.5 - s1).abs().max()) <= 1e-5 * float(s1.abs().max()) + 1e-6
    assert float((dw2 + 0.25 - s2).abs().max()) <= 1e-5 * float(s2.abs().max()) + 1e-6


# ------------------------------------------------------------------------------------------------
# depthwise convolution (EfficientNet MBConv) vs plain PyTorch fp32 (F.pad + grouped F.conv2d autograd)
# ------------------------------------------------------------------------------------------------
def dwconv_case(lib, device, N, H, W, C, K, stride, seed=0):
    import torch.nn.functional as F
    from hifihr_amd.effnet import static_same_pad
    gen = torch.Generator().manual_seed(seed)
    pl, pr, pt, pb = static_same_pad(K, stride)
    x = torch.randn(N, C, H, W, generator=gen); w = torch.randn(C, 1, K, K, generator=gen) / K
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, None, stride, 0, 1, C)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    d = lambda t: t.to(device).contiguous()
    xd, wd, gyd = d(x.permute(0, 2, 3, 1)), d(w.reshape(C, K, K)), d(gy.permute(0, 2, 3, 1))
    out = torch.empty(N, OH, OW, C, device=device)
    stats = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.dwconv2d_fwd(xd, wd, out, N, H, W, C, OH, OW, K, stride, pt, pl, stats=stats)
    ref = y.detach().permute(0, 2, 3, 1)
    assert float((out.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6, "dw fwd"
    st = bn_slots(stats, C).sum(0).cpu()
    flat = ref.reshape(-1, C)
    np.testing.assert_allclose(st[0].numpy(), flat.sum(0).numpy(), rtol=1e-4, atol=2e-3, err_msg="dw fwd: batch-norm sum")
    np.testing.assert_allclose(st[1].numpy(), (flat ** 2).sum(0).numpy(), rtol=1e-4, atol=2e-3, err_msg="dw fwd: batch-norm sum of squares")
    out2 = torch.empty_like(out)
    lib.dwconv2d_fwd(xd, wd, out2, N, H, W, C, OH, OW, K, stride, pt, pl)            # without statistics
    assert torch.equal(out2, out)
    dx = torch.empty(N, H, W, C, device=device)
    lib.dwconv2d_bwd_data(gyd, wd, dx, N, H, W, C, OH, OW, K, stride, pt, pl)
    refx = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu() - refx).abs().max()) <= 2e-5 * float(refx.abs().max()) + 1e-6, "dw bwd data"
    dw = torch.zeros(C, K, K, device=device)
    lib.dwconv2d_bwd_weight(xd, gyd, dw, N, H, W, C, OH, OW, K, stride, pt, pl)
    refw = wr.grad.reshape(C, K, K)
    assert float((dw.cpu() - refw).abs().max()) <= 1e-4 * float(refw.abs().max()) + 1e-6, "dw bwd weight"


def dwconv_bnswish_case(lib, device, N, H, W, C, K, stride, seed=0, mean=0.3, std=1.5):
    """The expand half of an MBConv block without its activated tensor (hifihr_bn_finalize_fwd + hifihr_dwconv2d_fwd_bnswish /
    _bwd_weight_bnswish + hifihr_dwconv2d_bwd_data + hifihr_bn_act_bwd(swish)) vs plain PyTorch: nn.BatchNorm2d (training mode, eps
    1e-3, momentum 0.01 as the reference's blocks) -> x * sigmoid(x) -> F.pad + grouped F.conv2d, forward, every gradient, the
    running statistics; and bit for bit against the library's own unfused pair hifihr_bn_act_fwd + hifihr_dwconv2d_fwd."""
    import torch.nn.functional as F
    from hifihr_amd.effnet import static_same_pad
    gen = torch.Generator().manual_seed(seed)
    pl, pr, pt, pb = static_same_pad(K, stride)
    e = torch.randn(N, C, H, W, generator=gen) * std + mean
    w = torch.randn(C, 1, K, K, generator=gen) / K
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3
    eps, mom = 1e-3, 0.01
    bn = torch.nn.BatchNorm2d(C, eps=eps, momentum=mom).train()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    er, wr = e.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z = bn(er)
    a = z * torch.sigmoid(z)
    y = F.conv2d(F.pad(a, (pl, pr, pt, pb)), wr, None, stride, 0, 1, C)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    M = N * H * W
    d = lambda t: t.to(device).contiguous()
    ed, wd, gyd, gd, bd = d(e.permute(0, 2, 3, 1)), d(w.reshape(C, K, K)), d(gy.permute(0, 2, 3, 1)), d(gamma), d(beta)
    f = lambda *shape: torch.empty(*shape, device=device)
    # statistics of e as its producer leaves them (the slot buffer), consumed by the finalize call
    stats = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.bn_stats(ed, M, C, stats)
    stats2 = stats.clone()
    mean_d, invstd_d, rm, rv = f(C), f(C), torch.zeros(C, device=device), torch.ones(C, device=device)
    lib.bn_finalize_fwd(stats, M, C, eps, mom, mean_d, invstd_d, rm, rv)
    assert float(stats.abs().max()) == 0.0, "finalize must hand the slots back zeroed"
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    out = f(N, OH, OW, C)
    ystats = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.dwconv2d_fwd_bnswish(ed, mean_d, invstd_d, gd, bd, wd, out, N, H, W, C, OH, OW, K, stride, pt, pl, stats=ystats)
    ref = y.detach().permute(0, 2, 3, 1)
    err = float((out.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 3e-5, f"dw fwd (bn + swish on load): {err}"
    st = bn_slots(ystats, C).sum(0).cpu()
    flat = ref.reshape(-1, C)
    np.testing.assert_allclose(st[0].numpy(), flat.sum(0).numpy(), rtol=2e-4, atol=5e-3)
    np.testing.assert_allclose(st[1].numpy(), (flat ** 2).sum(0).numpy(), rtol=2e-4, atol=5e-3)
    # the unfused pair of the same library: bn_act_fwd (swish) then dwconv2d_fwd -- the same bits
    a_d, m2, i2 = f(N, H, W, C), f(C), f(C)
    lib.bn_act_fwd(ed, stats2, gd, bd, None, 2, M, C, eps, mom, a_d, m2, i2, None, None)
    assert torch.equal(m2, mean_d) and torch.equal(i2, invstd_d)
    out_u = f(N, OH, OW, C)
    lib.dwconv2d_fwd(a_d, wd, out_u, N, H, W, C, OH, OW, K, stride, pt, pl)
    eu = float((out_u - out).abs().max()) / float(ref.abs().max())
    assert eu <= 2e-6, f"fused vs unfused forward: {eu}"
    # weight gradient from e
    dw = torch.zeros(C, K, K, device=device)
    lib.dwconv2d_bwd_weight_bnswish(ed, mean_d, invstd_d, gd, bd, gyd, dw, N, H, W, C, OH, OW, K, stride, pt, pl)
    refw = wr.grad.reshape(C, K, K)
    assert float((dw.cpu() - refw).abs().max()) <= 2e-4 * float(refw.abs().max()) + 1e-6, "dw bwd weight (bn + swish on load)"
    # gradient wrt e, gamma, beta: dwconv_bwd_data then the fused batch-norm backward with act = swish on (d a, e)
    da = f(N, H, W, C)
    lib.dwconv2d_bwd_data(gyd, wd, da, N, H, W, C, OH, OW, K, stride, pt, pl)
    red = torch.zeros(lib.bn_stats_floats(C), device=device)
    de, dg, db = f(N, H, W, C), torch.zeros(C, device=device), torch.zeros(C, device=device)
    lib.bn_act_bwd(da, None, ed, mean_d, invstd_d, gd, bd, 2, M, C, red, de, None, dg, db)
    refe = er.grad.permute(0, 2, 3, 1)
    assert float((de.cpu() - refe).abs().max()) <= 3e-4 * float(refe.abs().max()) + 1e-7, "d e"
    assert float((dg.cpu() - bn.weight.grad).abs().max()) <= 3e-4 * float(bn.weight.grad.abs().max()) + 1e-5, "d gamma"
    assert float((db.cpu() - bn.bias.grad).abs().max()) <= 3e-4 * float(bn.bias.grad.abs().max()) + 1e-5, "d beta"


# ------------------------------------------------------------------------------------------------
# pooling (csrc/pool.hip) vs plain PyTorch fp32
# ------------------------------------------------------------------------------------------------
def mmpool_case(lib, device, B, H, W, C, p0=0.3, seed=0, ties=False):
    """MMPool((1,1)): adaptive max + adaptive avg mixed by sigmoid(p) (reference network/res_encoder.py:247-265)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen)
    if ties:
        x = torch.relu(x - 1.5)                    # mostly zeros with repeated maxima: first-index tie rule
    xr = x.clone().requires_grad_(True); pr = torch.tensor([p0], requires_grad=True)
    w = torch.sigmoid(pr)
    y = (F.adaptive_max_pool2d(xr, (1, 1)) * w + F.adaptive_avg_pool2d(xr, (1, 1)) * (1 - w)).reshape(B, C)
    gy = torch.randn(B, C, generator=gen)
    y.backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device); pd = torch.tensor([p0], device=device)
    out = torch.empty(B, C, device=device); am = torch.empty(B, C, dtype=torch.int32, device=device)
    xmax, xavg = torch.empty_like(out), torch.empty_like(out)
    lib.mmpool_fwd(xd, pd, B, H * W, C, out, am, xmax, xavg)
    assert float((out.cpu() - y.detach()).abs().max()) <= 1e-5 * max(1.0, float(y.detach().abs().max())), "mmpool fwd"
    _, ref_idx = F.adaptive_max_pool2d(x, (1, 1), return_indices=True)
    assert torch.equal(am.cpu().long(), ref_idx.reshape(B, C)), "mmpool argmax (first maximum in scan order)"
    dx = torch.empty(B, H, W, C, device=device); dp = torch.full((1,), 0.25, device=device)
    lib.mmpool_bwd(gy.to(device), pd, am, xmax, xavg, B, H * W, C, dx, dp)
    refdx = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu() - refdx).abs().max()) <= 1e-6 + 1e-5 * float(refdx.abs().max()), "mmpool dx"
    assert abs(float(dp.cpu()) - 0.25 - float(pr.grad)) <= 1e-4 * max(1.0, abs(float(pr.grad))), "mmpool dp (accumulates)"


def bn_relu_maxpool_case(lib, device, N, H, W, C, seed=0):
    """hifihr_bn_relu_maxpool_{fwd,bwd} vs (a) the unfused kernels of the same library (forward: the same bits; backward: to the
    rounding of the slot atomics) and (b) plain PyTorch fp32 autograd of MaxPool2d(3, 2, 1)(relu(batch_norm(x)))."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    M = N * H * W
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(N, C, H, W, generator=gen) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=gen); beta = 0.1 * torch.randn(C, generator=gen) - 0.3     # plenty of ReLU zeros: ties
    gamma[1] = 2e-4; beta[1] = 0.05                      # a near-zero scale (positive shift: every tap passes the ReLU)
    gamma[2] = -0.8                                      # a negative scale: the winner is the SMALLEST x of the window
    rm0, rv0 = torch.randn(C, generator=gen) * 0.1, 1 + 0.1 * torch.rand(C, generator=gen)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    out = F.max_pool2d(F.relu(F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)), 3, 2, 1)
    gy = torch.randn(out.shape, generator=gen)
    out.backward(gy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(device)
    xd, gd, bd, gyd = nhwc(x), gamma.to(device), beta.to(device), nhwc(gy)
    assert lib.bn_relu_maxpool_supported(N, H, W, C)
    # fused
    stats = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.bn_stats(xd, M, C, stats)
    stats2 = stats.clone()
    y = torch.empty(N, OH, OW, C, device=device); tap = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=device)
    sm = torch.empty(C, device=device); si = torch.empty(C, device=device)
    rmd, rvd = rm0.clone().to(device), rv0.clone().to(device)
    lib.bn_relu_maxpool_fwd(xd, stats, gd, bd, N, H, W, C, 1e-5, 0.1, y, tap, sm, si, rmd, rvd)
    assert float(stats.abs().max()) == 0.0, "bn_relu_maxpool_fwd must leave the slots and arrival counters zeroed"
    # unfused kernels of the same library
    yf = torch.empty(N, H, W, C, device=device); sm2 = torch.empty(C, device=device); si2 = torch.empty(C, device=device)
    lib.bn_act_fwd(xd, stats2, gd, bd, None, 1, M, C, 1e-5, 0.1, yf, sm2, si2, None, None)
    y2 = torch.empty_like(y); tap2 = torch.empty_like(tap)
    lib.maxpool2d_fwd(yf, N, H, W, C, 3, 2, 1, y2, tap2)
    assert torch.equal(sm, sm2) and torch.equal(si, si2), "fused stem: batch statistics"
    assert torch.equal(y, y2) and torch.equal(tap, tap2), "fused stem forward == bn_act_fwd + maxpool2d_fwd (bits, winning taps)"
    ref = out.detach().permute(0, 2, 3, 1)
    assert float((y.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), "fused stem fwd vs torch"
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    # backward
    red = torch.zeros(lib.bn_stats_floats(C), device=device)
    dx = torch.full((N, H, W, C), 7.0, device=device)
    dg = torch.full((C,), 0.5, device=device); db = torch.full((C,), -0.25, device=device)
    lib.bn_relu_maxpool_bwd(gyd, tap, xd, sm, si, gd, bd, N, H, W, C, red, dx, dg, db)
    assert float(red.abs().max()) == 0.0, "bn_relu_maxpool_bwd must leave the slots and arrival counters zeroed"
    dyf = torch.empty(N, H, W, C, device=device)
    lib.maxpool2d_bwd(gyd, tap, N, H, W, C, 3, 2, 1, dyf)
    dx2 = torch.empty_like(dx); dg2 = torch.zeros(C, device=device); db2 = torch.zeros(C, device=device)
    lib.bn_act_bwd(dyf, None, xd, sm, si, gd, bd, 1, M, C, red, dx2, None, dg2, db2)
    scale = float(dx2.abs().max())
    assert float((dx - dx2).abs().max()) <= 2e-6 * scale + 1e-9, "fused stem backward vs maxpool2d_bwd + bn_act_bwd"
    assert float((dg - 0.5 - dg2).abs().max()) <= 1e-5 * float(dg2.abs().max()) + 1e-6
    assert float((db + 0.25 - db2).abs().max()) <= 1e-5 * float(db2.abs().max()) + 1e-6
    refdx = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu() - refdx).abs().max()) <= 2e-4 * float(refdx.abs().max()) + 1e-7, "fused stem dx vs torch"
    assert float((dg.cpu() - 0.5 - gr.grad).abs().max()) <= 2e-4 * float(gr.grad.abs().max()) + 1e-5
    assert float((db.cpu() + 0.25 - br.grad).abs().max()) <= 2e-4 * float(br.grad.abs().max()) + 1e-5
    # the same backward with the reduction walked over the POOLED grid (hifihr_bn_relu_maxpool_bwd_y: xhat of a window's winner recovered
    # from the pooled value; channels with |gamma| < 1e-3 -- channel 1 of this case -- gather the winner's x through the tap)
    dx3 = torch.full((N, H, W, C), 7.0, device=device)
    dg3 = torch.full((C,), 0.5, device=device); db3 = torch.full((C,), -0.25, device=device)
    lib.bn_relu_maxpool_bwd_y(gyd, y, tap, xd, sm, si, gd, bd, N, H, W, C, red, dx3, dg3, db3)
    assert float(red.abs().max()) == 0.0, "bn_relu_maxpool_bwd_y must leave the slots and arrival counters zeroed"
    assert float((dx3 - dx).abs().max()) <= 5e-6 * scale + 1e-9, "pooled-grid reduction vs the pass over x (dx)"
    assert float((dg3 - dg).abs().max()) <= 2e-5 * float(dg2.abs().max()) + 1e-6
    assert float((db3 - db).abs().max()) <= 2e-5 * float(db2.abs().max()) + 1e-6
    assert float((dx3.cpu() - refdx).abs().max()) <= 2e-4 * float(refdx.abs().max()) + 1e-7, "pooled-grid stem dx vs torch"
    assert float((dg3.cpu() - 0.5 - gr.grad).abs().max()) <= 2e-4 * float(gr.grad.abs().max()) + 1e-5


def maxpool_case(lib, device, N, H, W, C, seed=0, ties=False, ksp=(3, 2, 1)):
    """nn.MaxPool2d(k, s, p) forward / backward on NHWC."""
    k, s_, p_ = ksp
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen)
    if ties:
        x = torch.relu(x)                          # post-ReLU activations: many exact zeros tie inside a window
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, k, s_, p_)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    out = torch.empty(N, OH, OW, C, device=device); tap = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=device)
    lib.maxpool2d_fwd(xd, N, H, W, C, k, s_, p_, out, tap)
    assert torch.equal(out.cpu(), y.detach().permute(0, 2, 3, 1)), "maxpool fwd (exact)"
    dx = torch.full((N, H, W, C), 7.0, device=device)          # overwritten, not accumulated
    lib.maxpool2d_bwd(gy.permute(0, 2, 3, 1).contiguous().to(device), tap, N, H, W, C, k, s_, p_, dx)
    refdx = xr.grad.permute(0, 2, 3, 1)
    # overlapping windows (stride < kernel): several gradients are summed per input pixel, in a different order than ATen
    assert float((dx.cpu() - refdx).abs().max()) <= 1e-6 + 2e-6 * float(refdx.abs().max()), "maxpool bwd"
    # the flattening form: y as the [N, C * OH * OW] matrix of `y.view(N, -1)` (NCHW order), gy likewise -- the same bits as the form above
    if hasattr(lib, "maxpool2d_fwd_flat"):
        flat = torch.full((N, C * OH * OW), 7.0, device=device); tap2 = torch.empty_like(tap)
        lib.maxpool2d_fwd_flat(xd, N, H, W, C, k, s_, p_, flat, tap2)
        assert torch.equal(flat.cpu(), y.detach().reshape(N, -1)) and torch.equal(tap2, tap), "maxpool fwd, flattened output"
        dx2 = torch.full((N, H, W, C), 7.0, device=device)
        lib.maxpool2d_bwd_flat(gy.reshape(N, -1).contiguous().to(device), tap2, N, H, W, C, k, s_, p_, dx2)
        assert torch.equal(dx2, dx), "maxpool bwd from the flattened gradient"


# ------------------------------------------------------------------------------------------------
# fused losses (csrc/losses.hip) vs the torch-op restatement in oracle/loss_oracle.py (pinned against the reference's
# own functions by tests/golden/losses.npz and loss_dict.npz in test_host_logic.py / test_oracle_losses.py)
# ------------------------------------------------------------------------------------------------
def vertex_face_csr(faces, V):
    f = np.asarray(faces, dtype=np.int64).reshape(-1)
    order = np.argsort(f, kind="stable")
    off = np.zeros(V + 1, dtype=np.int32)
    np.add.at(off, f + 1, 1)
    return np.cumsum(off).astype(np.int32), ((order // 3) * 4 + (order % 3)).astype(np.int32)


def geom_loss_case(lib, device, B, V, F, mse, seed=0, J=21, NS=10, NP=48):
    import torch.nn.functional as Fn
    from oracle.loss_oracle import edge_length_loss
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    joints, jgt, verts, vgt = rnd(B, J, 3) * 0.05, rnd(B, J, 3) * 0.05, rnd(B, V, 3) * 0.05, rnd(B, V, 3) * 0.05
    shape, pose = rnd(B, NS), rnd(B, NP)
    faces = torch.stack([torch.randperm(V, generator=gen)[:3] for _ in range(F)]).int() if F else None
    lam = [1e4, 1e4, 1e2, 0.25, 0.5]
    base = Fn.mse_loss if mse else Fn.l1_loss
    jr, vr, sr, pr = (t.clone().requires_grad_(True) for t in (joints, verts, shape, pose))
    ref = [lam[0] * base(jr, jgt), lam[1] * base(vr, vgt),
           lam[2] * edge_length_loss(vr, vgt, faces.unsqueeze(0)) if F else torch.zeros(()),
           lam[3] * Fn.mse_loss(sr, torch.zeros_like(sr)), lam[4] * Fn.mse_loss(pr, torch.zeros_like(pr))]
    gout = torch.tensor([0.7, 1.3, 0.9, 1.1, 0.6])
    sum(g * r for g, r in zip(gout, ref)).backward()
    d = lambda t: t.to(device).contiguous() if t is not None else None
    partial = torch.empty(B * 5, device=device); out = torch.empty(5, device=device)
    args = (d(joints), d(jgt), d(verts), d(vgt), d(shape), d(pose), d(faces))
    lib.geom_loss_fwd(*args, mse, lam, partial, out)
    refv = torch.stack([r.detach() for r in ref])
    np.testing.assert_allclose(out.cpu().numpy(), refv.numpy(), rtol=2e-5, atol=1e-7)
    off, idx = vertex_face_csr(faces.numpy(), V) if F else (None, None)
    gj, gv, gs, gp = (torch.full(t.shape, 7.0, device=device) for t in (joints, verts, shape, pose))
    lib.geom_loss_bwd(*args, d(torch.from_numpy(off)) if F else None, d(torch.from_numpy(idx)) if F else None, mse, lam, d(gout), gj, gv, gs, gp)
    for got, want, name in ((gj, jr.grad, "joints"), (gv, vr.grad, "verts"), (gs, sr.grad, "shape"), (gp, pr.grad, "pose")):
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-9, f"geom loss grad {name}"


def light_split_case(lib, device, B=7, seed=0):
    """hifihr_light_split_fwd / _bwd vs torch: colors = hardtanh(lights[:, :3]), directions = lights[:, 3:] and their gradient, incl.
    values exactly at the clamp ends (gradient 0 there, as nn.Hardtanh) and absent gradients."""
    gen = torch.Generator().manual_seed(seed)
    l = (torch.randn(B, 6, generator=gen) * 1.5)
    l[0, 0] = 1.0; l[min(1, B - 1), 1] = -1.0
    l[B - 1, 2] = float("nan")                         # a NaN colour stays NaN and passes its gradient, exactly as torch's hardtanh
    l.requires_grad_(True)
    c = torch.nn.functional.hardtanh(l[:, :3]); dd = l[:, 3:]
    gc, gd = torch.randn(B, 3, generator=gen), torch.randn(B, 3, generator=gen)
    ((c * gc).sum() + (dd * gd).sum()).backward()
    ld = l.detach().to(device).contiguous()
    oc, od = torch.full((B, 3), 7.0, device=device), torch.full((B, 3), 7.0, device=device)
    lib.light_split_fwd(ld, oc, od)
    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    assert same(oc.cpu(), c.detach()) and bool(oc.cpu()[B - 1, 2].isnan()) and torch.equal(od.cpu(), dd.detach().contiguous())
    gl = torch.full((B, 6), 7.0, device=device)
    lib.light_split_bwd(ld, gc.to(device), gd.to(device), gl)
    assert same(gl.cpu(), l.grad) and float(gl.cpu()[B - 1, 2]) == float(gc[B - 1, 2])
    lib.light_split_bwd(ld, None, gd.to(device), gl)
    assert float(gl.cpu()[:, :3].abs().max()) == 0.0 and torch.equal(gl.cpu()[:, 3:], gd)
    lib.light_split_bwd(ld, gc.to(device), None, gl)
    assert float(gl.cpu()[:, 3:].abs().max()) == 0.0


def loss_total_case(lib, device, seed=0):
    """hifihr_loss_total_fwd / _bwd: the sum of the leading entries of up to four small vectors, and its gradient (reference
    train_hrnet.py:98-104: loss = sum of the selected loss_dic entries)."""
    gen = torch.Generator().manual_seed(seed)
    parts = [torch.randn(5, generator=gen), torch.randn(4, generator=gen), torch.randn(1, generator=gen)]
    counts = [5, 3, 1]
    d = [p.to(device) for p in parts]
    total = torch.full((), 7.0, device=device)
    lib.loss_total_fwd(d, counts, total)
    want = sum(float(p[:n].double().sum()) for p, n in zip(parts, counts))
    assert abs(float(total) - want) <= 1e-6 * max(1.0, abs(want))
    g = torch.tensor(1.75, device=device)
    grads = [torch.full_like(p, 9.0) for p in d]
    lib.loss_total_bwd(g, grads, counts)
    for gr, n in zip(grads, counts):
        assert torch.equal(gr.cpu()[:n], torch.full((n,), 1.75)) and float(gr.cpu()[n:].abs().sum()) == 0.0
    one = torch.randn(1, generator=gen).to(device)
    lib.loss_total_fwd([one], [1], total)
    assert float(total) == float(one)


def photo_loss_case(lib, device, B, H, W, seed=0, with_g=True):
    import torch.nn.functional as Fn
    gen = torch.Generator().manual_seed(seed)
    rgba = torch.rand(B, 4, H, W, generator=gen)
    rgba[:, 3] = torch.where(torch.rand(B, H, W, generator=gen) > 0.5, rgba[:, 3], torch.zeros(B, H, W))     # holes: alpha == 0
    imgs = torch.rand(B, 3, H, W, generator=gen)
    seg = (torch.rand(B, H, W, generator=gen) > 0.4).long()
    l_tex, l_mrgb, l_sil = 0.005, 0.005, 0.1
    rr = rgba.clone().requires_grad_(True)
    re_sil = rr[:, 3:4].detach()
    re_sil = torch.where(re_sil > 0, torch.full_like(re_sil, 255.0), re_sil)
    segf = seg.unsqueeze(1).float()
    mask_rgbs = segf * imgs
    re_img = rr[:, :3] * (re_sil / 255.0)
    tex = l_tex * Fn.l1_loss(re_img, mask_rgbs)
    mrgb = l_mrgb * Fn.mse_loss(torch.mean(mask_rgbs), torch.mean(re_img))
    sil = l_sil * Fn.l1_loss(re_sil, segf)
    g_re = torch.randn(B, 3, H, W, generator=gen) * 1e-6 if with_g else None
    gout = torch.tensor([0.8, 1.7, 0.0, 0.0])
    tot = gout[0] * tex + gout[1] * mrgb
    if with_g:
        tot = tot + (re_img * g_re).sum()
    tot.backward()
    d = lambda t: t.to(device).contiguous() if t is not None else None
    rd, idd, sd = d(rgba), d(imgs), d(seg)
    re_m = torch.empty(B, 3, H, W, device=device); mk = torch.empty_like(re_m)
    partial = torch.empty(lib.photo_loss_partial_floats(), device=device); out = torch.empty(4, device=device)
    lib.photo_loss_fwd(rd, idd, sd, l_tex, l_mrgb, l_sil, re_m, mk, partial, out)
    assert float((re_m.cpu() - re_img.detach()).abs().max()) <= 1e-6 and torch.equal(mk.cpu(), mask_rgbs)
    want = torch.stack([tex.detach(), mrgb.detach(), sil, (re_img.mean() - mask_rgbs.mean()).detach()])
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=3e-5, atol=1e-9)
    grad = torch.full((B, 4, H, W), 7.0, device=device)
    lib.photo_loss_bwd(rd, re_m, mk, d(g_re), d(gout), out, l_tex, l_mrgb, grad)
    assert float((grad.cpu() - rr.grad).abs().max()) <= 2e-5 * float(rr.grad.abs().max()) + 1e-12, "photo loss grad"
    rs = torch.empty(B, 1, H, W, device=device); mrgbs = torch.empty(B, 3, H, W, device=device)
    lib.sil_post(rd, idd, rs, mrgbs)
    assert torch.equal(rs.cpu(), re_sil) and torch.equal(mrgbs.cpu(), imgs * (re_sil > 0).float())


# ------------------------------------------------------------------------------------------------
# small-batch fully connected layer (csrc/mlp.hip) vs nn.Linear (+ BatchNorm1d training mode) (+ ReLU)
# ------------------------------------------------------------------------------------------------
def linear_case(lib, device, B, I, O, act, bn, seed=0, need_dx=True):
    import torch.nn.functional as Fn
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    x, w, b = rnd(B, I), rnd(O, I) / I ** 0.5, rnd(O) * 0.1
    gamma, beta = 1 + 0.2 * rnd(O), 0.1 * rnd(O)
    rm0, rv0 = 0.1 * rnd(O), 1 + 0.1 * torch.rand(O, generator=gen)
    xr, wr, br, gr, ber = (t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta))
    rm, rv = rm0.clone(), rv0.clone()
    out = Fn.linear(xr, wr, br)
    if bn:
        out = Fn.batch_norm(out, rm, rv, gr, ber, training=True, momentum=0.1, eps=1e-5)
    if act:
        out = Fn.relu(out)
    gy = rnd(B, O)
    out.backward(gy)
    d = lambda t: t.to(device).contiguous()
    xd, wd, bd = d(x), d(w), d(b)
    y = torch.empty(B, O, device=device)
    bnf = bnb = None
    if bn:
        z, sm, si = torch.empty(B, O, device=device), torch.empty(O, device=device), torch.empty(O, device=device)
        rmd, rvd = d(rm0), d(rv0)
        bnf = (d(gamma), d(beta), 1e-5, 0.1, rmd, rvd, z, sm, si)
    lib.linear_fwd(xd, wd, bd, act, y, bnf)
    tol = 3e-5 * max(1.0, float(out.detach().abs().max()))
    assert float((y.cpu() - out.detach()).abs().max()) <= tol, "linear fwd"
    if bn:
        np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    dz = torch.empty(B, O, device=device)
    dW = torch.full((O, I), 0.5, device=device); db = torch.full((O,), -0.25, device=device)       # accumulate semantics
    dx = torch.full((B, I), 7.0, device=device) if need_dx else None                               # overwritten
    if bn:
        dg, dbt = torch.full((O,), 0.125, device=device), torch.full((O,), 2.0, device=device)
        bnb = (bnf[0], z, sm, si, dg, dbt)
    lib.linear_bwd(d(gy), y if act else None, xd, wd, act, dz, dW, db, dx, bnb)
    rel = lambda got, want, name, t=2e-4: (float((got.cpu() - want).abs().max()) <= t * float(want.abs().max()) + 1e-6) or \
        (_ for _ in ()).throw(AssertionError(f"linear {name}: {float((got.cpu() - want).abs().max())} vs {float(want.abs().max())}"))
    rel(dW - 0.5, wr.grad, "dW")
    if bn:       # the true bias gradient under batch-norm is zero (the batch mean removes it): absolute check
        assert float((db.cpu() + 0.25).abs().max()) <= 1e-4, "linear db (batch-norm: ~0)"
    else:
        rel(db + 0.25, br.grad, "db")
    if need_dx:
        rel(dx, xr.grad, "dx")
    if bn:
        rel(dg - 0.125, gr.grad, "dgamma")
        rel(dbt - 2.0, ber.grad, "dbeta")


def conv_relu_nobias_case(lib, device, N, H, W, C, K, R, stride, seed=0, pad=0):
    """hifihr_conv2d_fwd(bias = NULL, act = 1): the output is clamped whatever kernel the shape dispatches to (the 1x1 / stride-1
    shapes with K % 128 == 0 once took the GEMM kernels, which have no activation epilogue: round-2 advisor finding)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen); w = torch.randn(K, C, R, R, generator=gen) / (C * R * R) ** 0.5
    ref = F.relu(F.conv2d(x, w, None, stride=stride, padding=pad)).permute(0, 2, 3, 1)
    d = lambda t: t.to(device).contiguous()
    out = torch.full(tuple(ref.shape), -3.0, device=device)
    lib.conv2d_fwd(d(x.permute(0, 2, 3, 1)), d(w.permute(0, 2, 3, 1)), None, out, N, H, W, C, K, R, R, stride, pad, act=1)
    assert float(out.min()) >= 0.0, "ReLU epilogue dropped"
    assert float((out.cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-6, "conv+relu fwd"


def conv_wino2_case(lib, device, N, H, W, seed=0, bias_relu=False, rtol=2e-5):
    """hifihr_conv3x3_c64_wino (conv_wino2_kernel: 64 -> 64, 3x3 / stride 1 / pad 1 as Winograd F(2x2, 3x3) with the transforms in registers)
    vs F.conv2d: forward (+ bias + ReLU, + batch-norm statistics slots) and backward-data through U' of the transposed, rotated filter."""
    import torch.nn.functional as F
    assert lib.conv3x3_c64_wino_supported(N, H, W, 64, 64)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 64, H, W, generator=gen) + 0.5
    w = torch.randn(64, 64, 3, 3, generator=gen) / 24.0
    b = torch.randn(64, generator=gen) * 0.3 if bias_relu else None
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, w, b, padding=1)
    y = F.relu(z) if bias_relu else z
    gy = torch.randn(z.shape, generator=gen)
    z.backward(gy)
    d = lambda t: t.to(device).contiguous()
    w_krsc = d(w.permute(0, 2, 3, 1))
    U = torch.empty(16 * 64 * 64, device=device)
    lib.wino_weight_transform(w_krsc, U, 64, 64, 0)
    out = torch.full((N, H, W, 64), 7.0, device=device)
    stats = None if bias_relu else torch.zeros(lib.bn_stats_floats(64), device=device)
    lib.conv3x3_c64_wino(d(x.permute(0, 2, 3, 1)), U, d(b) if bias_relu else None, bias_relu, out, stats, N, H, W)
    ref = y.detach().permute(0, 2, 3, 1)
    err = float((out.cpu() - ref).abs().max())
    assert err <= rtol * float(ref.abs().max()) + 1e-6, f"wino2 conv fwd: {err}"
    if stats is not None:
        sl = bn_slots(stats.cpu(), 64).sum(0)
        r2 = ref.double().reshape(-1, 64)
        assert float((sl[0] - r2.sum(0)).abs().max()) <= 2e-5 * float(r2.abs().sum(0).max()), "wino2 conv: channel sums"
        assert float((sl[1] - (r2 * r2).sum(0)).abs().max()) <= 2e-5 * float((r2 * r2).sum(0).max()), "wino2 conv: channel sums of squares"
    if not bias_relu:
        # backward-data: U' from the [C][R][S][K] transpose with flip (what hifihr_weight_prep kind 2 produces from w directly)
        wt = d(w.permute(1, 2, 3, 0))
        U2 = torch.empty(16 * 64 * 64, device=device)
        lib.wino_weight_transform(wt, U2, 64, 64, 1)
        dx = torch.full((N, H, W, 64), 7.0, device=device)
        lib.conv3x3_c64_wino(d(gy.permute(0, 2, 3, 1)), U2, None, False, dx, None, N, H, W)
        refx = xr.grad.permute(0, 2, 3, 1)
        err = float((dx.cpu() - refx).abs().max())
        assert err <= rtol * float(refx.abs().max()) + 1e-6, f"wino2 conv bwd data: {err}"


def conv_c64_bwd_pair_case(lib, device, N, H, W, seed=0, with_res=False):
    """hifihr_conv3x3_c64_bwd_pair (conv_c64_bwd_pair_kernel: data gradient + weight gradient of a 64 -> 64 3x3 layer in one launch) vs the two
    separate calls it replaces: dx bit for bit (every output element is computed the same way whatever the workgroup's share), dw within
    the summation-order difference of the slab split and equal to itself on a second launch; both vs F.conv2d's gradients."""
    import torch.nn.functional as F
    assert lib.conv3x3_c64_bwd_pair_supported(N, H, W)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 64, H, W, generator=gen) + 0.5
    w = torch.randn(64, 64, 3, 3, generator=gen) / 24.0
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, None, padding=1)
    gy = torch.randn(z.shape, generator=gen)
    z.backward(gy)
    res = torch.randn(N, H, W, 64, generator=gen) if with_res else None
    d = lambda t: t.to(device).contiguous()
    U2 = torch.empty(16 * 64 * 64, device=device)
    lib.wino_weight_transform(d(w.permute(1, 2, 3, 0)), U2, 64, 64, 1)
    x_d, gy_d = d(x.permute(0, 2, 3, 1)), d(gy.permute(0, 2, 3, 1))
    res_d = d(res) if with_res else None
    dx0 = torch.full((N, H, W, 64), 7.0, device=device)
    if with_res:
        lib.conv3x3_c64_wino_res(gy_d, U2, res_d, dx0, N, H, W)
    else:
        lib.conv3x3_c64_wino(gy_d, U2, None, False, dx0, None, N, H, W)
    dw0 = torch.full((64, 3, 3, 64), 0.25, device=device)
    lib.conv2d_bwd_weight(x_d, gy_d, dw0, N, H, W, 64, 64, 3, 3, 1, 1)
    outs = []
    for _ in range(2):
        dx = torch.full((N, H, W, 64), 7.0, device=device); dw = torch.full((64, 3, 3, 64), 0.25, device=device)
        lib.conv3x3_c64_bwd_pair(gy_d, U2, res_d, dx, x_d, dw, N, H, W)
        outs.append((dx, dw))
    assert torch.equal(outs[0][0], dx0), "pair: dx differs from the separate launch"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "pair: not bit-reproducible"
    refw = wr.grad.permute(0, 2, 3, 1)
    err = float((outs[0][1].cpu() - 0.25 - refw).abs().max())
    assert err <= 2e-4 * float(refw.abs().max()) + 1e-6, f"pair: dw vs torch {err}"
    err0 = float((outs[0][1] - dw0).abs().max())
    assert err0 <= 2e-5 * float(refw.abs().max()) + 1e-6, f"pair: dw vs the separate launch {err0}"
    refx = xr.grad.permute(0, 2, 3, 1) + (res if with_res else 0.0)
    err = float((outs[0][0].cpu() - refx).abs().max())
    assert err <= 2e-5 * float(refx.abs().max()) + 1e-6, f"pair: dx vs torch {err}"
    # the same launch with the slab sum left to hifihr_conv_halo_wgrad_reduce_multi (the step's deferred form): two "layers" in one reduce launch,
    # each bit-identical to the sum the pair call makes itself
    nb = lib.conv2d_wgrad_workspace_bytes(N, H, W, 64, 64, 3, 3, 1, 1)
    jobs, dxs = [], []
    for rep in range(2):
        slabs = torch.full(((nb + 3) // 4,), float("nan"), device=device)
        dx = torch.full((N, H, W, 64), 7.0, device=device); dw = torch.full((64, 3, 3, 64), 0.25, device=device)
        ns = lib.conv3x3_c64_bwd_pair_slabs(gy_d, U2, res_d, dx, x_d, slabs, N, H, W)
        assert ns > 0
        jobs.append((slabs, ns, dw)); dxs.append(dx)
    lib.conv_halo_wgrad_reduce_multi(jobs)
    for (slabs, ns, dw), dx in zip(jobs, dxs):
        assert torch.equal(dx, outs[0][0]) and torch.equal(dw, outs[0][1]), "deferred slab sum differs from the pair call's own"


def conv_bias_relu_case(lib, device, N, H, W, C, K, R, stride, seed=0, pad=0):
    """conv + bias + ReLU in one launch (act = 1) and its backward prologue bias_relu_bwd, vs torch."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen); w = torch.randn(K, C, R, R, generator=gen) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=gen) * 0.3
    br = b.clone().requires_grad_(True)
    z = F.conv2d(x, w, br, stride=stride, padding=pad)
    z.retain_grad()
    y = F.relu(z)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    d = lambda t: t.to(device).contiguous()
    out = torch.empty(N, OH, OW, K, device=device)
    lib.conv2d_fwd(d(x.permute(0, 2, 3, 1)), d(w.permute(0, 2, 3, 1)), d(b), out, N, H, W, C, K, R, R, stride, pad, act=1)
    ref = y.detach().permute(0, 2, 3, 1)
    assert float((out.cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-6, "conv+bias+relu fwd"
    g = torch.full_like(out, 7.0); db = torch.full((K,), 0.5, device=device)
    lib.bias_relu_bwd(d(gy.permute(0, 2, 3, 1)), out, N * OH * OW, K, g, db)
    # the mask comes from OUR y (elements within rounding of 0 may differ from torch's): compare where |z| is not tiny
    zr = z.detach().permute(0, 2, 3, 1)
    safe = zr.abs() > 1e-5
    refg = z.grad.permute(0, 2, 3, 1)
    assert float(((g.cpu() - refg) * safe).abs().max()) <= 1e-6, "masked gradient"
    assert float((db.cpu() - 0.5 - br.grad).abs().max()) <= 1e-4 * float(br.grad.abs().max()) + 1e-5, "bias gradient (accumulates)"


# ------------------------------------------------------------------------------------------------
# squeeze-and-excitation (csrc/se.hip + the linear kernels with swish / sigmoid epilogues) vs torch autograd
# ------------------------------------------------------------------------------------------------
def se_case(lib, device, B, H, W, C, SQ, seed=0):
    import torch.nn.functional as Fn
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    x = rnd(B, C, H, W); w1 = rnd(SQ, C) / C ** 0.5; b1 = rnd(SQ) * 0.1; w2 = rnd(C, SQ) / SQ ** 0.5; b2 = rnd(C) * 0.1
    xr, w1r, b1r, w2r, b2r = (t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    m = xr.mean((2, 3))
    z1 = Fn.linear(m, w1r, b1r)
    h1 = z1 * torch.sigmoid(z1)
    gate = torch.sigmoid(Fn.linear(h1, w2r, b2r))
    y = xr * gate[:, :, None, None]
    gy = rnd(B, C, H, W)
    y.backward(gy)
    d = lambda t: t.to(device).contiguous()
    HW = H * W
    xd = d(x.permute(0, 2, 3, 1)); gyd = d(gy.permute(0, 2, 3, 1))
    mean = torch.zeros(B, C, device=device)
    lib.se_pool(xd, B, HW, C, mean)
    assert float((mean.cpu() - m.detach()).abs().max()) <= 1e-5, "se pool"
    h1d, z1d, gd = torch.empty(B, SQ, device=device), torch.empty(B, SQ, device=device), torch.empty(B, C, device=device)
    w1d, b1d, w2d, b2d = d(w1), d(b1), d(w2), d(b2)
    lib.linear_fwd(mean, w1d, b1d, 2, h1d, z=z1d)
    lib.linear_fwd(h1d, w2d, b2d, 3, gd)
    assert float((gd.cpu() - gate.detach()).abs().max()) <= 2e-5, "se gate"
    yd = torch.empty(B, H, W, C, device=device)
    lib.se_scale(xd, gd, None, 0.0, B, HW, C, yd)
    ref = y.detach().permute(0, 2, 3, 1)
    assert float((yd.cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()), "se fwd"
    dgate = torch.zeros(B, C, device=device)
    lib.se_bwd_gate(gyd, xd, B, HW, C, dgate)
    dw1, db1, dw2, db2 = (torch.zeros_like(t) for t in (w1d, b1d, w2d, b2d))
    dh1, dmean = torch.empty(B, SQ, device=device), torch.empty(B, C, device=device)
    dz2, dz1 = torch.empty(B, C, device=device), torch.empty(B, SQ, device=device)
    lib.linear_bwd(dgate, gd, h1d, w2d, 3, dz2, dw2, db2, dh1)
    lib.linear_bwd(dh1, None, mean, w1d, 2, dz1, dw1, db1, dmean, z=z1d)
    dx = torch.empty(B, H, W, C, device=device)
    lib.se_scale(gyd, gd, dmean, 1.0 / HW, B, HW, C, dx)
    def rel(got, want, name, t=3e-4):
        err, mag = float((got.cpu() - want).abs().max()), float(want.abs().max())
        assert err <= t * mag + 1e-7, f"se {name}: {err} vs {mag}"
    rel(dx, xr.grad.permute(0, 2, 3, 1), "dx")
    rel(dw1, w1r.grad, "dw1"); rel(db1, b1r.grad, "db1"); rel(dw2, w2r.grad, "dw2"); rel(db2, b2r.grad, "db2")
    # ---- the two layers fused (se_mlp_fwd / se_mlp_bwd): same numbers, accumulators handed back zeroed, gradients ACCUMULATED
    assert lib.se_mlp_supported(C, SQ)
    acc = torch.zeros(B, C, device=device)
    lib.se_pool(xd, B, HW, C, acc)
    w2t = d(w2.t())
    mean2, z2, h2, g2 = torch.empty(B, C, device=device), torch.empty(B, SQ, device=device), torch.empty(B, SQ, device=device), torch.empty(B, C, device=device)
    lib.se_mlp_fwd(acc, w1d, b1d, w2t, b2d, B, C, SQ, mean2, z2, h2, g2)
    assert float(acc.abs().max()) == 0.0, "se_mlp_fwd hands the accumulator back zeroed"
    assert float((mean2.cpu() - m.detach()).abs().max()) <= 1e-5 and float((g2.cpu() - gate.detach()).abs().max()) <= 2e-5, "fused se gate"
    assert float((z2.cpu() - z1.detach()).abs().max()) <= 2e-5 and float((h2.cpu() - h1.detach()).abs().max()) <= 2e-5
    dacc = torch.zeros(B, C, device=device)
    lib.se_bwd_gate(gyd, xd, B, HW, C, dacc)
    pre = 0.25                                                    # the gradient buffers already hold something: += semantics
    fw1, fb1, fw2, fb2 = (torch.full_like(t, pre) for t in (w1d, b1d, w2d, b2d))
    dz2f, dz1f, dmeanf = torch.empty(B, C, device=device), torch.empty(B, SQ, device=device), torch.empty(B, C, device=device)
    lib.se_mlp_bwd(dacc, g2, z2, h2, mean2, w1d, w2t, B, C, SQ, dz2f, dz1f, dmeanf, fw1, fb1, fw2, fb2)
    assert float(dacc.abs().max()) == 0.0, "se_mlp_bwd hands the accumulator back zeroed"
    dx2 = torch.empty(B, H, W, C, device=device)
    lib.se_scale(gyd, g2, dmeanf, 1.0 / HW, B, HW, C, dx2)
    rel(dx2, xr.grad.permute(0, 2, 3, 1), "fused dx")
    rel(fw1 - pre, w1r.grad, "fused dw1"); rel(fb1 - pre, b1r.grad, "fused db1"); rel(fw2 - pre, w2r.grad, "fused dw2"); rel(fb2 - pre, b2r.grad, "fused db2")


# ------------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) path (csrc/wino.hip + the batched MFMA GEMM) vs torch conv2d, forward and backward-data
# ------------------------------------------------------------------------------------------------
def wino4_dw_multi_case(lib, device, seed=0):
    """hifihr_wino4_dw_transform_multi: the F(4x4) weight-gradient transforms of several layers in one launch == one
    hifihr_wino_dw_transform_parts_m(..., 4) call per layer, bit for bit (same per-item arithmetic, slabs in slab order) -- wide-form layers
    (<= 8 192 items), item-per-thread layers, 1 .. 7 slabs, a layer whose item count is not a multiple of the workgroup's, accumulation
    into non-zero targets; more jobs than one launch's argument block holds (24)."""
    gen = torch.Generator().manual_seed(seed)
    shapes = [(64, 64, 3), (128, 128, 7), (256, 128, 1), (72, 100, 2), (264, 128, 2)] + [(16, 8 + 4 * i, 1 + i % 3) for i in range(22)]
    jobs, want = [], []
    for K, C, parts in shapes:
        dU = torch.randn(parts * 36 * K * C, generator=gen).to(device)
        base = torch.randn(K * 9 * C, generator=gen).to(device)
        single = base.clone()
        lib.wino_dw_transform_parts(dU, parts, single, K, C, 4)
        tgt = base.clone()
        jobs.append((dU, parts, tgt, K, C)); want.append(single)
    lib.wino4_dw_transform_multi(jobs)
    for (dU, parts, tgt, K, C), single in zip(jobs, want):
        assert torch.equal(tgt, single), (K, C, parts, float((tgt - single).abs().max()))


def wino_case(lib, device, N, H, W, C, K, seed=0, with_stats=True, use_ws=True, m=2):
    """m = 2: F(2x2, 3x3) (csrc/wino.hip, 16 positions); m = 4: F(4x4, 3x3) (csrc/wino4.hip, 36 positions, slab backward-weight only)."""
    P = (m + 2) ** 2
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=gen); w = torch.randn(K, C, 3, 3, generator=gen) / (9 * C) ** 0.5
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, 1, 1)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    d = lambda t: t.to(device).contiguous()
    T = lib.wino_tiles(N, H, W, m)               # (the library's count: 16 images share a mosaic of tiles at m = 4, H % 4 in {1, 2})
    xd, wd, gyd = d(x.permute(0, 2, 3, 1)), d(w.permute(0, 2, 3, 1)), d(gy.permute(0, 2, 3, 1))
    nb = max(lib.wino_gemm_workspace_bytes(N, H, W, C, K, m), lib.wino_gemm_workspace_bytes(N, H, W, K, C, m)) if use_ws else 0
    ws = torch.zeros(nb // 4, device=device) if nb else None
    # forward
    U = torch.empty(P, K, C, device=device); V = torch.empty(P, T, C, device=device); M = torch.empty(P, T, K, device=device)
    out = torch.full((N, H, W, K), 7.0, device=device)
    stats = torch.zeros(lib.bn_stats_floats(K), device=device) if with_stats else None
    lib.wino_weight_transform(wd, U, K, C, 0, m)
    lib.wino_input_transform(xd, V, N, H, W, C, m)
    lib.wino_gemm(V, U, M, N, H, W, C, K, ws=ws, m=m)
    lib.wino_output_transform(M, out, stats, N, H, W, K, m=m)
    ref = y.detach().permute(0, 2, 3, 1)
    err = float((out.cpu() - ref).abs().max())
    # F(4x4, 3x3) in fp32: 8-9e-6 of max |y| typical (profiles/r02_wino43_error.txt), 2.0e-5 at 512 x 512 channels (4 608-term sums;
    # flake check of round 5): the bound leaves a factor 2.5 over the worst shape.  F(2x2): ~1e-6.
    wtol = 5e-5 if m == 4 else 3e-5
    assert err <= wtol * float(ref.abs().max()) + 1e-6, f"winograd fwd: {err} vs {float(ref.abs().max())}"
    if with_stats:
        st = bn_slots(stats, K).sum(0).cpu(); flat = ref.reshape(-1, K)
        np.testing.assert_allclose(st[0].numpy(), flat.sum(0).numpy(), rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(st[1].numpy(), (flat ** 2).sum(0).numpy(), rtol=1e-4, atol=2e-3)
    # bias (+ ReLU) epilogue of the output transform (VGG19 layers): same M
    bias = torch.randn(K, generator=gen) * 0.3
    for act in (0, 1):
        out2 = torch.full((N, H, W, K), 7.0, device=device)
        lib.wino_output_transform(M, out2, None, N, H, W, K, bias=d(bias), act=act, m=m)
        ref2 = ref + bias
        ref2 = F.relu(ref2) if act else ref2
        assert float((out2.cpu() - ref2).abs().max()) <= wtol * float(ref.abs().max()) + 1e-6, f"winograd bias/act epilogue (act={act})"
    # backward-data: the same pipeline on dy with the transposed, rotated filter
    wt = torch.empty(C, 3, 3, K, device=device)
    lib.weight_transpose(wd, wt, K, 9, C)
    U2 = torch.empty(P, C, K, device=device); V2 = torch.empty(P, T, K, device=device); M2 = torch.empty(P, T, C, device=device)
    dx = torch.full((N, H, W, C), 7.0, device=device)
    lib.wino_weight_transform(wt, U2, C, K, 1, m)
    lib.wino_input_transform(gyd, V2, N, H, W, K, m)
    # one read of dy for both backward transforms == the two separate kernels, bit for bit
    V2b = torch.full_like(V2, 7.0); Ytb = torch.full((P, T, K), 7.0, device=device); Yt_ref = torch.empty(P, T, K, device=device)
    lib.wino_input_dy_transform(gyd, V2b, Ytb, N, H, W, K, m)
    lib.wino_dy_transform(gyd, Yt_ref, N, H, W, K, m)
    assert torch.equal(V2b, V2) and torch.equal(Ytb, Yt_ref), "dual dy transform"
    lib.wino_gemm(V2, U2, M2, N, H, W, K, C, ws=ws, m=m)
    lib.wino_output_transform(M2, dx, None, N, H, W, C, m=m)
    refx = xr.grad.permute(0, 2, 3, 1)
    err = float((dx.cpu() - refx).abs().max())
    assert err <= wtol * float(refx.abs().max()) + 1e-6, f"winograd bwd data: {err} vs {float(refx.abs().max())}"
    assert ws is None or float(ws.abs().max()) == 0.0
    # backward-weight: dU = sum over tiles of (A dy A^T) . (B^T d B), then dw += G^T dU G
    lib.wino_input_transform(xd, V, N, H, W, C, m)
    Yt = torch.empty(P, T, K, device=device); dU = torch.zeros(P, K, C, device=device)
    lib.wino_dy_transform(gyd, Yt, N, H, W, K, m)
    refw = wr.grad.permute(0, 2, 3, 1)
    if m == 2:                                                           # the atomics form exists for F(2x2, 3x3) only
        lib.wino_wgrad_gemm(V, Yt, dU, N, H, W, C, K)
        dw = torch.full((K, 3, 3, C), 0.5, device=device)                # accumulate semantics
        lib.wino_dw_transform(dU, dw, K, C)
        assert float(dU.abs().max()) == 0.0, "the dU accumulator must come back zeroed"
        err = float((dw.cpu() - 0.5 - refw).abs().max())
        assert err <= 1e-4 * float(refw.abs().max()) + 1e-6, f"winograd bwd weight: {err} vs {float(refw.abs().max())}"
    # the same reduction as slabs on csrc/gemm.hip (no atomics, nothing zero-initialised), summed by the slab form of the transform
    parts = lib.wino_wgrad_parts(N, H, W, C, K, m)
    assert m == 2 or parts > 0
    if parts > 0:
        dUp = torch.full((parts, P, K, C), 7.0, device=device)
        lib.wino_wgrad_gemm_parts(V, Yt, dUp, N, H, W, C, K, parts, m)
        dw2 = torch.full((K, 3, 3, C), 0.5, device=device)
        lib.wino_dw_transform_parts(dUp, parts, dw2, K, C, m)
        err = float((dw2.cpu() - 0.5 - refw).abs().max())
        assert err <= 1e-4 * float(refw.abs().max()) + 1e-6, f"winograd bwd weight (slabs): {err} vs {float(refw.abs().max())}"
        if m == 4 and C % 64 == 0 and K % 64 == 0:
            # the two backward products in ONE launch (hifihr_wino4_bwd_gemm_pair): the same bits as the two separate launches
            M2p = torch.full_like(M2, 7.0); dUq = torch.full_like(dUp, 7.0)
            lib.wino4_bwd_gemm_pair(V2, U2, M2p, V, Yt, dUq, N, H, W, C, K, parts)
            M2s = torch.full_like(M2, 7.0)
            lib.wino_gemm(V2, U2, M2s, N, H, W, K, C, ws=None, m=m)            # (same kernel family as the pair: no balanced-schedule workspace)
            Tr = lib.wino_tiles_computed(N, H, W, m)                            # (rows behind the last mosaic tile are padding: never written)
            assert torch.equal(M2p[:, :Tr], M2s[:, :Tr]), "pair launch: backward-data product"
            assert torch.equal(dUq, dUp), "pair launch: backward-weight slabs"
    return 0 if ws is None else 1


def bgemm_case(lib, device, M, N, K, batch, seed=0):
    """csrc/gemm.hip through the C ABI vs torch matmul (fp64 reference): NT (c = a b^T) and TN (slabs of a^T b)."""
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(batch, M, K, generator=gen); b = torch.randn(batch, N, K, generator=gen)
    c = torch.full((batch, M, N), 7.0, device=device)
    nb = lib.bgemm_nt_workspace_bytes(M, N, K, batch)          # > 0: the persistent, balanced kernel (zero-initialised, self-cleaning)
    ws = torch.zeros(nb // 4, device=device) if nb else None
    ad, bd = a.to(device).contiguous(), b.to(device).contiguous()
    ref = torch.matmul(a.double(), b.double().transpose(1, 2))
    for rep in range(2 if nb else 1):                         # twice on the same workspace: it must come back clean
        c.fill_(7.0)
        lib.bgemm_nt(ad, bd, c, M, N, K, batch, ws=ws)
        err = float((c.cpu().double() - ref).abs().max())
        assert err <= 2e-6 * K ** 0.5 * float(ref.abs().max()) + 1e-6, f"bgemm_nt {M}x{N}x{K}x{batch} (rep {rep}, ws {nb}): {err}"
        assert ws is None or float(ws.abs().max()) == 0.0, "workspace not handed back clean"
    return nb


def bgemm_tn_case(lib, device, M, N, T, batch, seed=0):
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(batch, T, M, generator=gen); b = torch.randn(batch, T, N, generator=gen)
    parts = lib.bgemm_tn_parts(M, N, T, batch)
    assert parts >= 1
    cp = torch.full((parts, batch, M, N), 7.0, device=device)
    lib.bgemm_tn(a.to(device).contiguous(), b.to(device).contiguous(), cp, M, N, T, batch, parts)
    ref = torch.matmul(a.double().transpose(1, 2), b.double())
    err = float((cp.cpu().double().sum(0) - ref).abs().max())
    assert err <= 2e-6 * T ** 0.5 * float(ref.abs().max()) + 1e-6, f"bgemm_tn {M}x{N}x{T}x{batch} ({parts} parts): {err}"
    return parts


# ------------------------------------------------------------------------------------------------
# evaluation: batched Procrustes-with-scale alignment (csrc/eval.hip) vs the reference's align_w_scale vectors
# ------------------------------------------------------------------------------------------------
def procrustes_case(lib, device, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "eval.npz"))
    for pr, gt, al, key in ((g["pr_j"], g["gt_j"], g["al_j"], "mpjpe"), (g["pr_v"], g["gt_v"], g["al_v"], "mpvpe")):
        pred = torch.from_numpy(pr).float().to(device).contiguous(); gtt = torch.from_numpy(gt).float().to(device).contiguous()
        aligned = torch.full_like(pred, 7.0); err = torch.full((pred.shape[0],), 7.0, device=device)
        lib.procrustes_error(pred, gtt, aligned, err)
        np.testing.assert_allclose(aligned.cpu().numpy(), al, atol=2e-7, rtol=0)     # metres; fp32 storage of ~0.1 m values
        got = float(err.sum()) / (pred.shape[0] * pred.shape[1])
        assert abs(got - float(g[key])) <= 1e-6 * float(g[key]), (key, got, float(g[key]))
        err2 = torch.full((pred.shape[0],), 7.0, device=device)
        lib.procrustes_error(pred, gtt, None, err2)                                  # error only
        assert torch.equal(err, err2)
    # properties: a similarity transform (incl. a reflection) of the ground truth aligns back exactly; coplanar input is finite
    gen = torch.Generator().manual_seed(3)
    gt = torch.randn(4, 50, 3, generator=gen) * 0.05
    q, _ = torch.linalg.qr(torch.randn(4, 3, 3, generator=gen))
    pred = 1.7 * gt @ q.transpose(1, 2) + torch.randn(4, 1, 3, generator=gen)
    flat = gt.clone(); flat[3, :, 2] = 0.0; pflat = pred.clone(); pflat[3] = flat[3] * 2.0 + 0.3
    for p_, g_ in ((pred, gt), (pflat, flat)):
        aligned = torch.empty(4, 50, 3, device=device); err = torch.empty(4, device=device)
        lib.procrustes_error(p_.to(device).contiguous(), g_.to(device).contiguous(), aligned, err)
        assert float((aligned.cpu() - g_).abs().max()) <= 2e-6 and float(err.max()) <= 50 * 2e-6


# ------------------------------------------------------------------------------------------------
# FreiHAND augmentation warp (csrc/augment.hip) vs the reference's PIL path (tests/golden/data_path.npz)
# ------------------------------------------------------------------------------------------------
def augment_case(lib, device, golden_dir):
    import os
    from hifihr_amd.data import affine_for_rotation, pil_affine_fixed_terms
    g = np.load(os.path.join(golden_dir, "data_path.npz"))
    for i in range(int(g["n"])):
        img, mask, rot = g[f"img{i}"], g[f"mask{i}"], float(g[f"rot{i}"])
        res = img.shape[0]
        total, post = affine_for_rotation(np.asarray([res // 2, res // 2]), res, [res, res], rot)
        assert np.array_equal(total, g[f"aff{i}"]) and np.array_equal(post, g[f"post{i}"]), "get_affine_transform restatement"
        # a cache of three images with the wanted one in the middle: the gather index is exercised too
        rgbx = np.zeros((3, res, res, 4), np.uint8); rgbx[1, :, :, :3] = img; rgbx[0] = 9; rgbx[2] = 17
        mk = np.zeros((3, res, res), np.uint8); mk[1] = mask; mk[0] = 255
        cache = torch.from_numpy(rgbx).to(device).view(torch.int32).reshape(3, res, res)
        out_i = torch.full((2, 3, res, res), 7.0, device=device); out_m = torch.full((2, 3, res, res), 7.0, device=device)
        idx = torch.tensor([1, 1], dtype=torch.int32, device=device)
        ident = pil_affine_fixed_terms(np.eye(3, dtype=np.float32))
        coef = torch.tensor([pil_affine_fixed_terms(total), ident], dtype=torch.int32, device=device)
        lib.freihand_augment(cache, torch.from_numpy(mk).to(device), idx, coef, out_i, out_m)
        want = torch.from_numpy(g[f"timg{i}"]).permute(2, 0, 1).float().div(255)
        assert torch.equal(out_i[0].cpu(), want), f"image warp case {i} (rot {rot})"
        wm = torch.round(torch.from_numpy(g[f"tmask{i}"]).float().div(255))
        assert torch.equal(out_m[0].cpu(), wm.unsqueeze(0).repeat(3, 1, 1)), f"mask warp case {i}"
        assert torch.equal(out_i[1].cpu(), torch.from_numpy(img).permute(2, 0, 1).float().div(255)), "identity warp"
        # K and joints (data/dataset.py:258-260, 271-275)
        np.testing.assert_array_equal(post.dot(g[f"K{i}"]).astype(np.float32), g[f"tK{i}"])


def freihand_batch_case(lib, device, seed=0, B=5, n=7, res=32, J=21, V=50):
    """hifihr_freihand_batch == hifihr_freihand_augment + the torch broadcast expressions of hifihr_amd/data.py:batch and
    traineval.data_dic (Ks, Ps, joints, verts, j2d_gt, scales, idxs, segms_gt)."""
    from hifihr_amd.data import batch_affine_terms
    from hifihr_amd.traineval import proj_func
    rng = np.random.default_rng(seed)
    rgbx = rng.integers(0, 256, (n, res, res, 4), dtype=np.uint8)
    mk = (rng.random((n, res, res)) > 0.5).astype(np.uint8) * 255
    Ks = np.tile(np.array([[400.0, 0, 112], [0, 410.0, 108], [0, 0, 1]], np.float32), (n, 1, 1)) + rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    joints = (rng.normal(0, 0.05, (n, J, 3)) + np.array([0, 0, 0.6])).astype(np.float32)
    verts = (rng.normal(0, 0.05, (n, V, 3)) + np.array([0, 0, 0.6])).astype(np.float32)
    scales = rng.random(n).astype(np.float32)
    idx = rng.integers(0, n, B)
    rots = rng.uniform(-np.pi, np.pi, B)
    fixed, post, rmat = batch_affine_terms(np.asarray([res // 2, res // 2]), res, [res, res], rots)
    packed = np.concatenate([idx.astype(np.int32), fixed.reshape(-1), post.reshape(-1).view(np.int32), rmat.reshape(-1).view(np.int32)])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    cache = d(rgbx).view(torch.int32).reshape(n, res, res)
    f = lambda *shape: torch.full(shape, 7.0, device=device)
    out = {"imgs": f(B, 3, res, res), "masks": f(B, 3, res, res), "segms_gt": torch.full((B, res, res), 7, dtype=torch.int64, device=device),
           "Ks": f(B, 3, 3), "Ps": f(B, 3, 4), "joints": f(B, J, 3), "verts": f(B, V, 3), "j2d_gt": f(B, J, 2), "scales": f(B),
           "idxs": torch.full((B,), 7, dtype=torch.int64, device=device)}
    lib.freihand_batch(cache, d(mk), d(Ks), d(joints), d(verts), d(scales), d(packed), B, out)
    wi, wm = f(B, 3, res, res), f(B, 3, res, res)
    lib.freihand_augment(cache, d(mk), d(idx.astype(np.int32)), d(fixed), wi, wm)
    assert torch.equal(out["imgs"], wi) and torch.equal(out["masks"], wm) and torch.equal(out["segms_gt"], wm[:, 0].long())
    il = torch.from_numpy(idx).long()
    post_t, rmat_t = torch.from_numpy(post), torch.from_numpy(rmat)
    wK = (post_t.unsqueeze(3) * torch.from_numpy(Ks)[il].unsqueeze(1)).sum(2)
    rot = lambda pts: (pts.unsqueeze(2) * rmat_t.unsqueeze(1)).sum(3)
    wj, wv = rot(torch.from_numpy(joints)[il]), rot(torch.from_numpy(verts)[il])
    close = lambda a, b, tol=2e-6: float((a.cpu() - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(out["Ks"], wK) and close(out["joints"], wj) and close(out["verts"], wv)
    assert torch.equal(out["Ps"][:, :, :3], out["Ks"]) and float(out["Ps"][:, :, 3].abs().max()) == 0.0
    assert close(out["j2d_gt"], proj_func(wj, wK), 1e-5)
    assert torch.equal(out["scales"].cpu(), torch.from_numpy(scales)[il]) and torch.equal(out["idxs"].cpu(), il)
    # hifihr_freihand_batch_step: the same outputs bit for bit + what a training iteration derives from them (train_hrnet.py:62-68,
    # models_res_nimble.py:228-235), for a root inside the skeleton and for "no root"
    for root_id in (min(9, J - 1), 0, -1):
        out2 = {k: torch.full_like(v, 7) for k, v in out.items()}
        out2.update({"root_xyz": f(B, 1, 3), "joints_rel": f(B, J, 3), "verts_rel": f(B, V, 3), "cam_ndc": f(B, 4)})
        lib.freihand_batch(cache, d(mk), d(Ks), d(joints), d(verts), d(scales), d(packed), B, out2, root_id=root_id, image_size=res)
        for k in out:
            assert torch.equal(out[k], out2[k]), k
        root = out["joints"][:, root_id:root_id + 1] if root_id >= 0 else torch.zeros(B, 1, 3, device=device)
        assert torch.equal(out2["root_xyz"], root)
        assert torch.equal(out2["joints_rel"], out["joints"] - root) and torch.equal(out2["verts_rel"], out["verts"] - root)
        K = out["Ks"].cpu()
        cam = torch.stack([-2 * K[:, 0, 0] / res, -2 * K[:, 1, 1] / res, 1 - 2 * K[:, 0, 2] / res, 1 - 2 * K[:, 1, 2] / res], 1)
        assert close(out2["cam_ndc"], cam, 1e-6)


def ho3d_batch_case(lib, device, golden_dir):
    """hifihr_ho3d_batch vs tests/golden/ho3d_path.npz: Pillow's own crop + resize outputs (bit for bit) and the reference's uv21_crop /
    K_crop lines; windows from hifihr_amd.data.ho3d_crop_windows, itself checked against the reference's window lines here."""
    import os
    from hifihr_amd.data import ho3d_crop_windows
    g = np.load(os.path.join(golden_dir, "ho3d_path.npz"))
    ids = [i for i in range(int(g["n"])) if f"img_crop{i}" in g.files]
    allids = list(range(int(g["n"])))
    center, scale, size, box = ho3d_crop_windows(np.stack([g[f"uv21_{i}"] for i in allids]), np.stack([g[f"noise{i}"] for i in allids]),
                                                 np.concatenate([g[f"scale_noise{i}"] for i in allids]))
    for k, i in enumerate(allids):
        assert np.array_equal(center[k], g[f"crop_center{i}"]) and scale[k] == g[f"scale{i}"][0] and size[k] == g[f"size{i}"][0], i
        x1, y1, sz = float(g[f"x1_{i}"].reshape(-1)[0]), float(g[f"y1_{i}"].reshape(-1)[0]), float(g[f"size{i}"].reshape(-1)[0])
        assert tuple(box[k]) == tuple(int(round(v)) for v in (x1, y1, x1 + sz, y1 + sz)), i
    FH, FW = g[f"img{ids[0]}"].shape[:2]
    frames = np.zeros((len(ids) + 1, FH, FW, 4), np.uint8); masks = np.zeros((len(ids) + 1, FH, FW), np.uint8)
    for k, i in enumerate(ids):
        frames[k + 1, :, :, :3] = g[f"img{i}"]; masks[k + 1] = g[f"mask{i}"]
    frames[0] = 200; masks[0] = 255
    order = [2, 0, 3, 1]                                      # batch order != cache order: the gather index is exercised
    sel = [ids[k] for k in order]
    B, S = len(order), 224
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    packed = np.concatenate([np.asarray([k + 1 for k in order], np.int32), np.stack([box[i] for i in sel]).reshape(-1),
                             np.concatenate([np.stack([center[i] for i in sel]), np.asarray([scale[i] for i in sel])[:, None]], 1)
                             .astype(np.float32).reshape(-1).view(np.int32)])
    Ks = np.stack([np.eye(3, dtype=np.float32)] + [g[f"K{i}"] for i in ids]); uv = np.stack([np.zeros((21, 2), np.float32)] + [g[f"uv21_{i}"] for i in ids])
    xyz = np.random.default_rng(0).normal(size=(len(ids) + 1, 21, 3)).astype(np.float32)
    ws = torch.empty(lib.ho3d_workspace_bytes(B, S) // 4 + 1, dtype=torch.int32, device=device)
    f = lambda *shape: torch.full(shape, 7.0, device=device)
    out = {"img_crop": f(B, 3, S, S), "hand_mask_crop": f(B, 1, S, S), "K_crop": f(B, 3, 3), "uv21_crop": f(B, 21, 2), "xyz21": f(B, 21, 3)}
    lib.ho3d_batch(d(frames).view(torch.int32).reshape(len(ids) + 1, FH, FW), d(masks), d(Ks), d(uv), d(xyz), d(packed), B, S, ws, out)
    for b, i in enumerate(sel):
        want = torch.from_numpy(g[f"img_crop{i}"]).permute(2, 0, 1).float().div(255)
        assert torch.equal(out["img_crop"][b].cpu(), want), f"frame crop {i}"
        wm = torch.round(torch.from_numpy(g[f"mask_crop{i}"]).float().div(255))
        assert torch.equal(out["hand_mask_crop"][b, 0].cpu(), wm), f"mask crop {i}"
        assert torch.equal(out["uv21_crop"][b].cpu(), torch.from_numpy(g[f"uv21_crop{i}"])), f"uv21_crop {i}"
        np.testing.assert_allclose(out["K_crop"][b].cpu().numpy(), g[f"K_crop{i}"], rtol=1e-6, atol=1e-4)
        assert torch.equal(out["xyz21"][b].cpu(), torch.from_numpy(xyz[order[b] + 1]))


# ------------------------------------------------------------------------------------------------
# one-launch weight re-layout (hifihr_weight_prep) == the separate transpose / Winograd weight transforms, bit for bit
# ------------------------------------------------------------------------------------------------
def weight_prep_case(lib, device, seed=0):
    gen = torch.Generator().manual_seed(seed)
    shapes = [(32, 64, 3), (64, 32, 3), (48, 16, 1), (16, 4, 7), (128, 128, 3)]
    jobs, want = [], []
    for K, C, R in shapes:
        w = torch.randn(K, R, R, C, generator=gen).to(device)
        wt = torch.empty(C, R, R, K, device=device)
        lib.weight_transpose(w, wt, K, R * R, C)
        jobs.append((w, torch.full((K * R * R * C,), 7.0, device=device), K, C, R * R, 0)); want.append(wt.reshape(-1))
        if R == 3:
            U = torch.empty(16 * K * C, device=device); U2 = torch.empty(16 * K * C, device=device)
            lib.wino_weight_transform(w, U, K, C, 0)
            lib.wino_weight_transform(wt, U2, C, K, 1)
            jobs.append((w, torch.full((16 * K * C,), 7.0, device=device), K, C, 9, 1)); want.append(U)
            jobs.append((w, torch.full((16 * K * C,), 7.0, device=device), K, C, 9, 2)); want.append(U2)
            U4 = torch.empty(36 * K * C, device=device); U42 = torch.empty(36 * K * C, device=device)      # F(4x4, 3x3): kinds 3 / 4
            lib.wino_weight_transform(w, U4, K, C, 0, 4)
            lib.wino_weight_transform(wt, U42, C, K, 1, 4)
            jobs.append((w, torch.full((36 * K * C,), 7.0, device=device), K, C, 9, 3)); want.append(U4)
            jobs.append((w, torch.full((36 * K * C,), 7.0, device=device), K, C, 9, 4)); want.append(U42)
    for K, C, R in ((64, 3, 7), (8, 5, 3), (4, 1, 1)):          # kind 5: channels zero-padded to a multiple of 4
        w = torch.randn(K, R, R, C, generator=gen)
        C4 = (C + 3) // 4 * 4
        ref = torch.zeros(K, R, R, C4); ref[..., :C] = w
        jobs.append((w.to(device), torch.full((K * R * R * C4,), 7.0, device=device), K, C, R * R, 5)); want.append(ref.reshape(-1))
    table = lib.prep_jobs(jobs, device)
    lib.weight_prep(table, len(jobs), 3)
    for (_, dst, K, C, RS, kind), ref in zip(jobs, want):
        assert torch.equal(dst.cpu(), ref.cpu()), (K, C, RS, kind)


def stem_c3_wgrad_case(lib, device, N=2, H=56, seed=0):
    """hifihr_conv2d_bwd_weight_c3 (gradient in the 3-channel parameter's layout) == hifihr_conv2d_bwd_weight on the padded
    problem, first three channels; accumulate semantics; bit-reproducible."""
    gen = torch.Generator().manual_seed(seed)
    K, R, stride, pad = 64, 7, 2, 3
    assert lib.conv2d_bwd_weight_c3_supported(N, H, H, K, R, R, stride, pad)
    assert not lib.conv2d_bwd_weight_c3_supported(N, H, H, K, 3, 3, 1, 1)
    OH = (H + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, H, 4, generator=gen); x[..., 3] = 0
    dy = torch.randn(N, OH, OH, K, generator=gen)
    xd, dyd = x.to(device), dy.to(device)
    nws = lib.conv2d_wgrad_workspace_bytes(N, H, H, 4, K, R, R, stride, pad)
    assert nws > 0
    ws = torch.empty(nws // 4, device=device)
    dw4 = torch.zeros(K, R, R, 4, device=device)
    lib.conv2d_bwd_weight(xd, dyd, dw4, N, H, H, 4, K, R, R, stride, pad, ws=ws)
    dw3 = torch.full((K, R, R, 3), 0.25, device=device)
    lib.conv2d_bwd_weight_c3(xd, dyd, dw3, N, H, H, K, R, R, stride, pad, ws)
    assert torch.equal(dw3 - 0.25, (dw4[..., :3] + 0.25) - 0.25), "3-channel stem gradient == padded problem (same slabs, same order)"
    again = torch.full((K, R, R, 3), 0.25, device=device)
    lib.conv2d_bwd_weight_c3(xd, dyd, again, N, H, H, K, R, R, stride, pad, ws)
    assert torch.equal(again, dw3)
    ref = torch.nn.functional.conv2d(x[..., :3].permute(3, 0, 1, 2).contiguous(), dy.permute(3, 0, 1, 2).contiguous(), None, 1, pad, stride)
    ref = ref[:, :, :R, :R].permute(1, 2, 3, 0)                   # [K][R][S][3]
    assert float((dw3.cpu() - 0.25 - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


# ------------------------------------------------------------------------------------------------
# grouped linear layers (csrc/mlp.hip *_group_kernel) == the single-layer launches
# ------------------------------------------------------------------------------------------------
def linear_group_case(lib, device, B=32, seed=0):
    gen = torch.Generator().manual_seed(seed)
    shapes = [(512, 128, 1), (512, 128, 1), (128, 48, 0), (128, 10, 0), (32, 3, 0), (32, 1, 0)]
    d = lambda t: t.to(device).contiguous()
    mem, ref = [], []
    for I, O, act in shapes:
        x = torch.randn(B, I, generator=gen); w = torch.randn(O, I, generator=gen) / I ** 0.5; b = torch.randn(O, generator=gen) * 0.1
        dy = torch.randn(B, O, generator=gen)
        m = dict(x=d(x), w=d(w), b=d(b), y=torch.full((B, O), 7.0, device=device), act=act)
        y1 = torch.empty(B, O, device=device)
        lib.linear_fwd(m["x"], m["w"], m["b"], act, y1)
        dz = torch.empty(B, O, device=device); dW = torch.full((O, I), 0.5, device=device); db = torch.full((O,), -0.5, device=device)
        dx = torch.full((B, I), 7.0, device=device)
        lib.linear_bwd(d(dy), y1, m["x"], m["w"], act, dz, dW, db, dx)
        m.update(dy=d(dy), dz=torch.empty(B, O, device=device), dW=torch.full((O, I), 0.5, device=device), db=torch.full((O,), -0.5, device=device),
                 dx=torch.full((B, I), 7.0, device=device))
        mem.append(m); ref.append((y1, dW, db, dx))
    mem[3]["dx"] = None                                   # a member without an input gradient
    lib.linear_fwd_group(mem)
    for m, r in zip(mem, ref):
        assert torch.equal(m["y"], r[0]), "grouped forward"
    lib.linear_bwd_group(mem)
    for i, (m, r) in enumerate(zip(mem, ref)):
        assert float((m["dW"] - r[1]).abs().max()) <= 1e-5 * float(r[1].abs().max()) and float((m["db"] - r[2]).abs().max()) <= 1e-4, i
        if m["dx"] is not None:
            assert float((m["dx"] - r[3]).abs().max()) <= 1e-5 * float(r[3].abs().max()) + 1e-6, i
    # members that read the SAME input may share ONE dx (ops._LinearGroup.backward does for the heads' first layers): it then holds the sum
    shared = torch.full((B, 512), 7.0, device=device)
    mem[1]["x"] = mem[0]["x"]
    lib.linear_bwd_group([dict(mem[0], dx=shared, dW=torch.zeros_like(mem[0]["dW"]), db=torch.zeros_like(mem[0]["db"])),
                          dict(mem[1], dx=shared, dW=torch.zeros_like(mem[1]["dW"]), db=torch.zeros_like(mem[1]["db"]))])
    dx0, dx1 = torch.empty(B, 512, device=device), torch.empty(B, 512, device=device)
    lib.linear_bwd_group([dict(mem[0], dx=dx0, dW=torch.zeros_like(mem[0]["dW"]), db=torch.zeros_like(mem[0]["db"])),
                          dict(mem[1], dx=dx1, dW=torch.zeros_like(mem[1]["dW"]), db=torch.zeros_like(mem[1]["db"]))])
    want = dx0 + dx1
    assert float((shared - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6, "shared dx"


MANO_PARENTS16 = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]     # ManoLayer's tree in its re-ordered joint index space (my_mano.py:395-434)


def random_lbs_tables(V, J, S, seed, K=4):
    """Small random skinned mesh: a random tree, <= K weights per vertex, a sparse convex joint regressor."""
    rng = np.random.RandomState(seed)
    vt = rng.randn(V, 3) * 0.05
    sd = rng.randn(V, 3, S) * 0.004
    parents = np.array([-1] + [rng.randint(0, j) for j in range(1, J)], dtype=np.int32)
    w = np.zeros((V, J))
    for v in range(V):
        idx = rng.choice(J, size=min(J, rng.randint(1, K + 1)), replace=False)
        w[v, idx] = rng.rand(idx.size) + 0.05
    w /= w.sum(1, keepdims=True)
    jr = np.zeros((J, V))
    for j in range(J):
        idx = rng.choice(V, size=min(V, 24), replace=False)
        jr[j, idx] = rng.rand(idx.size)
    jr /= jr.sum(1, keepdims=True)
    return tuple(np.ascontiguousarray(a, dtype=np.float32) for a in (vt, sd, jr, w)) + (parents,)


def lbs_case(lib, device, tabs, B, seed, pose_scale=0.6, vtol=2e-6, gtol=2e-4):
    """csrc/lbs.hip through the C-ABI vs oracle/lbs_oracle.py: verts, posed joints, d/dtheta, d/dbeta of a random scalar of both
    outputs.  (The backward's vertex sums are float atomics: gtol is relative to the largest gradient entry.)"""
    from oracle import lbs_oracle as lo
    vt, sd, jr, w, parents = tabs
    V, J, S = vt.shape[0], w.shape[1], sd.shape[2]
    gen = torch.Generator().manual_seed(seed)
    theta = torch.randn(B, J, 3, generator=gen) * pose_scale
    theta[0, min(1, J - 1)] = 0.0                           # the zero-angle branch of Rodrigues
    beta = torch.randn(B, S, generator=gen)
    wv, wj = torch.randn(B, V, 3, generator=gen), torch.randn(B, J, 3, generator=gen)
    th, be = theta.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rv, rj = lo.lbs_forward(vt, sd, jr, w, parents, th, be)
    ((rv * wv).sum() + (rj * wj).sum()).backward()
    h = lib.lbs_create(vt, sd, jr, w, parents)
    try:
        d = lambda t: t.to(device).contiguous()
        verts, joints = torch.empty(B, V, 3, device=device), torch.empty(B, J, 3, device=device)
        lib.lbs_fwd(h, d(theta), d(beta), verts, joints)
        scale = float(rv.detach().abs().max())
        assert float((verts.cpu() - rv.detach()).abs().max()) <= vtol * max(1.0, scale / 0.1), float((verts.cpu() - rv.detach()).abs().max())
        assert float((joints.cpu() - rj.detach()).abs().max()) <= vtol * max(1.0, scale / 0.1)
        scratch = torch.zeros(B, J, 12, device=device)
        gtheta, gbeta = torch.full((B, J, 3), 9.0, device=device), torch.zeros(B, S, device=device)
        lib.lbs_bwd(h, d(theta), d(beta), d(wv), d(wj), scratch, gtheta, gbeta)
        for got, ref, name in ((gtheta, th.grad, "gtheta"), (gbeta, be.grad, "gbeta")):
            if ref is None or ref.numel() == 0:
                continue
            err, mag = float((got.cpu() - ref).abs().max()), float(ref.abs().max())
            assert err <= gtol * mag, (name, err, mag)
        gtheta2, gbeta2 = torch.empty(B, J, 3, device=device), torch.zeros(B, S, device=device)      # gjoints = NULL
        scratch.zero_()
        lib.lbs_bwd(h, d(theta), d(beta), d(wv), None, scratch, gtheta2, gbeta2)
        th.grad = None; be.grad = None
        rv2, _ = lo.lbs_forward(vt, sd, jr, w, parents, th, be)
        (rv2 * wv).sum().backward()
        assert float((gtheta2.cpu() - th.grad).abs().max()) <= gtol * float(th.grad.abs().max())
        if S:
            assert float((gbeta2.cpu() - be.grad).abs().max()) <= gtol * float(be.grad.abs().max())
    finally:
        lib.lbs_destroy(h)


# ------------------------------------------------------------------------------------------------
# the reference trunk's batch-of-8 fixture (tests/golden/resnet18_b8.npz, tools/make_golden.py:gen_resnet18_b8)
# ------------------------------------------------------------------------------------------------
def resnet18_b8_inputs(g):
    """x, wl, wf regenerated from the generator seed the fixture was made with, verified against its float64 checksums."""
    gen = torch.Generator().manual_seed(1234)
    x = torch.rand(8, 3, 64, 64, generator=gen)
    wl = torch.randn(tuple(g["low"].shape), generator=gen); wf = torch.randn(tuple(g["feat"].shape), generator=gen)
    got = np.array([x.double().sum().item(), wl.double().sum().item(), wf.double().sum().item()])
    assert np.allclose(got, g["checksums"], rtol=0, atol=1e-9), "the CPU generator no longer reproduces the fixture's inputs"
    return x, wl, wf


def resnet18_b8_check(g, net, low, feat, out_atol, grad_rtol, grad_l2=None):
    """net = a module with torchvision's ResNet attribute names whose .grad fields are filled; every gradient the fixture holds:
    max error / max |reference| < grad_rtol and, if given, relative L2 error < grad_l2."""
    np.testing.assert_allclose(low.detach().cpu().numpy(), g["low"], atol=out_atol, rtol=1e-4)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat"], atol=out_atol, rtol=1e-4)
    params = dict(net.named_parameters())
    worst, l2 = {}, {}
    for key in g.files:
        if not key.startswith("g_"):
            continue
        toks = key[2:].split("_")               # g_layer2_0_downsample_0_weight -> layer2.0.downsample.0.weight
        name = ".".join(toks)
        grad = params[name].grad.detach().cpu().numpy()
        ref = g[key]
        if grad.ndim == 4:
            grad = grad[:8]
        worst[name] = float(np.abs(grad - ref).max() / (np.abs(ref).max() + 1e-12))
        l2[name] = float(np.linalg.norm((grad - ref).ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30))
    bad = {k: v for k, v in worst.items() if v >= grad_rtol}
    assert not bad, bad
    if grad_l2 is not None:
        bad = {k: v for k, v in l2.items() if v >= grad_l2}
        assert not bad, ("relative L2", bad)
    return worst, l2


# ------------------------------------------------------------------------------------------------
# batch statistics of activations with |mean| >> std (round-2 review: E[x^2] - mean^2 in fp32 loses the variance there)
# ------------------------------------------------------------------------------------------------
def bn_large_mean_case(lib, device, producer, seed=0, mean=50.0, std=0.1):
    """A batch-norm whose input has mean 50 / std 0.1 per channel (mean^2 / var = 2.5e5: in fp32 `E[x^2] - mean^2` has no correct digit
    left), statistics from every kind of producer, against F.batch_norm computed in float64 on the same values.
    producer: "stats" (hifihr_bn_stats on the tensor), "conv3x3" (implicit-GEMM epilogue), "conv1x1" (GEMM row-share epilogue),
    "halo" (layer-1 kernel), "wino4" / "wino2" (Winograd output transforms), "dw" (depthwise epilogue)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    d = lambda t: t.to(device).contiguous()
    if producer == "stats":
        N, H, W, K = 4, 9, 7, 64
        y = mean + std * torch.randn(N, K, H, W, generator=gen)
        yd = d(y.permute(0, 2, 3, 1))
        stats = torch.zeros(lib.bn_stats_floats(K), device=device)
        lib.bn_stats(yd, N * H * W, K, stats)
    elif producer == "dw":
        N, H, W, K = 2, 9, 8, 64
        x = mean + std * torch.randn(N, K, H, W, generator=gen)
        w = torch.full((K, 1, 3, 3), 1.0 / 9) + 0.01 * torch.randn(K, 1, 3, 3, generator=gen)
        y = F.conv2d(x.double(), w.double(), None, 1, 0, groups=K).float()
        OH, OW = y.shape[2], y.shape[3]
        yd = torch.empty(N, OH, OW, K, device=device); stats = torch.zeros(lib.bn_stats_floats(K), device=device)
        lib.dwconv2d_fwd(d(x.permute(0, 2, 3, 1)), d(w.reshape(K, 3, 3)), yd, N, H, W, K, OH, OW, 3, 1, 0, 0, stats=stats)
        H, W = OH, OW
    else:
        N, H, W, C, K, R = {"conv3x3": (2, 9, 7, 16, 64, 3), "conv1x1": (2, 8, 8, 32, 128, 1), "halo": (2, 10, 14, 64, 64, 3),
                            "wino4": (2, 8, 8, 64, 64, 3), "wino2": (2, 8, 8, 64, 64, 3)}[producer]
        # a strongly positive input and filters that are (almost) a positive centre tap: every output channel sits at a large mean with a
        # small spread, at the zero-padded border too
        x = mean + std * torch.randn(N, C, H, W, generator=gen)
        w = 1e-4 * torch.randn(K, C, R, R, generator=gen) / C
        w[:, :, R // 2, R // 2] = (1.0 + 0.05 * torch.randn(K, C, generator=gen)) / C
        pad = R // 2
        y = F.conv2d(x.double(), w.double(), None, 1, pad).float()
        xd, wd = d(x.permute(0, 2, 3, 1)), d(w.permute(0, 2, 3, 1))
        yd = torch.empty(N, H, W, K, device=device); stats = torch.zeros(lib.bn_stats_floats(K), device=device)
        if producer in ("wino4", "wino2"):
            m = 4 if producer == "wino4" else 2
            P, T = (m + 2) ** 2, lib.wino_tiles(N, H, W, m)
            U = torch.empty(P, K, C, device=device); V = torch.empty(P, T, C, device=device); Mm = torch.empty(P, T, K, device=device)
            lib.wino_weight_transform(wd, U, K, C, 0, m); lib.wino_input_transform(xd, V, N, H, W, C, m)
            lib.wino_gemm(V, U, Mm, N, H, W, C, K, ws=None, m=m); lib.wino_output_transform(Mm, yd, stats, N, H, W, K, m=m)
        else:
            lib.conv2d_fwd_bnstats(xd, wd, yd, stats, N, H, W, C, K, R, R, 1, pad)
    # interior channel statistics as they are (what the kernel saw is its OWN y: compare against statistics of that tensor in float64)
    yk = yd.cpu().double().reshape(-1, K)
    mu64, var64 = yk.mean(0), yk.var(0, unbiased=False)
    assert float((mu64.abs() / var64.sqrt()).min()) > 30.0, "the case is meant to have |mean| >> std"
    gamma = 1 + 0.1 * torch.randn(K, generator=gen); beta = 0.1 * torch.randn(K, generator=gen)
    out = torch.empty_like(yd); sm = torch.empty(K, device=device); si = torch.empty(K, device=device)
    rm, rv = torch.zeros(K, device=device), torch.ones(K, device=device)
    lib.bn_act_fwd(yd, stats, d(gamma), d(beta), None, 0, N * H * W, K, 1e-5, 0.1, out, sm, si, rm, rv)
    assert float(stats.abs().max()) == 0.0
    np.testing.assert_allclose(sm.cpu().double().numpy(), mu64.numpy(), rtol=2e-7, atol=0)
    inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
    np.testing.assert_allclose(si.cpu().double().numpy(), inv64.numpy(), rtol=2e-4, atol=0, err_msg=f"{producer}: 1/sqrt(var + eps)")
    ref = ((yk - mu64) * inv64 * gamma.double() + beta.double()).float().reshape(out.shape)
    # the normalised output amplifies the rounding of y and of the saved fp32 mean (ulp(50) / std = 4e-5 of a standard deviation each)
    assert float((out.cpu() - ref).abs().max()) <= 2e-3, (producer, float((out.cpu() - ref).abs().max()))
    np.testing.assert_allclose(rv.cpu().double().numpy(), (0.9 + 0.1 * var64 * (N * H * W) / (N * H * W - 1)).numpy(), rtol=2e-4)


# ------------------------------------------------------------------------------------------------
# batch-norm fused into the Winograd F(4x4, 3x3) input transform (csrc/wino4_bn.hip) vs the two separate launches
# ------------------------------------------------------------------------------------------------
def wino_bn_input_case(lib, device, N, H, W, C, residual, seed=0):
    """hifihr_wino_bn_input_transform == hifihr_bn_act_fwd (ReLU, optional residual) followed by hifihr_wino_input_transform_m, bit for bit
    (same scale / shift / residual / ReLU expression, same transform), and both against torch: V through the known input transform of a
    torch-computed activation."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(seed)
    d = lambda t: t.to(device).contiguous()
    x = torch.randn(N, H, W, C, generator=gen) * 1.3 + 0.2
    res = torch.randn(N, H, W, C, generator=gen) if residual else None
    gamma = 1 + 0.1 * torch.randn(C, generator=gen); beta = 0.1 * torch.randn(C, generator=gen)
    M = N * H * W
    T = lib.wino_tiles(N, H, W, 4)

    def fresh():
        st = torch.zeros(lib.bn_stats_floats(C), device=device)
        lib.bn_stats(d(x), M, C, st)
        return st, torch.zeros(C, device=device), torch.ones(C, device=device), torch.empty(C, device=device), torch.empty(C, device=device)
    # the two separate launches
    st, rm0, rv0, sm0, si0 = fresh()
    a0 = torch.empty(N, H, W, C, device=device)
    lib.bn_act_fwd(d(x), st, d(gamma), d(beta), d(res) if residual else None, 1, M, C, 1e-5, 0.1, a0, sm0, si0, rm0, rv0)
    V0 = torch.empty(36, T, C, device=device)
    lib.wino_input_transform(a0, V0, N, H, W, C, 4)
    # the fused launch
    st, rm1, rv1, sm1, si1 = fresh()
    V1 = torch.full((36, T, C), 7.0, device=device)
    out = torch.full((N, H, W, C), 7.0, device=device) if residual else None
    lib.wino_bn_input_transform(d(x), st, d(gamma), d(beta), d(res) if residual else None, out, V1, N, H, W, C, 4, 1e-5, 0.1, sm1, si1, rm1, rv1)
    assert float(st.abs().max()) == 0.0, "statistics slots and counters must come back zeroed"
    assert torch.equal(V1, V0), float((V1 - V0).abs().max())
    assert torch.equal(sm1, sm0) and torch.equal(si1, si0) and torch.equal(rm1, rm0) and torch.equal(rv1, rv0)
    if residual:
        assert torch.equal(out, a0)
    # and against torch
    ref = F.batch_norm(x.permute(0, 3, 1, 2), None, None, gamma, beta, True, 0.0, 1e-5)
    if residual:
        ref = ref + res.permute(0, 3, 1, 2)
    ref = F.relu(ref).permute(0, 2, 3, 1)
    assert float((a0.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def wino_bn_bwd_case(lib, device, N, H, W, C, residual, addend, seed=0):
    """The backward of the batch-norm / Winograd fusion (csrc/wino4_bn.hip) against the launches it replaces:
      hifihr_wino_output_transform_bnred + hifihr_bn_bwd_apply  ==  hifihr_wino_output_transform_m (+ add) + hifihr_bn_act_bwd
      hifihr_wino_bn_bwd_dual_transform                         ==  hifihr_bn_bwd_apply + hifihr_wino_input_dy_transform_m   (bit for bit)"""
    gen = torch.Generator().manual_seed(seed)
    d = lambda t: t.to(device).contiguous()
    M = N * H * W
    T = lib.wino_tiles(N, H, W, 4)
    x = d(torch.randn(N, H, W, C, generator=gen) * 1.3 + 0.2)
    res = d(torch.randn(N, H, W, C, generator=gen)) if residual else None
    gadd = d(torch.randn(N, H, W, C, generator=gen)) if addend else None
    gamma = d(1 + 0.1 * torch.randn(C, generator=gen)); beta = d(0.1 * torch.randn(C, generator=gen))
    Mm = d(torch.randn(36, T, C, generator=gen))
    # forward statistics and block output (for the mask)
    st = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.bn_stats(x, M, C, st)
    out = torch.empty(N, H, W, C, device=device); sm = torch.empty(C, device=device); si = torch.empty(C, device=device)
    lib.bn_act_fwd(x, st, gamma, beta, res, 1, M, C, 1e-5, 0.1, out, sm, si, None, None)
    # the separate launches
    dA = torch.empty(N, H, W, C, device=device)
    lib.wino_output_transform(Mm, dA, None, N, H, W, C, m=4)
    if addend:
        dA = dA + gadd
    red = torch.zeros(lib.bn_stats_floats(C), device=device)
    dx0 = torch.empty_like(x); dres0 = torch.empty_like(x); dg0 = torch.zeros(C, device=device); db0 = torch.zeros(C, device=device)
    lib.bn_act_bwd(dA, out if residual else None, x, sm, si, gamma, beta, 1, M, C, red, dx0, dres0, dg0, db0)
    # fused part 1 + apply
    g = torch.full((N, H, W, C), 7.0, device=device)
    red1 = torch.zeros(lib.bn_stats_floats(C), device=device)
    lib.wino_output_transform_bnred(Mm, x, out if residual else None, gadd, sm, si, gamma, beta, red1, g, N, H, W, C, 4)
    assert torch.equal(g, dres0), float((g - dres0).abs().max())            # the masked gradient, bit for bit
    red_keep = red1.clone()
    dx1 = torch.empty_like(x); dg1 = torch.zeros(C, device=device); db1 = torch.zeros(C, device=device)
    lib.bn_bwd_apply(g, x, sm, si, gamma, M, C, red1, dx1, dg1, db1)
    assert float(red1[:32 * 4 * C + 64].abs().max()) == 0.0
    tol = lambda ref: 2e-5 * float(ref.abs().max()) + 1e-7      # the reductions add in a different order
    assert float((dx1 - dx0).abs().max()) <= tol(dx0), float((dx1 - dx0).abs().max())
    assert float((dg1 - dg0).abs().max()) <= tol(dg0) and float((db1 - db0).abs().max()) <= tol(db0)
    # fused part 2 == apply + dual transform on the SAME sums, bit for bit
    V0 = torch.empty(36, T, C, device=device); Y0 = torch.empty(36, T, C, device=device)
    lib.wino_input_dy_transform(dx1, V0, Y0, N, H, W, C, 4)
    V1 = torch.full((36, T, C), 7.0, device=device); Y1 = torch.full((36, T, C), 7.0, device=device)
    dg2 = torch.zeros(C, device=device); db2 = torch.zeros(C, device=device)
    lib.wino_bn_bwd_dual_transform(g, x, sm, si, gamma, red_keep, V1, Y1, N, H, W, C, 4, dg2, db2)
    assert float(red_keep[:32 * 4 * C + 64].abs().max()) == 0.0
    assert torch.equal(V1, V0) and torch.equal(Y1, Y0), (float((V1 - V0).abs().max()), float((Y1 - Y0).abs().max()))
    assert torch.equal(dg2, dg1) and torch.equal(db2, db1)


# ------------------------------------------------------------------------------------------------
# joint_2d / bone_direc / bone_direc_3d (csrc/losses.hip joint_terms_*) vs the torch restatement pinned by tests/golden/losses.npz
# ------------------------------------------------------------------------------------------------
def joint_terms_case(lib, device, B, mse, seed=0, use2=True, use3=True):
    import torch.nn.functional as F
    from oracle.loss_oracle import bone_direction_loss
    gen = torch.Generator().manual_seed(seed)
    j2d = (torch.rand(B, 21, 2, generator=gen) * 224).requires_grad_(True); j2d_gt = torch.rand(B, 21, 2, generator=gen) * 224
    j3d = (torch.randn(B, 21, 3, generator=gen) * 0.05).requires_grad_(True); j3d_gt = torch.randn(B, 21, 3, generator=gen) * 0.05
    lam = (0.7, 1.3, 2.1)
    base = F.mse_loss if mse else F.l1_loss
    one = torch.ones(B, 21, 1)
    ref = [lam[0] * base(j2d_gt, j2d), lam[1] * bone_direction_loss(j2d, j2d_gt, one), lam[2] * bone_direction_loss(j3d, j3d_gt, one)]
    w = torch.tensor([0.9, -1.1, 0.6])
    tot = sum(wi * r for wi, r, u in zip(w, ref, (use2, use2, use3)) if u)
    tot.backward()
    d = lambda t: t.detach().to(device).contiguous()
    out = torch.empty(3, device=device)
    a2 = (d(j2d), d(j2d_gt)) if use2 else (None, None)
    a3 = (d(j3d), d(j3d_gt)) if use3 else (None, None)
    lib.joint_terms_fwd(a2[0], a2[1], a3[0], a3[1], mse, lam, out)
    for k, u in enumerate((use2, use2, use3)):
        want = float(ref[k]) if u else 0.0
        assert abs(float(out[k]) - want) <= 2e-5 * max(1.0, abs(want)), (k, float(out[k]), want)
    g2 = torch.full((B, 21, 2), 7.0, device=device) if use2 else None
    g3 = torch.full((B, 21, 3), 7.0, device=device) if use3 else None
    lib.joint_terms_bwd(a2[0], a2[1], a3[0], a3[1], mse, lam, d(w), g2, g3)
    if use2:
        assert float((g2.cpu() - j2d.grad).abs().max()) <= 1e-5 * float(j2d.grad.abs().max()) + 1e-9
    if use3:
        assert float((g3.cpu() - j3d.grad).abs().max()) <= 1e-5 * float(j3d.grad.abs().max()) + 1e-9
