"""GPU parity: renderer HIP kernels through the C ABI vs oracle/raster_oracle.c (bit-exact face indices) and
oracle/render_oracle.py (pixels 1e-4, gradients), plus size-independent properties at the BASELINE batch."""
import numpy as np
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hifihr_amd._lib import get_lib
    assert torch.cuda.is_available()
    return get_lib()


@pytest.mark.parametrize("image_size,aa,B", [(224, 3, 2), (64, 3, 5), (100, 2, 3), (37, 1, 2), (512, 1, 2), (512, 3, 1)])
def test_render_vs_oracle(lib, synth_tables, image_size, aa, B):
    kc.render_case(lib, synth_tables, "cuda", B=B, seed=20 + image_size, image_size=image_size, aa=aa, rgb_atol=1e-4)


@pytest.mark.parametrize("image_size,aa,B", [(224, 3, 2), (512, 1, 1), (64, 3, 3)])
def test_render_nimble_sized_mesh_vs_oracle(lib, synth_tables, image_size, aa, B):
    """The renderer on a mesh with the NIMBLE skin's size (5990 vertices, 11976 faces: 7.8x the MANO mesh -- per-tile face lists, list
    passes and the LDS budget all scale with it): face ids bit-exact, pixels and gradients against the oracle."""
    mesh = kc.nimble_sized_mesh(B, seed=image_size + aa)
    kc.render_case(lib, synth_tables, "cuda", B=B, seed=60 + image_size, image_size=image_size, aa=aa, rgb_atol=1e-4, mesh=mesh)


@pytest.mark.parametrize("image_size,aa,B", [(224, 3, 2), (64, 1, 3)])
def test_render_point_lights_vs_oracle(lib, synth_tables, image_size, aa, B):
    """The light_estimation = false branch (reference models_res_nimble.py:191-198): PointLights, direction = location - point per
    sample; pixels and the vertex / colour gradients (the direction now depends on the position) against the oracle."""
    kc.render_case(lib, synth_tables, "cuda", B=B, seed=80 + image_size, image_size=image_size, aa=aa, rgb_atol=1e-4, point_lights=True)


def _render(lib, h, verts, vcol, cam, lc, ld, H, aa):
    B = verts.shape[0]
    ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
    rgba = torch.empty(B, 4, H, H, device="cuda")
    fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
    return rgba, fid, ws


def test_render_properties_full_batch(lib, synth_tables):
    """BASELINE config: B=32, 224x224, aa=3.  Properties that need no oracle."""
    B, H, aa = 32, 224, 3
    verts, vcol, cam, lc, ld = (t.cuda().contiguous() for t in kc.make_render_inputs(synth_tables, B, 7, H))
    h = lib.renderer_create(synth_tables.faces, 778, image_size=H, aa=aa)
    rgba, fid, _ = _render(lib, h, verts, vcol, cam, lc, ld, H, aa)
    rgba2, fid2, _ = _render(lib, h, verts, vcol, cam, lc, ld, H, aa)
    assert torch.equal(rgba, rgba2) and torch.equal(fid, fid2)                     # forward is deterministic
    hit = (fid >= 0).float().view(B, H, aa, H, aa).mean(dim=(2, 4))
    torch.testing.assert_close(rgba[:, 3], hit, atol=1e-6, rtol=0)                 # alpha = coverage of the 3x3 block
    bg = rgba[:, 3] == 0
    assert bool((rgba[:, :3].permute(0, 2, 3, 1)[bg] == 1.0).all())                # untouched pixels are exactly white
    assert int(fid.max()) < 1538 and int(fid.min()) == -1
    assert float(hit.mean()) > 0.03
    # colours are linear in the vertex colours: render(a*c) - render(0) == a * (render(c) - render(0)) on covered pixels
    z, _, _ = _render(lib, h, verts, torch.zeros_like(vcol), cam, lc, ld, H, aa)
    half, _, _ = _render(lib, h, verts, 0.5 * vcol, cam, lc, ld, H, aa)
    torch.testing.assert_close(half[:, :3] - z[:, :3], 0.5 * (rgba[:, :3] - z[:, :3]), atol=2e-6, rtol=1e-5)
    # shared [V,3] colours == the same colours repeated per batch item
    shared, _, _ = _render(lib, h, verts, vcol[0].contiguous(), cam, lc, ld, H, aa)
    rep, _, _ = _render(lib, h, verts, vcol[:1].repeat(B, 1, 1).contiguous(), cam, lc, ld, H, aa)
    assert torch.equal(shared, rep)
    lib.renderer_destroy(h)


def test_render_mesh_behind_camera_is_empty(lib, synth_tables):
    B, H, aa = 1, 64, 3
    verts, vcol, cam, lc, ld = (t.cuda().contiguous() for t in kc.make_render_inputs(synth_tables, B, 3, H))
    verts[..., 2] -= 5.0
    h = lib.renderer_create(synth_tables.faces, 778, image_size=H, aa=aa)
    rgba, fid, _ = _render(lib, h, verts, vcol, cam, lc, ld, H, aa)
    assert int(fid.max()) == -1 and float(rgba[:, 3].max()) == 0.0 and float(rgba[:, :3].min()) == 1.0
    lib.renderer_destroy(h)


@pytest.mark.parametrize("B,image_size,aa", [(2, 224, 3), (3, 64, 2)])
def test_render_textures_uv(lib, synth_tables, B, image_size, aa):
    """TexturesUV mode of the renderer (two passes around the fused tile kernels) vs the oracle's grid_sample restatement: face ids exact,
    pixels 1e-4, gradients w.r.t. vertices (incl. the path through uv), texture maps and light."""
    kc.render_uv_case(lib, synth_tables, "cuda", B=B, seed=90 + image_size, image_size=image_size, aa=aa, rgb_atol=1e-4)


def test_render_textures_uv_border_padding(lib, synth_tables):
    """uvs outside [0, 1] (border padding: clamped coordinate, no uv gradient there) and a tiny odd-sized texture."""
    kc.render_uv_case(lib, synth_tables, "cuda", B=2, seed=97, image_size=96, aa=3, TH=9, TW=5, rgb_atol=1e-4, uv_scale=1.5)


def test_render_textures_uv_texel_table_overflow(lib, synth_tables):
    """A texture fine enough that a 16 x 16-pixel tile of the backward touches more distinct texels than its LDS table holds (2 048 slots:
    csrc/render_bwd.hip TexAcc): the texels that find the table full go out as global atomics -- same gradients either way."""
    kc.render_uv_case(lib, synth_tables, "cuda", B=2, seed=131, image_size=224, aa=3, TH=512, TW=512, rgb_atol=1e-4)
