"""Runs the renderer HIP kernel SOURCES on the CPU through tests/hostsim and checks them against
oracle/raster_oracle.c (face indices, bit-exact) and oracle/render_oracle.py (pixels, gradients)."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("image_size,aa", [(32, 3), (40, 2), (24, 1), (36, 2), (20, 3)])     # 36, 20: a ragged last 8-pixel tile (binning by binary search)
def test_render_fwd_bwd(hostsim_lib, synth_tables, image_size, aa):
    kc.render_case(hostsim_lib, synth_tables, "cpu", B=2, seed=10 + aa, image_size=image_size, aa=aa)


@pytest.mark.parametrize("image_size,aa", [(32, 3), (24, 2)])
def test_render_textures_uv(hostsim_lib, synth_tables, image_size, aa):
    """TexturesUV mode (texuv_fwd / texuv_bwd kernels around the tile kernels) vs the oracle's grid_sample restatement."""
    kc.render_uv_case(hostsim_lib, synth_tables, "cpu", B=2, seed=30 + aa, image_size=image_size, aa=aa)


def test_render_textures_uv_border_padding(hostsim_lib, synth_tables):
    kc.render_uv_case(hostsim_lib, synth_tables, "cpu", B=1, seed=41, image_size=32, aa=2, TH=9, TW=5, uv_scale=1.5)


def _fwd3_counts(lib, reset=True):
    import ctypes
    out = (ctypes.c_int * 4)()
    lib.c.hifihr_hostsim_render_fwd3_counts(out, 1 if reset else 0)
    return list(out)


def test_render_fwd3_reaches_split_merge_and_fill_paths(hostsim_lib, synth_tables):
    """render_fwd3_kernel (persistent workgroups on a work queue): at 32 x 32 pixels the whole MANO mesh falls on sixteen tiles, so their face
    lists are hundreds long and get SPLIT -- parts merged through the tile-shared buffer, the last arriver resolving -- while the border tiles
    are filled as background strips.  Face ids must stay bit-exact (render_case), and the counters say that every path ran."""
    _fwd3_counts(hostsim_lib)
    kc.render_case(hostsim_lib, synth_tables, "cpu", B=3, seed=77, image_size=32, aa=3, check_grad=False)
    merges, resolves, strips, _ = _fwd3_counts(hostsim_lib)
    assert merges >= 4 and resolves >= 2 and merges > resolves, (merges, resolves, strips)
    assert strips == 3 * 4 * 2                       # B x tile rows x two halves


def test_render_fwd3_equals_second_form(hostsim_lib, synth_tables, monkeypatch):
    """A tile cut into parts gives the same face ids and pixels as the one-workgroup-per-tile form on a dense mesh (a 5 990-vertex sphere-like
    skin: > 1 000 faces in a tile -> the 8-part cap and multi-pass parts)."""
    import numpy as np
    import torch
    from hifihr_amd.nimble_tables import synthetic_nimble_tables
    nt = synthetic_nimble_tables(0)
    faces = np.asarray(nt.faces)
    V = int(nt.v_template.shape[0])
    B, image_size, aa = 1, 24, 2
    verts, vcol, cam, lc, ld = kc.make_render_inputs(synth_tables, B, 5, image_size)
    gen = torch.Generator().manual_seed(3)
    mv = torch.as_tensor(np.asarray(nt.v_template), dtype=torch.float32)[None]
    mv = (mv - mv.mean(1, keepdim=True))
    mv = mv / mv.abs().max() * 0.3 + verts.mean(1, keepdim=True)
    col = 0.3 + 0.6 * torch.rand(B, V, 3, generator=gen)
    S = image_size * aa
    h = hostsim_lib.renderer_create(faces, V, image_size=image_size, aa=aa)
    try:
        ws = torch.empty(hostsim_lib.render_workspace_bytes(h, B), dtype=torch.uint8)
        rgba = torch.empty(B, 4, image_size, image_size); fid = torch.empty(B, S, S, dtype=torch.int32)
        _fwd3_counts(hostsim_lib)
        hostsim_lib.render_fwd(h, mv.contiguous(), col, cam, lc, ld, rgba, fid, ws)
        merges, resolves, _, _ = _fwd3_counts(hostsim_lib)
        assert merges >= 8 and resolves >= 1
        from oracle import render_oracle as ro
        rgba_ref, p2f_ref = ro.render(mv, col, cam, lc, ld, torch.as_tensor(faces).long(), image_size=image_size, aa=aa)
        assert (p2f_ref >= 0).mean() > 0.05
        np.testing.assert_array_equal(fid.numpy(), p2f_ref)
        np.testing.assert_allclose(rgba.numpy(), rgba_ref.numpy(), atol=2e-5, rtol=0)
    finally:
        hostsim_lib.renderer_destroy(h)


def test_render_textures_uv_texel_table_overflow(hostsim_lib, synth_tables):
    """More distinct texels per backward tile than the LDS table has slots (csrc/render_bwd.hip TexAcc): the overflow path (global atomics)."""
    kc.render_uv_case(hostsim_lib, synth_tables, "cpu", B=1, seed=43, image_size=32, aa=3, TH=384, TW=384)
