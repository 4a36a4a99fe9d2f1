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
