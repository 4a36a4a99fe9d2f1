"""oracle/ho3d_oracle.py (the HO-3D sample assembly, reference data/dataset.py:1105-1215) against tests/golden/ho3d_path.npz: the crop
window / cropped joints / K_crop of the reference's own lines executed from source, and Pillow's crop + resize outputs for those windows
(tools/make_golden.py:gen_ho3d_path).  Integer / pixel results bit for bit; float32 results to the last bit where the operation order is
the reference's."""
import os

import numpy as np
import pytest

from oracle import ho3d_oracle as ho


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ho3d_path.npz"))


def test_crop_window_matches_reference_lines(g):
    for i in range(int(g["n"])):
        win = ho.crop_window(g[f"uv21_{i}"], g[f"noise{i}"], float(g[f"scale_noise{i}"][0]))
        np.testing.assert_array_equal(win["crop_center"], g[f"crop_center{i}"])
        assert win["scale"] == g[f"scale{i}"].reshape(()) and win["crop_size_scales"] == g[f"size{i}"].reshape(())
        assert win["x1"] == g[f"x1_{i}"].reshape(()) and win["y1"] == g[f"y1_{i}"].reshape(())
        uv_crop, K_crop = ho.crop_targets(g[f"uv21_{i}"], g[f"K{i}"], win)
        np.testing.assert_array_equal(uv_crop, g[f"uv21_crop{i}"])
        np.testing.assert_allclose(K_crop, g[f"K_crop{i}"], rtol=1e-6, atol=1e-4)      # (torch.mm's summation order is not specified)


def test_resized_crop_matches_pillow(g):
    seen = 0
    for i in range(int(g["n"])):
        if f"img_crop{i}" not in g.files:
            continue
        top, left, size = (float(g[k + str(i)].reshape(-1)[0]) for k in ("y1_", "x1_", "size"))
        assert np.array_equal(ho.resized_crop_u8(g[f"img{i}"], top, left, size, size, 224, "bilinear"), g[f"img_crop{i}"]), i
        assert np.array_equal(ho.resized_crop_u8(g[f"mask{i}"], top, left, size, size, 224, "bicubic"), g[f"mask_crop{i}"]), i
        seen += 1
    assert seen >= 4


def test_resample_coefficients_are_normalised():
    """Resample.c's fixed-point rows sum to 2^22 within the rounding of their entries (property, any size pair)."""
    for in_size, out_size, name in ((640, 224, "bilinear"), (100, 224, "bicubic"), (367, 224, "bicubic"), (224, 224, "bilinear")):
        ksize, bounds, kk = ho.precompute_coeffs(in_size, 0.0, float(in_size), out_size, name)
        assert kk.shape == (out_size, ksize) and np.abs(kk.sum(1) - (1 << ho.PRECISION_BITS)).max() <= ksize
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= in_size).all()
