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


def test_product_window_code_matches_oracle_and_reference(g):
    """hifihr_amd.data.ho3d_crop_windows (the batched host code the device path uses) == the oracle's per-sample restatement == the
    reference's lines, on the golden joints and on random ones (windows larger than the frame, joints outside it, the 10x scale clamp)."""
    from hifihr_amd.data import ho3d_crop_windows
    rng = np.random.default_rng(7)
    uv = [g[f"uv21_{i}"] for i in range(int(g["n"]))]
    noise = [g[f"noise{i}"] for i in range(int(g["n"]))]
    sn = [float(g[f"scale_noise{i}"][0]) for i in range(int(g["n"]))]
    for spread, cx, cy in ((3.0, 320, 240), (80.0, 20, 460), (400.0, 600, 30), (1.0, -50, 500), (150.0, 320, 240)):
        uv.append(np.stack([cx + rng.normal(size=21) * spread, cy + rng.normal(size=21) * spread], 1).astype(np.float32))
        noise.append((rng.normal(size=2) * 5).astype(np.float32)); sn.append(float(np.float32(0.9 - 0.1 * rng.random())))
    center, scale, size, box = ho3d_crop_windows(np.stack(uv), np.stack(noise), np.asarray(sn, np.float32))
    for k in range(len(uv)):
        win = ho.crop_window(uv[k], noise[k], sn[k])
        assert np.array_equal(center[k], win["crop_center"]) and scale[k] == win["scale"] and size[k] == win["crop_size_scales"]
        assert tuple(box[k]) == ho.pil_crop_box(float(win["x1"]), float(win["y1"]), float(size[k]), float(size[k]))
    for i in range(int(g["n"])):
        assert scale[i] == g[f"scale{i}"][0] and size[i] == g[f"size{i}"][0]


def test_product_window_from_hand_box_corners():
    """The evaluation split (dataset.py:1071-1080) forms the window from the two corners of the hand bounding box: the batched host code on
    [B, 2, 2] points == the oracle's lines on the same two points (boxes inside, across the border of and larger than the 640 x 480 frame)."""
    from hifihr_amd.data import ho3d_crop_windows
    rng = np.random.default_rng(11)
    boxes = np.array([[[150, 100], [260, 210]], [[400, 300], [700, 520]], [[-30, -20], [90, 140]], [[0, 0], [640, 480]], [[300, 200], [304, 203]]], np.float32)
    noise = (rng.normal(size=(len(boxes), 2)) * 5).astype(np.float32)
    sn = (0.9 - 0.1 * rng.random(len(boxes))).astype(np.float32)
    center, scale, size, box = ho3d_crop_windows(boxes, noise, sn)
    for k in range(len(boxes)):
        win = ho.crop_window(boxes[k], noise[k], float(sn[k]))
        assert np.array_equal(center[k], win["crop_center"]) and scale[k] == win["scale"] and size[k] == win["crop_size_scales"]
        assert tuple(box[k]) == ho.pil_crop_box(float(win["x1"]), float(win["y1"]), float(size[k]), float(size[k]))
