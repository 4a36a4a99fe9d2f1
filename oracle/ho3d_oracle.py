"""ORACLE (test infrastructure, not product code): the HO-3D training sample assembly, numpy on the CPU.

Restates reference data/dataset.py:1023-1215 (the `dat_name == 'HO3D'` branch of `__getitem__`): the crop window from the projected
joints (centre = middle of their bounding box + noise, size = 4 x the larger half extent, clamped, x scale noise), the image / mask
crops resized to 224 x 224 with torchvision's `resized_crop`, the crop's effect on the 2-D joints and on the intrinsics.

`resized_crop` is third-party: torchvision is a pip dependency of the reference at an unpinned version and absent from this image.
Its PIL backend is `img.crop((left, top, left + width, top + height)).resize((w, h), interpolation)` [recalled: torchvision
transforms/_functional_pil.py crop / resize], which is what is restated here on top of Pillow's own algorithms:
  * `Image.crop` rounds the box to integers (Python `round`, half to even) and fills what lies outside the image with zeros;
  * `Image.resize` for 8-bit images (Pillow src/libImaging/Resample.c, stable since 3.x): a horizontal then a vertical pass, each a
    normalised filter of support (filter support x max(scale, 1)) evaluated in double, quantised to 22-bit fixed point, accumulated in
    32-bit integers with rounding and clipped to uint8 BETWEEN the passes; bilinear = triangle, bicubic = Keys a = -0.5.
PINNED against Pillow itself (the version in this container, 12.2): tests/golden/ho3d_path.npz holds PIL's own outputs for random
crops (tools/make_golden.py:gen_ho3d_path), tests/test_oracle_ho3d.py checks this file against them bit for bit.
No imports from the product.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int, name: str):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (ksize, bounds [out, 2] = (xmin, count), kk [out, ksize] int32)."""
    filt, fsupport = FILTERS[name]
    scale = (in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_u8(img: np.ndarray, out_h: int, out_w: int, name: str) -> np.ndarray:
    """Image.resize((out_w, out_h), filter) of an 8-bit image [h, w] or [h, w, c]; equal sizes: a copy (Image.resize returns self.copy())."""
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    h, w, c = a.shape
    if (h, w) == (out_h, out_w):
        return img.copy()
    src = a.astype(np.int64)
    if w != out_w:                                             # horizontal pass
        _, bounds, kk = precompute_coeffs(w, 0.0, float(w), out_w, name)
        tmp = np.empty((h, out_w, c), np.uint8)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    if h != out_h:                                             # vertical pass (on the uint8 result of the horizontal one)
        _, bounds, kk = precompute_coeffs(h, 0.0, float(h), out_h, name)
        out = np.empty((out_h, src.shape[1], c), np.uint8)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (src[y0:y0 + n, :, :] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
    else:
        out = src.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def pil_crop_box(left: float, top: float, width: float, height: float):
    """Image.crop((left, top, left + width, top + height)): the box rounded half-to-even, as Pillow does."""
    return tuple(int(round(v)) for v in (left, top, left + width, top + height))


def pil_crop_u8(img: np.ndarray, box) -> np.ndarray:
    x0, y0, x1, y1 = box
    h, w = img.shape[:2]
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0)) + img.shape[2:], img.dtype)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def resized_crop_u8(img: np.ndarray, top: float, left: float, height: float, width: float, size: int, name: str) -> np.ndarray:
    """torchvision.transforms.functional.resized_crop(PIL image, top, left, height, width, [size, size], interpolation) [recalled]."""
    return pil_resize_u8(pil_crop_u8(img, pil_crop_box(left, top, width, height)), size, size, name)


def crop_window(uv21: np.ndarray, center_noise: np.ndarray, scale_noise: float, inp_res: int = 224, img_wh=(640.0, 480.0)):
    """dataset.py:1106-1161 (`ho_scope = 0`: the 21 projected joints alone, scale_num = 4), float32 like the torch code.
    uv21 [21, 2] (u, v); center_noise [2] = 5 * randn(2) (:1120); scale_noise = (1 - 1.1) * rand(1) + 1 - 0.1 (:1126).
    -> dict(crop_center [2], scale, crop_size_scales, x1, y1) as float32."""
    f = np.float32
    uv = uv21.astype(f)
    crop_center = (uv.max(0) + uv.min(0)) / f(2)                                           # :1114
    crop_center = center_noise.astype(f) + crop_center                                      # :1121
    min_uv = np.maximum(uv.min(0), np.zeros(2, f)) - np.array([10.0, 10.0], f)              # :1133
    max_uv = np.minimum(uv.max(0), np.array(img_wh, f)) + np.array([10.0, 10.0], f)         # :1135
    best = f(4) * np.maximum(max_uv - crop_center, crop_center - min_uv)                    # :1140
    best = best.max()
    best = np.minimum(np.maximum(best, f(50.0)), f(640.0))                                  # :1142
    scale = f(inp_res) * (f(1) / best)                                                      # :1145  (`int / tensor` is tensor.reciprocal() * int in torch)
    scale = np.minimum(scale, f(10.0))                                                      # :1147
    scale = f(scale * f(scale_noise))                                                       # :1148
    size = f(f(inp_res) * (f(1) / scale))                                                   # :1151  (reciprocal, then the product: two roundings)
    half = np.floor(size / f(2))                                                            # `crop_size_scales // 2` on a float tensor
    y1 = f(crop_center[1] - half)                                                           # :1154
    x1 = f(crop_center[0] - half)                                                           # :1157
    return {"crop_center": crop_center, "scale": f(scale), "crop_size_scales": size, "x1": x1, "y1": y1}


def crop_targets(uv21: np.ndarray, K: np.ndarray, win: dict, inp_res: int = 224):
    """dataset.py:1187-1215: uv21_crop = (uv21 - centre) * scale + inp_res // 2; K_crop = T . S . K."""
    f = np.float32
    c, s = win["crop_center"], win["scale"]
    uv_crop = (uv21.astype(f) - c[None, :]) * s + f(inp_res // 2)
    S = np.array([[s, 0, 0], [0, s, 0], [0, 0, 1]], f)
    t1, t2 = f(c[0] * s - f(inp_res // 2)), f(c[1] * s - f(inp_res // 2))
    T = np.array([[1, 0, -t1], [0, 1, -t2], [0, 0, 1]], f)
    return uv_crop.astype(f), (T @ (S @ K.astype(f))).astype(f)


def ho3d_sample(image_u8: np.ndarray, mask_u8: np.ndarray, uv21: np.ndarray, K: np.ndarray, center_noise, scale_noise, inp_res: int = 224):
    """The tensors of the sample dict that `data_dic`'s HO3D branch reads (img_crop, hand_mask_crop, uv21_crop, K_crop) for one frame.
    image_u8 [480, 640, 3]; mask_u8 [480, 640] = channel 0 of the reference's mask image (the hand)."""
    win = crop_window(uv21, np.asarray(center_noise), float(scale_noise), inp_res)
    top, left, size = float(win["y1"]), float(win["x1"]), float(win["crop_size_scales"])
    img = resized_crop_u8(image_u8, top, left, size, size, inp_res, "bilinear")            # :1165 (resized_crop's default: bilinear)
    msk = resized_crop_u8(mask_u8, top, left, size, size, inp_res, "bicubic")              # :1175 (BICUBIC), then to_tensor().round()
    uv_crop, K_crop = crop_targets(uv21, K, win, inp_res)
    return {"img_crop": img.transpose(2, 0, 1).astype(np.float32) / np.float32(255),
            "hand_mask_crop": np.round(msk.astype(np.float32) / np.float32(255))[None], "uv21_crop": uv_crop, "K_crop": K_crop, "window": win}
