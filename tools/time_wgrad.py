#!/usr/bin/env python3
"""conv_wgrad tile / split variants (HIFIHR_WGRAD_TILE, HIFIHR_WGRAD_SPLITS) on the ResNet-18 shapes, B = 32."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
B = 32
for H, C, K, Rr, s, p in [(224, 4, 64, 7, 2, 3), (56, 64, 64, 3, 1, 1), (28, 128, 128, 3, 1, 1), (14, 256, 256, 3, 1, 1), (14, 512, 512, 3, 1, 1)]:
    OH = (H + 2 * p - Rr) // s + 1
    x = torch.randn(B, H, H, C, device="cuda"); gy = torch.randn(B, OH, OH, K, device="cuda"); dw = torch.zeros(K, Rr, Rr, C, device="cuda")
    gf = 2.0 * B * OH * OH * K * Rr * Rr * C / 1e9
    row = []
    for tile in ("", "1", "2", "6", "8"):
        for splits in ("",):
            for k, v in (("HIFIHR_WGRAD_TILE", tile), ("HIFIHR_WGRAD_SPLITS", splits)):
                if v: os.environ[k] = v
                else: os.environ.pop(k, None)
            if tile == "" and False: continue
            t = timeit(lambda: lib.conv2d_bwd_weight(x, gy, dw, B, H, H, C, K, Rr, Rr, s, p), n=20)
            row.append(f"tile[{tile or 'dflt'}] {t:6.1f}us {gf / t * 1e3:5.1f}TF")
    print(f"H={H:3d} C={C:3d} K={K:3d} {gf:5.1f}GF: " + " | ".join(row))
