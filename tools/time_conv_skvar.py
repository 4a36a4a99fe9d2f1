#!/usr/bin/env python3
"""Balanced-schedule variants (HIFIHR_CONV_SK_VARIANT x HIFIHR_CONV_SK_OCC) on the ResNet-18 3x3 shapes, forward, B = 32."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
B = 32
ws = torch.zeros(64 << 20, device="cuda")
for H, C, K in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (14, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda")
    gf = 2.0 * B * H * H * K * 9 * C / 1e9
    row = []
    for var, occ in [(0, 4), (0, 3), (0, 5), (1, 3), (1, 2), (1, 4), (2, 2), (3, 2), (3, 1)]:
        os.environ["HIFIHR_CONV_SK_VARIANT"] = str(var); os.environ["HIFIHR_CONV_SK_OCC"] = str(occ)
        if lib.conv2d_workspace_bytes(B, H, H, C, K, 3, 3, 1, 1, False) == 0:
            row.append(f"v{var}o{occ}   n/a"); continue
        t = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=20)
        assert float(ws.abs().max()) == 0
        row.append(f"v{var}o{occ} {gf / t * 1e3:5.1f}")
    print(f"H={H:3d} C={C:4d} K={K:4d} TF: " + " | ".join(row))
