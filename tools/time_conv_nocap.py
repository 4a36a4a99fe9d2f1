#!/usr/bin/env python3
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import HifihrLib, LIB_PATH
from time_kernels import timeit
libs = {"cap128": HifihrLib(LIB_PATH), "nocap": HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_nocap.so"))}
B = 32
ws = torch.zeros(64 << 20, device="cuda")
for H, C, K in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (14, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda")
    gf = 2.0 * B * H * H * K * 9 * C / 1e9
    row = []
    for name, lib in libs.items():
        for occ in (4, 3):
            os.environ["HIFIHR_CONV_SK_VARIANT"] = "0"; os.environ["HIFIHR_CONV_SK_OCC"] = str(occ)
            t = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=20)
            row.append(f"{name} occ{occ} {gf / t * 1e3:5.1f}")
    print(f"H={H:3d} C={C:4d} K={K:4d} TF: " + " | ".join(row))
