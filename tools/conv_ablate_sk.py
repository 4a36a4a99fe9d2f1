#!/usr/bin/env python3
"""Ablation of the BALANCED convolution kernel (workspace given): prod vs probe builds (p-1 = no barrier only)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import HifihrLib, LIB_PATH
from time_kernels import timeit
B = 32
libs = [("prod", HifihrLib(LIB_PATH))] + [(f"p{n}", HifihrLib(os.path.join(R, "tools", "_probe", f"libhifihr_p{n}.so"))) for n in (-1, 1, 2, 3, 4)]
ws = torch.zeros(64 << 20, device="cuda")
for H, C, K in [(28, 128, 128), (14, 256, 256), (14, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda")
    gf = 2.0 * B * H * H * K * 9 * C / 1e9
    row = []
    for name, lib in libs:
        t = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=20)
        row.append(f"{name} {gf / t * 1e3:6.1f}TF")
        ws.zero_()
    print(f"H={H} C={C} K={K}: " + " | ".join(row))
