#!/usr/bin/env python3
"""Ablation timing of conv_igemm_kernel on the GPU box: the production library against tools/_probe/libhifihr_p{1..4}.so
(tools/build_conv_probes.sh) -- which part of the main loop keeps the MFMA pipe idle?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import HifihrLib, LIB_PATH
from time_kernels import timeit

B = 32
SHAPES = [(56, 64, 64, 3, 1, 1), (28, 128, 128, 3, 1, 1), (14, 256, 256, 3, 1, 1), (7, 512, 512, 3, 1, 1)]
libs = [("prod", HifihrLib(LIB_PATH))] + [(f"p{n}", HifihrLib(os.path.join(R, "tools", "_probe", f"libhifihr_p{n}.so"))) for n in (1, 2, 3, 4)]
print("probes: p1 no global loads in loop | p2 + no LDS stores | p3 + no barrier | p4 + no LDS reads (MFMA only)")
for tile in ("2", "1", "0"):
    os.environ["HIFIHR_CONV_TILE"] = tile
    print(f"--- tile {['128x128', '128x64', '64x64'][int(tile)]}")
    for sh in SHAPES:
        H, C, K, Rr, s, p = sh
        OH = (H + 2 * p - Rr) // s + 1
        x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, Rr, Rr, C, device="cuda") * 0.05
        y = torch.empty(B, OH, OH, K, device="cuda")
        gf = 2.0 * B * OH * OH * K * Rr * Rr * C / 1e9
        row = []
        for name, lib in libs:
            t = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, Rr, Rr, s, p), n=30)
            row.append(f"{name} {t:7.1f}us {gf / t * 1e3:6.1f}TF")
        print(f"{str(sh):28s} {gf:6.2f} GF | " + " | ".join(row))
