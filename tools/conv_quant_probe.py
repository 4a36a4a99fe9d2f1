#!/usr/bin/env python3
"""Does the hardware dispatcher spread T workgroups evenly over the 256 CUs?  Times the 64x64-tile forward conv (C=K=512, 3x3) at
row counts that give 768 / 784 / 1024 / 1040 / 1280 tiles: with an even spread time is proportional to ceil(T / 256)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
C = K = 512
for N, H in [(24, 16), (32, 14), (32, 16), (26, 16), (33, 16), (40, 16), (16, 16), (8, 16), (9, 16)]:
    x = torch.randn(N, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(N, H, H, K, device="cuda")
    t = timeit(lambda: lib.conv2d_fwd(x, w, None, y, N, H, H, C, K, 3, 3, 1, 1), n=30)
    tiles = ((N * H * H + 63) // 64) * (K // 64)
    gf = 2.0 * N * H * H * K * 9 * C / 1e9
    print(f"N={N:3d} H={H}: tiles {tiles:5d} ({tiles / 256:.3f}/CU)  {t:7.1f} us  {gf / t * 1e3:6.1f} TF   us per ceil(tiles/256) = {t / -(-tiles // 256):.1f}")
