#!/usr/bin/env python3
"""Data-parallel grid vs balanced (stream-K) schedule of the 64x64 MFMA convolution, forward and backward-data, B = 32."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
B = 32
for H, C, K in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda"); gy = torch.randn_like(y); dx = torch.empty_like(x); scr = torch.empty(w.numel(), device="cuda")
    nbf = lib.conv2d_workspace_bytes(B, H, H, C, K, 3, 3, 1, 1, False); nbb = lib.conv2d_workspace_bytes(B, H, H, C, K, 3, 3, 1, 1, True)
    ws = torch.zeros(max(nbf, nbb, 4) // 4, device="cuda")
    gf = 2.0 * B * H * H * K * 9 * C / 1e9
    t0 = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1), n=30)
    t1 = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=30) if nbf else float("nan")
    d0 = timeit(lambda: lib.conv2d_bwd_data(gy, w, dx, scr, B, H, H, C, K, 3, 3, 1, 1), n=30)
    d1 = timeit(lambda: lib.conv2d_bwd_data(gy, w, dx, scr, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=30) if nbb else float("nan")
    print(f"H={H:3d} C={C:4d} K={K:4d} {gf:6.2f} GF | fwd grid {t0:7.1f} us ({gf / t0 * 1e3:5.1f} TF)  balanced {t1:7.1f} us ({gf / t1 * 1e3:5.1f} TF)"
          f" | dgrad(+transpose) grid {d0:7.1f}  balanced {d1:7.1f}")
