#!/usr/bin/env python3
"""Winograd F(2x2,3x3) pipeline vs the direct (balanced) kernel, forward, B = 32: per-stage times."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
B = 32
ws = torch.zeros(96 << 20, device="cuda")
for H, C, K in [(28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)]:
    T = B * ((H + 1) // 2) ** 2
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda")
    U = torch.empty(16, K, C, device="cuda"); V = torch.empty(16, T, C, device="cuda"); M = torch.empty(16, T, K, device="cuda")
    stats = torch.zeros(lib.bn_stats_floats(K), device="cuda")
    td = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws), n=20)
    tw = timeit(lambda: lib.wino_weight_transform(w, U, K, C, 0), n=20)
    ti = timeit(lambda: lib.wino_input_transform(x, V, B, H, H, C), n=20)
    tg = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws), n=20)
    to = timeit(lambda: lib.wino_output_transform(M, y, stats, B, H, H, K), n=20)
    def allw():
        lib.wino_weight_transform(w, U, K, C, 0); lib.wino_input_transform(x, V, B, H, H, C)
        lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws); lib.wino_output_transform(M, y, stats, B, H, H, K)
    ta = timeit(allw, n=20)
    gf = 2.0 * 16 * T * C * K / 1e9
    print(f"H={H} C={C} K={K}: direct {td:6.1f} us | winograd total {ta:6.1f} us = weight {tw:5.1f} + input {ti:5.1f} + gemm {tg:6.1f} ({gf / tg * 1e3:5.1f} TF) + output {to:5.1f}")

print("--- backward-weight")
for H, C, K in [(28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)]:
    T = B * ((H + 1) // 2) ** 2
    x = torch.randn(B, H, H, C, device="cuda"); gy = torch.randn(B, H, H, K, device="cuda"); dw = torch.zeros(K, 3, 3, C, device="cuda")
    V = torch.empty(16, T, C, device="cuda"); Yt = torch.empty(16, T, K, device="cuda"); dU = torch.zeros(16, K, C, device="cuda")
    td = timeit(lambda: lib.conv2d_bwd_weight(x, gy, dw, B, H, H, C, K, 3, 3, 1, 1), n=20)
    ti = timeit(lambda: lib.wino_input_transform(x, V, B, H, H, C), n=20)
    ty = timeit(lambda: lib.wino_dy_transform(gy, Yt, B, H, H, K), n=20)
    tz = timeit(lambda: dU.zero_(), n=20)
    tg = timeit(lambda: lib.wino_wgrad_gemm(V, Yt, dU, B, H, H, C, K), n=20)
    tt = timeit(lambda: lib.wino_dw_transform(dU, dw, K, C), n=20)
    gf = 2.0 * 16 * T * C * K / 1e9
    print(f"H={H} C={C} K={K}: direct {td:6.1f} us | winograd {ti + ty + tz + tg + tt:6.1f} = input {ti:5.1f} + dy {ty:5.1f} + zero {tz:4.1f} + gemm {tg:6.1f} ({gf / tg * 1e3:5.1f} TF) + dw {tt:5.1f}")
