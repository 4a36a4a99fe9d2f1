#!/usr/bin/env python3
"""Per-layer conv timing on the GPU box: hifihr MFMA implicit GEMM vs torch/MIOpen (NCHW and channels_last)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import torch.nn.functional as F
from hifihr_amd._lib import get_lib
from time_kernels import timeit
from test_gpu_conv import RESNET18_SHAPES

lib = get_lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tot = {"h_f": 0, "h_d": 0, "h_w": 0, "t_f": 0, "t_b": 0}
counts = {(56, 64, 64, 3, 1, 1): 4, (28, 128, 128, 3, 1, 1): 3, (14, 256, 256, 3, 1, 1): 3, (14, 512, 512, 3, 1, 1): 3}
print(f"B={B}   shape(H,C,K,R,s,p)            GFLOP | hifihr fwd / dgrad / wgrad us (TFLOP/s fwd) | torch(cl) fwd / bwd us")
for sh in RESNET18_SHAPES:
    H, C, K, Rr, s, p = sh
    OH = (H + 2 * p - Rr) // s + 1
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, Rr, Rr, C, device="cuda") * 0.05
    y = torch.empty(B, OH, OH, K, device="cuda"); gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    scr = torch.empty(w.numel(), device="cuda")
    gf = 2.0 * B * OH * OH * K * Rr * Rr * C / 1e9
    tf = timeit(lambda: lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, Rr, Rr, s, p))
    td = timeit(lambda: lib.conv2d_bwd_data(gy, w, dx, scr, B, H, H, C, K, Rr, Rr, s, p))
    tw = timeit(lambda: lib.conv2d_bwd_weight(x, gy, dw, B, H, H, C, K, Rr, Rr, s, p))
    xt = x.permute(0, 3, 1, 2).requires_grad_(True); wt = w.permute(0, 3, 1, 2).requires_grad_(True)   # channels_last views
    yt = F.conv2d(xt, wt, None, s, p)
    gt = gy.permute(0, 3, 1, 2)
    ttf = timeit(lambda: F.conv2d(xt, wt, None, s, p))
    def bwd():
        xt.grad = None; wt.grad = None
        F.conv2d(xt, wt, None, s, p).backward(gt)
    ttb = timeit(bwd) - ttf
    n = counts.get(sh, 1)
    tot["h_f"] += n * tf; tot["h_d"] += n * td * (0 if C == 4 else 1); tot["h_w"] += n * tw; tot["t_f"] += n * ttf; tot["t_b"] += n * ttb
    print(f"{str(sh):32s} {gf:7.2f} | {tf:8.1f} {td:8.1f} {tw:8.1f}  ({gf / tf * 1e3:6.1f}) | {ttf:8.1f} {ttb:8.1f}")
print("network totals (us): hifihr fwd %.0f dgrad %.0f wgrad %.0f = %.0f | torch fwd %.0f bwd %.0f = %.0f" % (
    tot["h_f"], tot["h_d"], tot["h_w"], tot["h_f"] + tot["h_d"] + tot["h_w"], tot["t_f"], tot["t_b"], tot["t_f"] + tot["t_b"]))
