#!/usr/bin/env python3
"""One conv shape, a few launches each of fwd / dgrad / wgrad (for rocprofv3 --pmc runs)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
B, H, C, K, Rr, s, p = 32, 14, 512, 512, 3, 1, 1
OH = (H + 2 * p - Rr) // s + 1
x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, Rr, Rr, C, device="cuda") * 0.05
y = torch.empty(B, OH, OH, K, device="cuda"); gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
scr = torch.empty(w.numel(), device="cuda")
for _ in range(5):
    lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, Rr, Rr, s, p)
    lib.conv2d_bwd_weight(x, gy, dw, B, H, H, C, K, Rr, Rr, s, p)
torch.cuda.synchronize()
