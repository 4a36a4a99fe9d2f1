#!/usr/bin/env python3
"""PMC companion of conv_ablate.py: 10 launches of each variant (prod, p1..p4) of ONE shape, in that order, so that the
counter CSV can be split by dispatch order.  usage: conv_ablate_pmc.py H C K"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib, LIB_PATH

H, C, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (14, 512, 512)
B, Rr, s, p = 32, 3, 1, 1
libs = [HifihrLib(LIB_PATH)] + [HifihrLib(os.path.join(R, "tools", "_probe", f"libhifihr_p{n}.so")) for n in (1, 2, 3, 4)]
x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, Rr, Rr, C, device="cuda") * 0.05
y = torch.empty(B, H, H, K, device="cuda")
for lib in libs:
    for _ in range(10):
        lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, Rr, Rr, s, p)
    torch.cuda.synchronize()
