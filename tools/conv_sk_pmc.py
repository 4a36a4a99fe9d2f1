#!/usr/bin/env python3
"""10 launches of the balanced forward conv (H C K from argv) for rocprofv3 --pmc runs."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
H, C, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (14, 512, 512)
B = 32
x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
y = torch.empty(B, H, H, K, device="cuda"); ws = torch.zeros(64 << 20, device="cuda")
for _ in range(10):
    lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws)
torch.cuda.synchronize()
