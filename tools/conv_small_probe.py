import subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(96, 4, 64, 7, 2, 3), (24, 64, 64, 3, 1, 1), (24, 64, 128, 3, 2, 1), (12, 128, 128, 3, 1, 1), (24, 64, 128, 1, 2, 0),
          (12, 128, 256, 3, 2, 1), (6, 256, 256, 3, 1, 1), (12, 128, 256, 1, 2, 0), (6, 256, 512, 3, 1, 1), (6, 512, 512, 3, 1, 1), (6, 256, 512, 1, 1, 0)]
T = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, kernel_cases as kc
from hifihr_amd._lib import get_lib
H, C, K, Rr, s, p = %r
kc.conv_case(get_lib(), "cuda", 2, H, H, C, K, Rr, s, p, seed=1, rtol=5e-5) if C != 4 else None
if C == 4:
    lib = get_lib(); import torch
    x = torch.randn(2, H, H, C, device="cuda"); w = torch.randn(K, Rr, Rr, C, device="cuda"); OH = (H + 2*p - Rr)//s + 1
    y = torch.empty(2, OH, OH, K, device="cuda"); dw = torch.zeros_like(w)
    lib.conv2d_fwd(x, w, None, y, 2, H, H, C, K, Rr, Rr, s, p); lib.conv2d_bwd_weight(x, torch.randn_like(y), dw, 2, H, H, C, K, Rr, Rr, s, p)
torch.cuda.synchronize(); print("ok")
'''
for sh in SHAPES:
    r = subprocess.run([sys.executable, "-c", T % (R, R, sh)], capture_output=True, text=True)
    err = [l for l in (r.stderr + r.stdout).splitlines() if "rror" in l or "fault" in l or "Assert" in l]
    print(sh, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:60], err[-1][:200] if err else "")
