import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from hifihr_amd._lib import get_lib
from time_kernels import timeit
lib = get_lib()
for (N, H, C) in [(32, 112, 64), (32, 56, 64), (32, 28, 128), (32, 14, 256), (32, 14, 512)]:
    M = N * H * H
    x = torch.randn(M, C, device="cuda"); y = torch.empty_like(x); res = torch.randn_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
    stats = torch.zeros(lib.bn_stats_floats(C), device="cuda"); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    sm = torch.empty(C, device="cuda"); si = torch.empty(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda"); red = torch.zeros(lib.bn_stats_floats(C), device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ts = timeit(lambda: lib.bn_stats(x, M, C, stats))
    tf = timeit(lambda: lib.bn_act_fwd(x, stats, g, b, res, True, M, C, 1e-5, 0.1, y, sm, si, rm, rv))
    tb = timeit(lambda: lib.bn_act_bwd(dy, y, x, sm, si, g, b, 1, M, C, red, dx, dres, dg, db))
    mb = M * C * 4 / 1e6
    print(f"M={M} C={C} ({mb:.1f} MB/tensor): stats {ts:.1f} us ({mb/ts:.2f} TB/s)  fwd {tf:.1f} us ({3*mb/tf:.2f} TB/s)  bwd {tb:.1f} us ({(3+3+2)*mb/tb:.2f} TB/s)")
