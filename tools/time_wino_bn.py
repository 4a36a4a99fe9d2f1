"""The fused batch-norm / Winograd transform kernels alone, on the three layer shapes of the ResNet-18 trunk at B = 32 (us per launch)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
N = 32
for (H, C) in ((28, 128), (14, 256), (14, 512)):
    M = N * H * H; T = lib.wino_tiles(N, H, H, 4)
    x = torch.randn(N, H, H, C, device="cuda"); res = torch.randn_like(x); g = torch.randn_like(x); out = torch.empty_like(x)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    sm = torch.zeros(C, device="cuda"); si = torch.ones(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    V = torch.empty(36, T, C, device="cuda"); Y = torch.empty(36, T, C, device="cuda"); Mm = torch.randn(36, T, C, device="cuda")
    st = torch.zeros(lib.bn_stats_floats(C), device="cuda"); red = torch.zeros(lib.bn_stats_floats(C), device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    r = {}
    r["input_transform (plain)"] = timeit(lambda: lib.wino_input_transform(x, V, N, H, H, C, 4))
    r["bn_input_transform"] = timeit(lambda: lib.wino_bn_input_transform(x, st, gamma, beta, None, None, V, N, H, H, C, 4, 1e-5, 0.1, sm, si, rm, rv))
    r["bn_input_transform + res"] = timeit(lambda: lib.wino_bn_input_transform(x, st, gamma, beta, res, out, V, N, H, H, C, 4, 1e-5, 0.1, sm, si, rm, rv))
    r["output_transform (plain, stats)"] = timeit(lambda: (lib.wino_output_transform(Mm, out, st, N, H, H, C, m=4), st.zero_()))
    r["output_transform_bnred"] = timeit(lambda: (lib.wino_output_transform_bnred(Mm, x, None, None, sm, si, gamma, beta, red, g, N, H, H, C, 4), red.zero_()))
    r["output_transform_bnred + res + add"] = timeit(lambda: (lib.wino_output_transform_bnred(Mm, x, out, res, sm, si, gamma, beta, red, g, N, H, H, C, 4), red.zero_()))
    r["dual transform (plain)"] = timeit(lambda: lib.wino_input_dy_transform(g, V, Y, N, H, H, C, 4))
    r["bn_bwd_dual_transform"] = timeit(lambda: lib.wino_bn_bwd_dual_transform(g, x, sm, si, gamma, red, V, Y, N, H, H, C, 4, dg, db))
    r["bn_bwd_apply"] = timeit(lambda: lib.bn_bwd_apply(g, x, sm, si, gamma, M, C, red, out, dg, db))
    r["(zero_ alone)"] = timeit(lambda: st.zero_())
    print(f"H = {H}, C = {C}: " + "; ".join(f"{k} {v:.1f}" for k, v in r.items()))
