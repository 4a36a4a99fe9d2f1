"""bn0 + swish inside the depthwise kernel's loads against the separate batch-norm launch, per EfficientNet-b3 block shape at batch 48:
forward (bn_act_fwd + dwconv2d_fwd  vs  bn_finalize_fwd + dwconv2d_fwd_bnswish) and backward-weight (plain on the activated tensor vs
_bnswish on the raw one).  usage: python tools/time_dw_bnswish.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.effnet import b3_block_table, static_same_pad
lib = get_lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
H = 112
seen = {}
tot = [0.0, 0.0, 0.0, 0.0]
for idx, (k, s, e, cin, cout) in enumerate(b3_block_table()):
    C = cin * e
    key = (k, s, C, H)
    if e != 1:
        pl, pr, pt, pb = static_same_pad(k, s)
        OH = (H + pt + pb - k) // s + 1
        if key not in seen:
            x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(C, k, k, device="cuda") / k
            g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
            M = B * H * H
            stats = torch.zeros(lib.bn_stats_floats(C), device="cuda"); lib.bn_stats(x, M, C, stats); keep = stats.clone()
            mean, inv = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
            a = torch.empty_like(x); y = torch.empty(B, OH, OH, C, device="cuda"); ys = torch.zeros(lib.bn_stats_floats(C), device="cuda")
            gy = torch.randn_like(y); dw = torch.zeros(C, k, k, device="cuda")
            def unfused():
                stats.copy_(keep)
                lib.bn_act_fwd(x, stats, g, b, None, 2, M, C, 1e-3, 0.01, a, mean, inv, None, None)
                lib.dwconv2d_fwd(a, w, y, B, H, H, C, OH, OH, k, s, pt, pl, stats=ys); ys.zero_()
            def fused():
                stats.copy_(keep)
                lib.bn_finalize_fwd(stats, M, C, 1e-3, 0.01, mean, inv, None, None)
                lib.dwconv2d_fwd_bnswish(x, mean, inv, g, b, w, y, B, H, H, C, OH, OH, k, s, pt, pl, stats=ys); ys.zero_()
            tu, tf = timeit(unfused), timeit(fused)
            wu = timeit(lambda: lib.dwconv2d_bwd_weight(a, gy, dw, B, H, H, C, OH, OH, k, s, pt, pl))
            wf = timeit(lambda: lib.dwconv2d_bwd_weight_bnswish(x, mean, inv, g, b, gy, dw, B, H, H, C, OH, OH, k, s, pt, pl))
            seen[key] = (tu, tf, wu, wf)
            del x, a, y, gy
        tu, tf, wu, wf = seen[key]
        tot[0] += tu; tot[1] += tf; tot[2] += wu; tot[3] += wf
        print(f"block {idx:2d} k{k} s{s} C={C:4d} H={H:3d}: forward bn+dw {tu:7.1f} us  fused {tf:7.1f} us | wgrad {wu:7.1f} us  fused {wf:7.1f} us")
    H = (H + sum(static_same_pad(k, s)[2:]) - k) // s + 1 if True else H
print(f"per step: forward {tot[0]:.0f} -> {tot[1]:.0f} us; backward-weight {tot[2]:.0f} -> {tot[3]:.0f} us  (includes two small copies / fills per call on both sides)")
