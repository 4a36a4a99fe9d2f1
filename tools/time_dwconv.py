"""The depthwise convolutions of EfficientNet-b3 at batch 48 (BASELINE configs[2]), one launch per direction and layer shape: us per launch and
the fraction of a 5 TB/s stream of the tensors each direction must move (x + y, dy + dx, x + dy).  usage: python tools/time_dwconv.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = int(os.environ.get("B", 48))
# (k, stride, channels, input H) of the b3 blocks (x repeats)
shapes = [(3, 1, 40, 112, 1), (3, 1, 24, 112, 1), (3, 2, 144, 112, 1), (3, 1, 192, 56, 2), (5, 2, 192, 56, 1), (5, 1, 288, 28, 2), (3, 2, 288, 28, 1),
          (3, 1, 576, 14, 4), (5, 1, 576, 14, 1), (5, 1, 816, 14, 4), (5, 2, 816, 14, 1), (5, 1, 1392, 7, 5), (3, 1, 1392, 7, 1), (3, 1, 2304, 7, 1)]
tot = [0.0, 0.0, 0.0]
for (k, s, C, H, rep) in shapes:
    OH = (H + s - 1) // s
    pad = max((OH - 1) * s + k - H, 0); pt = pad // 2
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(C, k, k, device="cuda"); y = torch.empty(B, OH, OH, C, device="cuda")
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    st = torch.zeros(lib.bn_stats_floats(C), device="cuda")
    tf = timeit(lambda: lib.dwconv2d_fwd(x, w, y, B, H, H, C, OH, OH, k, s, pt, pt, stats=st))
    tb = timeit(lambda: lib.dwconv2d_bwd_data(dy, w, dx, B, H, H, C, OH, OH, k, s, pt, pt))
    tw = timeit(lambda: lib.dwconv2d_bwd_weight(x, dy, dw, B, H, H, C, OH, OH, k, s, pt, pt))
    mb = (x.numel() + y.numel()) * 4 / 1e6
    floor = mb / 5.0            # us at 5 TB/s
    tot[0] += tf * rep; tot[1] += tb * rep; tot[2] += tw * rep
    print(f"k{k} s{s} C={C:4d} H={H:3d} x{rep}: fwd {tf:6.1f} us  bwd_data {tb:6.1f}  bwd_weight {tw:6.1f}   ({mb:6.1f} MB: {floor:5.1f} us at 5 TB/s; fwd = {floor / tf:.2f} of it)")
print(f"per step (x repeats): fwd {tot[0]:.0f} us, bwd_data {tot[1]:.0f} us, bwd_weight {tot[2]:.0f} us")
