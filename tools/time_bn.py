"""Times the batch-norm kernels (csrc/bn.hip) at the ResNet-18 shapes of the bench step (B = 32) through the C-ABI and prints the
algorithmic HBM rate of each launch group: forward = read x (+ residual), write y; backward = reduce (read dy, x [, y]) + apply
(read dy, x [, y], write dx [, dres]).  usage: python tools/time_bn.py [B]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from hifihr_amd._lib import get_lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = get_lib()
dev = "cuda"


def bench(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot_f = tot_b = 0.0
# (name, H, C, residual, launches per step)
for name, H, C, res, cnt in (("stem", 112, 64, False, 1), ("layer1 bn1", 56, 64, False, 2), ("layer1 bn2+res", 56, 64, True, 2),
                             ("layer2 bn1", 28, 128, False, 3), ("layer2 bn2+res", 28, 128, True, 2), ("layer3 bn1", 14, 256, False, 3),
                             ("layer3 bn2+res", 14, 256, True, 2), ("layer4 bn1", 7, 512, False, 3), ("layer4 bn2+res", 7, 512, True, 2)):
    M = B * H * H
    x = torch.randn(M, C, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    r = torch.randn_like(x) if res else None
    dres = torch.empty_like(x) if res else None
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    stats = torch.zeros(lib.bn_stats_floats(C), device=dev); red = torch.zeros(lib.bn_stats_floats(C), device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

    def fwd():        # (the slot buffer is self-cleaning: repeated calls see all-zero statistics, which changes no timing)
        lib.bn_act_fwd(x, stats, gamma, beta, r, 1, M, C, 1e-5, 0.1, y, mean, invstd, rm, rv)

    def bwd():
        lib.bn_act_bwd(dy, y if res else None, x, mean, invstd, gamma, beta, 1, M, C, red, dx, dres, dg, db)

    t_f = bench(fwd)
    t_b = bench(bwd)
    nb = M * C * 4
    bf = nb * (3 if res else 2)
    bb = nb * ((3 if res else 2) + (5 if res else 3))
    tot_f += t_f * cnt; tot_b += t_b * cnt
    print(f"{name:16s} M={M:7d} C={C:4d}: fwd {t_f:6.1f} us {bf / t_f / 1e6:5.2f} TB/s | bwd (reduce + apply) {t_b:6.1f} us {bb / t_b / 1e6:5.2f} TB/s")
print(f"per step (launch counts of ResNet-18): fwd {tot_f:.0f} us, bwd {tot_b:.0f} us")

# the fused stem: bn1 + ReLU + MaxPool2d(3, 2, 1) (bn_relu_pool_fwd_kernel; bn_pool_bwd_reduce_kernel + bn_pool_bwd_apply_kernel)
N, H, C = B, 112, 64
OH = (H - 1) // 2 + 1
x = torch.randn(N, H, H, C, device=dev); gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
stats0 = torch.zeros(lib.bn_stats_floats(C), device=dev)
lib.bn_stats(x, N * H * H, C, stats0)
stats = stats0.clone()
pooled = torch.empty(N, OH, OH, C, device=dev); tap = torch.empty(N * OH * OH * C, dtype=torch.uint8, device=dev)
mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
def fwd():
    stats.copy_(stats0)
    lib.bn_relu_maxpool_fwd(x, stats, gamma, beta, N, H, H, C, 1e-5, 0.1, pooled, tap, mean, invstd, None, None)
t_copy = bench(lambda: stats.copy_(stats0))
t_f = bench(fwd) - t_copy
gy = torch.randn_like(pooled); red = torch.zeros(lib.bn_stats_floats(C), device=dev); dx = torch.empty_like(x)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
t_b = bench(lambda: lib.bn_relu_maxpool_bwd(gy, tap, x, mean, invstd, gamma, beta, N, H, H, C, red, dx, dg, db))
mb_f = (x.numel() * 4 + pooled.numel() * 5) / 1e6
mb_b = (2 * x.numel() * 4 + 2 * pooled.numel() * 5 + x.numel() * 4) / 1e6
print(f"fused stem (bn1 + ReLU + maxpool) B={B}: fwd {t_f:6.1f} us ({mb_f / t_f * 1e-3 * 1e3:5.2f} TB/s of {mb_f:.0f} MB)   "
      f"bwd (reduce + apply) {t_b:6.1f} us ({mb_b / t_b * 1e-3 * 1e3:5.2f} TB/s of {mb_b:.0f} MB)")
