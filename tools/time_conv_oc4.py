"""Backward-data of VGG19 conv1_1 (64 -> 4 channels, 3x3) and the ragged-width halo convolution at the perceptual loss's sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for N, H in ((48, 224), (16, 512)):
    dy = torch.randn(N, H, H, 64, device="cuda"); wt = torch.randn(4, 3, 3, 64, device="cuda"); dx = torch.empty(N, H, H, 4, device="cuda")
    t = timeit(lambda: lib.conv2d_bwd_data_pre(dy, wt, dx, N, H, H, 4, 64, 3, 3, 1, 1))
    print(f"conv1_1 backward-data {N} x {H}^2: {t:8.1f} us  [HIFIHR_CONV_OC4={os.environ.get('HIFIHR_CONV_OC4', '1')}]")
    x = torch.randn(N, H, H, 64, device="cuda"); w = torch.randn(64, 3, 3, 64, device="cuda") / 24; b = torch.randn(64, device="cuda"); y = torch.empty_like(x)
    t = timeit(lambda: lib.conv2d_fwd(x, w, b, y, N, H, H, 64, 64, 3, 3, 1, 1, act=1))
    print(f"conv1_2 forward (bias + ReLU) {N} x {H}^2: {t:8.1f} us = {2.0 * N * H * H * 64 * 64 * 9 / t / 1e6:6.1f} TF  [{lib.conv2d_describe(N, H, H, 64, 64, 3, 3, 1, 1, 0)}, HIFIHR_CONV_HALO={os.environ.get('HIFIHR_CONV_HALO', '1')}]")
