"""Times the 64 -> 64 channel 3x3 convolution as register-resident Winograd F(2x2, 3x3) (hifihr_conv3x3_c64_wino: conv_wino2_kernel) against
the direct halo kernel (hifihr_conv2d_fwd: conv_halo_kernel) on ResNet layer 1's shape and on VGG19 conv1_2's.
usage: python tools/time_conv_wino2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hifihr_amd._lib import HifihrLib, get_lib  # noqa: E402

lib = HifihrLib(os.environ["W2_LIB"]) if os.environ.get("W2_LIB") else get_lib()
dev = "cuda"


def bench(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, H, W, epi) in ((32, 56, 56, False), (96, 224, 224, True), (8, 56, 56, False)):
    x = torch.randn(B, H, W, 64, device=dev)
    w = torch.randn(64, 3, 3, 64, device=dev) / 24.0
    b = torch.randn(64, device=dev) if epi else None
    U = torch.empty(16 * 64 * 64, device=dev)
    lib.wino_weight_transform(w, U, 64, 64, 0)
    o1 = torch.empty(B, H, W, 64, device=dev); o2 = torch.empty(B, H, W, 64, device=dev)
    stats = None if epi else torch.zeros(lib.bn_stats_floats(64), device=dev)
    nb = lib.conv2d_workspace_bytes(B, H, W, 64, 64, 3, 3, 1, 1, False)
    ws = torch.zeros(max(nb, 4) // 4, device=dev)
    if epi:
        direct = lambda: lib.conv2d_fwd(x, w, b, o1, B, H, W, 64, 64, 3, 3, 1, 1, act=1)
    else:
        direct = lambda: lib.conv2d_fwd_bnstats(x, w, o1, stats, B, H, W, 64, 64, 3, 3, 1, 1, ws=ws)
    wino = lambda: lib.conv3x3_c64_wino(x, U, b, epi, o2, stats, B, H, W)
    direct(); wino(); torch.cuda.synchronize()
    err = float((o1 - o2).abs().max()) / float(o1.abs().max())
    flop = 2.0 * B * H * W * 64 * 64 * 9
    t1, t2 = bench(direct), bench(wino)
    print(f"B={B} {H}x{W}{' +bias+relu' if epi else ' +stats'}: conv_halo_kernel {t1:8.1f} us ({flop / t1 / 1e6 / 157.3:.2f} of the f32 MFMA peak)   "
          f"conv_wino2_kernel {t2:8.1f} us ({flop / 2.25 / t2 / 1e6 / 157.3:.2f} of the peak on its own 2.25x fewer products; "
          f"{flop / t2 / 1e6:.0f} direct-equivalent TFLOP/s)   max relative difference {err:.2e}")
