"""Times the ResNet stem (7x7 / stride 2 / pad 3, NHWC4 image -> 64 channels, 224^2 -> 112^2) through the C-ABI.
HIFIHR_CONV_STEM=0 selects conv_igemm_kernel's generic gather, the default conv_stem_kernel.  usage: python tools/time_conv_stem.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hifihr_amd._lib import get_lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = get_lib()
dev = "cuda"
x = torch.randn(B, 224, 224, 4, device=dev); x[..., 3] = 0
w = torch.randn(64, 7, 7, 4, device=dev) / 12.0; w[..., 3] = 0
out = torch.empty(B, 112, 112, 64, device=dev); gy = torch.randn_like(out)
dw = torch.zeros(64, 7, 7, 4, device=dev)
stats = torch.zeros(lib.bn_stats_floats(64), device=dev)


def bench(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


flop = 2.0 * B * 112 * 112 * 64 * 49 * 3            # the 3 real channels
print("kernel:", lib.conv2d_describe(B, 224, 224, 4, 64, 7, 7, 2, 3, 0), "/", lib.conv2d_describe(B, 224, 224, 4, 64, 7, 7, 2, 3, 2))
for name, fn in (("fwd+bnstats", lambda: lib.conv2d_fwd_bnstats(x, w, out, stats, B, 224, 224, 4, 64, 7, 7, 2, 3)),
                 ("wgrad", lambda: lib.conv2d_bwd_weight(x, gy, dw, B, 224, 224, 4, 64, 7, 7, 2, 3))):
    us = bench(fn)
    print(f"{name:12s} B={B}: {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s on the 3 real channels ({flop / us / 1e6 / 157.3:.2f} of the f32 MFMA peak)")
