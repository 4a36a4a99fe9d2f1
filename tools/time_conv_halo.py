"""Times ResNet layer 1's 3x3 convolution (64 -> 64 channels, 56 x 56, stride 1) forward / backward-data through the C-ABI.
HIFIHR_CONV_HALO=0 selects conv_igemm_kernel, the default conv_halo_kernel.  usage: python tools/time_conv_halo.py [B]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os  # noqa: E402

from hifihr_amd._lib import HifihrLib, get_lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# HALO_LIB: a diagnostic build of the library (tools/_probe/libhifihr_halo_ab<N>.so: ablations, wrong results, timing only)
lib = HifihrLib(os.environ["HALO_LIB"]) if os.environ.get("HALO_LIB") else get_lib()
dev = "cuda"
H = W = 56
x = torch.randn(B, H, W, 64, device=dev)
w = torch.randn(64, 3, 3, 64, device=dev) / 24.0
out = torch.empty(B, H, W, 64, device=dev)
dx = torch.empty(B, H, W, 64, device=dev)
scratch = torch.empty(64 * 9 * 64, device=dev)
dw = torch.zeros(64, 3, 3, 64, device=dev)
stats = torch.zeros(lib.bn_stats_floats(64), device=dev)
nb_f = lib.conv2d_workspace_bytes(B, H, W, 64, 64, 3, 3, 1, 1, False)
nb_b = lib.conv2d_workspace_bytes(B, H, W, 64, 64, 3, 3, 1, 1, True)
ws = torch.zeros(max(nb_f, nb_b, 4) // 4, device=dev)


def bench(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


flop = 2.0 * B * H * W * 64 * 64 * 9
for name, fn in (("fwd+bnstats", lambda: lib.conv2d_fwd_bnstats(x, w, out, stats, B, H, W, 64, 64, 3, 3, 1, 1, ws=ws)),
                 ("fwd", lambda: lib.conv2d_fwd(x, w, None, out, B, H, W, 64, 64, 3, 3, 1, 1, ws=ws)),
                 ("dgrad", lambda: lib.conv2d_bwd_data(x, w, dx, scratch, B, H, W, 64, 64, 3, 3, 1, 1, ws=ws)),
                 ("wgrad", lambda: lib.conv2d_bwd_weight(x, out, dw, B, H, W, 64, 64, 3, 3, 1, 1))):
    us = bench(fn)
    print(f"{name:12s} B={B}: {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s  ({flop / us / 1e6 / 157.3:.2f} of the f32 MFMA peak)")
