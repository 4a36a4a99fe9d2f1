"""Times the generic LBS kernels (csrc/lbs.hip) at the NIMBLE sizes (V = 5990, J = 25, S = 20) and the MANO kernels beside them.
usage: python tools/time_lbs.py [B]   -> one line per kernel: average us over 200 launches, algorithmic GB/s."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from hifihr_amd import ops                                    # noqa: E402
from hifihr_amd.nimble_tables import synthetic_nimble_tables  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = synthetic_nimble_tables(0)
h = ops.LbsHandle(t.v_template, t.shapedirs, t.J_regressor, t.weights, t.parents)
lib = h.lib
dev = "cuda"
theta, beta = torch.randn(B, 25, 3, device=dev) * 0.3, torch.randn(B, 20, device=dev)
verts, joints = torch.empty(B, 5990, 3, device=dev), torch.empty(B, 25, 3, device=dev)
gv, gj = torch.randn(B, 5990, 3, device=dev), torch.randn(B, 25, 3, device=dev)
zero = torch.zeros(B * (25 * 12 + 20), device=dev)
gtheta = torch.empty(B, 25, 3, device=dev)


def bench(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def bwd():
    zero.zero_()
    lib.lbs_bwd(h.h, theta, beta, gv, gj, zero[:B * 300], gtheta, zero[B * 300:].view(B, 20))


tab_bytes = 4 * 5990 * (3 + 60 + 8)                       # template + shapedirs + 4 (index, weight) pairs, read once per launch (L2 after that)
fwd_us = bench(lambda: lib.lbs_fwd(h.h, theta, beta, verts, joints))
bwd_us = bench(bwd)
fwd_bytes = B * 5990 * 12 + tab_bytes
bwd_bytes = B * 5990 * 12 + tab_bytes
print(f"lbs_fwd  B={B}: {fwd_us:8.1f} us  algorithmic {fwd_bytes / 1e6:.2f} MB -> {fwd_bytes / fwd_us / 1e3:.1f} GB/s")
print(f"lbs_bwd  B={B}: {bwd_us:8.1f} us  (memset + vertex kernel + chain kernel)  algorithmic {bwd_bytes / 1e6:.2f} MB -> {bwd_bytes / bwd_us / 1e3:.1f} GB/s")
