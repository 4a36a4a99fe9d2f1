"""Back-to-back HIP-event timing of the small latency-bound kernels at the BASELINE configs[1] sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B, HW, C = 32, 196, 512
gy = torch.randn(B, C, device=dev); p = torch.zeros(1, device=dev); am = torch.randint(0, HW, (B, C), device=dev, dtype=torch.int32)
xm = torch.randn(B, C, device=dev); xa = torch.randn(B, C, device=dev); dx = torch.empty(B, HW, C, device=dev); dp = torch.zeros(1, device=dev)
print("mmpool_bwd", timeit(lambda: lib.mmpool_bwd(gy, p, am, xm, xa, B, HW, C, dx, dp)))
print("mmpool_bwd (no dp)", timeit(lambda: lib.mmpool_bwd(gy, p, am, xm, xa, B, HW, C, dx, None)))

from hifihr_amd import ops
from hifihr_amd.mano_tables import synthetic_mano_tables
h = ops.ManoLayerHandle(synthetic_mano_tables(0))
pose = (0.5 * torch.randn(B, 48, device=dev)).requires_grad_(True); beta = (0.5 * torch.randn(B, 10, device=dev)).requires_grad_(True)
verts, jtr = ops.mano_lbs(h, pose, beta)
gv = torch.randn_like(verts)
print("mano fwd+bwd (autograd)", timeit(lambda: torch.autograd.grad(ops.mano_lbs(h, pose, beta)[0], (pose, beta), gv)))
print("mano fwd only", timeit(lambda: ops.mano_lbs(h, pose, beta)))
