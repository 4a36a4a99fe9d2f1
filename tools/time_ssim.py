"""Fused SSIM forward / backward at the step's size ([32,3,224,224]).  usage: python tools/time_ssim.py   (HIFIHR_SSIM_WGS_PER_CU=0: one
workgroup per tile, the round-4 form; default 3: persistent workgroups with the next tile's halo prefetched)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd import ops
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
gen = torch.Generator().manual_seed(0)
a = torch.rand(32, 3, 224, 224, generator=gen).cuda().requires_grad_(True)
b = (a.detach() * (torch.rand(32, 1, 224, 224, generator=gen).cuda() > 0.7)).contiguous()
def fwd():
    return ops.ssim_loss(a, b, 0.2)
v = fwd()
def both():
    a.grad = None
    fwd().backward()
tf = timeit(fwd); tb = timeit(both)
print(f"HIFIHR_SSIM_WGS_PER_CU={os.environ.get('HIFIHR_SSIM_WGS_PER_CU', '3')}: forward (kernel + finish) {tf:6.1f} us, forward + backward {tb:6.1f} us, value {float(v):.7f}")
