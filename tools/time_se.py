"""The squeeze-excite layers of EfficientNet-b3 at batch 48: the two head-kernel launches per direction (round 3) against the fused
kernels (csrc/se.hip se_mlp_*), us per block shape and for the 26 blocks of a step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.effnet import b3_block_table
lib = get_lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = int(os.environ.get("B", "48"))
tot = {"old_fwd": 0.0, "new_fwd": 0.0, "old_bwd": 0.0, "new_bwd": 0.0}
seen = {}
for (k, s, e, i, o) in b3_block_table():
    C, SQ = i * e, max(1, int(i * 0.25))
    if (C, SQ) not in seen:
        dev = "cuda"
        mean = torch.randn(B, C, device=dev); w1 = torch.randn(SQ, C, device=dev) / C ** 0.5; b1 = torch.zeros(SQ, device=dev)
        w2 = torch.randn(C, SQ, device=dev) / SQ ** 0.5; b2 = torch.zeros(C, device=dev); w2t = w2.t().contiguous()
        h1, z1, gate = torch.empty(B, SQ, device=dev), torch.empty(B, SQ, device=dev), torch.empty(B, C, device=dev)
        mo = torch.empty(B, C, device=dev); acc = torch.randn(B, C, device=dev)
        dgate = torch.randn(B, C, device=dev); dz2, dz1, dh1, dmean = torch.empty(B, C, device=dev), torch.empty(B, SQ, device=dev), torch.empty(B, SQ, device=dev), torch.empty(B, C, device=dev)
        dw1, db1, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
        r = {}
        r["old_fwd"] = timeit(lambda: (lib.linear_fwd(mean, w1, b1, 2, h1, z=z1), lib.linear_fwd(h1, w2, b2, 3, gate)))
        r["new_fwd"] = timeit(lambda: lib.se_mlp_fwd(acc, w1, b1, w2t, b2, B, C, SQ, mo, z1, h1, gate))
        r["old_bwd"] = timeit(lambda: (lib.linear_bwd(dgate, gate, h1, w2, 3, dz2, dw2, db2, dh1), lib.linear_bwd(dh1, None, mean, w1, 2, dz1, dw1, db1, dmean, z=z1)))
        r["new_bwd"] = timeit(lambda: lib.se_mlp_bwd(dgate, gate, z1, h1, mean, w1, w2t, B, C, SQ, dz2, dz1, dmean, dw1, db1, dw2, db2))
        seen[(C, SQ)] = r
        print(f"C = {C:5d} SQ = {SQ:3d}: " + "  ".join(f"{k} {v:6.1f}" for k, v in r.items()))
    for k2 in tot: tot[k2] += seen[(C, SQ)][k2]
print(f"26 blocks, B = {B}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in tot.items()))
