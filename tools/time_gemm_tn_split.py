"""T-split of the TN products on the row-share kernel (round 5) against the per-tile kernels they ran on: the 1x1 backward-weight product
and the 36-problem products of the 128- / 256-channel F(4x4) layers.  usage: python tools/time_gemm_tn_split.py  (with and without
HIFIHR_GEMM_TN_SPLIT=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("HIFIHR_GEMM_TN_SPLIT =", os.environ.get("HIFIHR_GEMM_TN_SPLIT", "1"))
for (M, N, T, batch, what) in ((512, 256, 6272, 1, "1x1 wgrad 256->512 @14"), (128, 128, 1568, 36, "F(4x4) 128ch @28"), (256, 256, 480, 36, "F(4x4) 256ch @14"),
                               (256, 128, 6272, 1, "1x1 s2 wgrad-like"), (512, 512, 6272, 1, "1x1 512->512"), (1536, 384, 2368, 1, "effnet head wgrad B48"),
                               (128, 128, 2368 * 4, 1, "small")):
    A = torch.randn(batch, T, M, device="cuda"); Bm = torch.randn(batch, T, N, device="cuda")
    parts = lib.bgemm_tn_parts(M, N, T, batch)
    C = torch.empty(parts, batch, M, N, device="cuda")
    t = timeit(lambda: lib.bgemm_tn(A, Bm, C, M, N, T, batch, parts))
    ref = torch.matmul(A.transpose(1, 2), Bm)
    err = float((C.sum(0) - ref).abs().max() / ref.abs().max())
    gf = 2.0 * batch * T * M * N
    print(f"{what:26s} M={M:4d} N={N:4d} T={T:5d} x{batch:2d}: {lib.bgemm_describe(True, M, N, T, batch):30s} parts {parts:3d}: {t:6.1f} us ({gf / t / 1e6:5.1f} TF)  err {err:.1e}")
