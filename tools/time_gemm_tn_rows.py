"""Backward-weight products of the F(4x4, 3x3) layers (36 problems of [T' x K]^T . [T' x C]) on the default path: bgemm_tn_rows_kernel where it
applies (HIFIHR_GEMM_TN_ROWS=0: the per-tile kernels with T-split slabs), GEMM + the dw transform that follows it.
usage: python tools/time_gemm_tn_rows.py   (run twice, with and without HIFIHR_GEMM_TN_ROWS=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = 32
print("HIFIHR_GEMM_TN_ROWS =", os.environ.get("HIFIHR_GEMM_TN_ROWS", "1"))
for H, C, K in ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 256), (14, 512, 512)):
    T = B * ((H + 3) // 4) ** 2
    V = torch.randn(36, T, C, device="cuda"); Y = torch.randn(36, T, K, device="cuda"); dw = torch.zeros(K, 3, 3, C, device="cuda")
    parts = lib.bgemm_tn_parts(K, C, T, 36)
    dU = torch.empty(parts, 36, K, C, device="cuda")
    tg = timeit(lambda: lib.bgemm_tn(Y, V, dU, K, C, T, 36, parts))
    td = timeit(lambda: lib.wino_dw_transform_parts(dU, parts, dw, K, C, 4))
    gf = 2.0 * 36 * T * C * K
    print(f"T'={T:5d} C={C:3d} K={K:3d}: {lib.bgemm_describe(True, K, C, T, 36):28s} parts {parts}: {tg:6.1f} us ({gf / tg / 1e6:5.1f} TF) + dw {td:5.1f} us")
