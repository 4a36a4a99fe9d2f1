"""Sweep of tile shape / slab count for the backward-weight products of the Winograd F(4x4, 3x3) layers (36 problems of [T' x K]^T . [T' x C]).
GEMM time + the slab reduction (wino_dw_transform_parts) that follows it.  usage: python tools/time_gemm_tn_f4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def setenv(k, v):
    if v is None: os.environ.pop(k, None)
    else: os.environ[k] = str(v)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = 32
for H, C, K in ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)):
    T = B * ((H + 3) // 4) ** 2
    V = torch.randn(36, T, C, device="cuda"); Y = torch.randn(36, T, K, device="cuda"); dw = torch.zeros(K, 3, 3, C, device="cuda")
    gf = 2.0 * 36 * T * C * K / 1e9
    line = f"T'={T:5d} C={C:3d} K={K:3d} {gf:5.2f} GF |"
    for tile in (128128, 128064, 64128, 64064):
        for parts_req in (None, 1, 2, 4, 7):
            setenv("HIFIHR_GEMM_TN_TILE", tile); setenv("HIFIHR_GEMM_TN_PARTS", parts_req)
            parts = lib.bgemm_tn_parts(K, C, T, 36)
            if parts_req is not None and parts != parts_req: continue
            if parts * 36 * K * C * 4 > 2e9: continue
            dU = torch.empty(parts, 36, K, C, device="cuda")
            try:
                tg = timeit(lambda: lib.bgemm_tn(Y, V, dU, K, C, T, 36, parts))
            except Exception as e:
                continue
            td = timeit(lambda: lib.wino_dw_transform_parts(dU, parts, dw, K, C, 4))
            line += f" {tile // 1000}x{tile % 1000}/{'auto' if parts_req is None else 'p'}{parts}: {tg:5.1f}+{td:4.1f}={tg + td:5.1f}"
    setenv("HIFIHR_GEMM_TN_TILE", None); setenv("HIFIHR_GEMM_TN_PARTS", None)
    print(line, flush=True)
