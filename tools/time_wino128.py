"""The 128-channel Winograd GEMM (T = 6272, C = K = 128, 16 batches) under the balanced-schedule tile variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
B, H, C, K = 32, 28, 128, 128
T = B * 14 * 14
V = torch.randn(16, T, C, device="cuda"); U = torch.randn(16, K, C, device="cuda"); M = torch.empty(16, T, K, device="cuda")
ws = torch.zeros(64 << 20, device="cuda")


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


ref = None
for var, minch in (("", ""), ("0", "1"), ("1", "1"), ("2", "1"), ("3", "1")):
    for k, v in (("HIFIHR_CONV_SK_VARIANT", var), ("HIFIHR_CONV_SK_MINCH", minch)):
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    t = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws))
    if ref is None: ref = M.clone()
    err = float((M - ref).abs().max())
    print(f"variant [{var or 'default'}] minch [{minch or 'default'}]: {t:6.1f} us  ({2.0 * 16 * T * C * K / t / 1e6:5.1f} TF)  max diff vs default {err:.2e}")
