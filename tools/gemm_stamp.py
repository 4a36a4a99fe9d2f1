#!/usr/bin/env python3
"""In-kernel cycle stamps of the wave-specialised GEMM (tools/_probe/libhifihr_gemm_stamp.so): cycles per 32-deep chunk of an MFMA
wave (ideal: 128 MFMAs x 32 = 4096), the clock the chip holds under this load, time at the per-chunk barrier."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_gemm_stamp.so"))
read = lib.c.hifihr_gemm_stamp_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B = 32
for ws_ in (4,):
    os.environ["HIFIHR_GEMM_WS"] = str(ws_)
    for kind, H, C, K in (("nt", 14, 512, 512), ("tn", 14, 512, 512), ("nt", 14, 256, 256), ("tn", 14, 256, 256), ("nt", 28, 128, 128)):
        T = B * (H // 2) ** 2
        V = torch.randn(16, T, C, device="cuda"); U = torch.randn(16, K, C, device="cuda"); M = torch.empty(16, T, K, device="cuda")
        Y = torch.randn(16, T, K, device="cuda")
        os.environ["HIFIHR_GEMM_TN_PARTS"] = "1"
        parts = lib.wino_wgrad_parts(B, H, H, C, K)
        dUp = torch.empty(parts, 16, K, C, device="cuda")
        fn = (lambda: lib.wino_gemm(V, U, M, B, H, H, C, K)) if kind == "nt" else (lambda: lib.wino_wgrad_gemm_parts(V, Y, dUp, B, H, H, C, K, parts))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        read(buf, 1)
        n = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        read(buf, 1)
        v = list(buf)
        us = e0.elapsed_time(e1) / n * 1e3
        ch, waves = max(1, v[2]), max(1, v[4])
        if v[0] == 0:
            print(f'ws{ws_} {kind} H={H} C={C} K={K}: {us:.1f} us/launch; no stamps (raw {v}); kernel: {lib.bgemm_describe(kind == "tn", T if kind == "nt" else K, K if kind == "nt" else C, C if kind == "nt" else T)}', flush=True)
            continue
        print(f"ws{ws_} {kind} H={H} C={C} K={K}: {us:.1f} us/launch ({2.0 * 16 * T * C * K / us / 1e6:.1f} TF); wave 0 of {waves // n} workgroups: "
              f"{v[0] / ch:.0f} cycles per chunk (ideal 4096), {v[3] / ch:.0f} of them at the barrier; clock {v[0] / max(1, v[1]) * 100:.0f} MHz; "
              f"kernel entry -> end {v[5] / waves:.0f} cycles = {v[5] / waves / (v[0] / max(1, v[1]) * 100):.1f} us, main loop {v[0] / waves:.0f}"
              + (f"; rows kernel: epilogues {v[6] / waves:.0f} cycles, entry -> first barrier {v[7] / waves:.0f}" if kind == "nt" else ""), flush=True)
