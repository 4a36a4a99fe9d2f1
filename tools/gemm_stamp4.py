#!/usr/bin/env python3
"""In-kernel cycle stamps of the row-share GEMM at the step's F(4x4) shapes (tools/_probe/libhifihr_gemm_stamp.so, built by
tools/build_gemm_probe.sh): where a workgroup's life goes -- entry -> first barrier, chunk loops, barriers, epilogues.
NT alone (bgemm_nt_rows_kernel<0>) and TN alone (bgemm_tn_rows_kernel); the pair launch runs the same two bodies side by side."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_gemm_stamp.so"))
read = lib.c.hifihr_gemm_stamp_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B = 32
for kind, H, C, K in (("nt", 28, 128, 128), ("nt", 14, 256, 256), ("nt", 14, 512, 512)):
    T = lib.wino_tiles(B, H, H, 4)
    V = torch.randn(36, T, C, device="cuda"); U = torch.randn(36, K, C, device="cuda"); M = torch.empty(36, T, K, device="cuda")
    Y = torch.randn(36, T, K, device="cuda")
    parts = lib.wino_wgrad_parts(B, H, H, C, K, 4)
    dUp = torch.empty(max(parts, 1), 36, K, C, device="cuda")
    nb = lib.wino_gemm_workspace_bytes(B, H, H, C, K, 4)
    ws = torch.zeros(max(nb, 4) // 4 + 64, device="cuda")
    fn = (lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws, m=4)) if kind == "nt" else (lambda: lib.wino_wgrad_gemm_parts(V, Y, dUp, B, H, H, C, K, parts, m=4))
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
        print(f"{kind} H={H} C={C} K={K} T={T}: {us:.1f} us/launch; no stamps (raw {v})", flush=True)
        continue
    mhz = v[0] / max(1, v[1]) * 100
    print(f"{kind} H={H} {C}->{K} T={T} parts={parts}: {us:.1f} us/launch; wave 0 of {waves // n} workgroups: {ch / waves:.1f} chunks each, "
          f"{v[0] / ch:.0f} cycles per chunk (ideal 4096), {v[3] / ch:.0f} of them at the barrier; clock {mhz:.0f} MHz; "
          f"entry -> exit {v[5] / waves:.0f} cycles = {v[5] / waves / mhz:.1f} us: chunk loops {v[0] / waves:.0f}, epilogues {v[6] / waves:.0f}, "
          f"entry -> first barrier {v[7] / waves:.0f}, rest {(v[5] - v[0] - v[6] - v[7]) / waves:.0f}", flush=True)
