#!/usr/bin/env python3
"""Per-phase cycle breakdown of the balanced conv main loop from the stamped diagnostic build (tools/_probe/libhifihr_stamp.so)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_stamp.so"))
read = lib.c.hifihr_conv_stamp_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B = 32
ws = torch.zeros(64 << 20, device="cuda")
for H, C, K in [(28, 128, 128), (14, 256, 256), (14, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(B, H, H, K, device="cuda")
    for _ in range(3):
        lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    read(buf, 1)
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.conv2d_fwd(x, w, None, y, B, H, H, C, K, 3, 3, 1, 1, ws=ws)
    e1.record(); torch.cuda.synchronize()
    read(buf, 1)
    v = list(buf)
    ch = max(1, v[4])
    names = ["load issue", "lds read + 16 mfma", "vmcnt wait + lds store", "barrier"]
    tot = sum(v[:4]) / ch
    print(f"H={H} C={C} K={K}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/launch; per (wave, chunk) memtime ticks (100 MHz? or core clk): total {tot:.0f} = " +
          ", ".join(f"{nm} {v[i] / ch:.0f} ({100 * v[i] / sum(v[:4]):.0f}%)" for i, nm in enumerate(names)))
