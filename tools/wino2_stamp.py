#!/usr/bin/env python3
"""In-kernel cycle stamps of conv_wino2_kernel (tools/_probe/libhifihr_halo_stamp.so, built by tools/build_halo_probe.sh): cycles per
stage (one transform position: 32 MFMAs x 32 = 1024 on a full tile) of MFMA wave 0, time at the per-stage barrier, epilogue, loader waits."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.environ.get("STAMP_LIB") or os.path.join(R, "tools", "_probe", "libhifihr_halo_stamp.so"))
read = lib.c.hifihr_halo_stamp_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
for (B, H, W) in ((32, 56, 56), (48, 224, 224)):
    x = torch.randn(B, H, W, 64, device="cuda"); w = torch.randn(64, 3, 3, 64, device="cuda") / 24; out = torch.empty_like(x)
    U = torch.empty(16 * 64 * 64, device="cuda"); lib.wino_weight_transform(w, U, 64, 64, 0)
    fn = lambda: lib.conv3x3_c64_wino(x, U, None, False, out, None, B, H, W)
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
    ch, wgs = max(1, v[2]), max(1, v[4])
    mhz = v[0] / max(1, v[1]) * 100
    print(f"B={B} {H}x{W}: {us:.1f} us/launch; {wgs // n} workgroups; {v[0] / ch:.0f} cycles per stage in the loops (ideal 2048 on a full tile: 64 MFMAs), {v[3] / ch:.0f} of "
          f"them at the barrier; clock {mhz:.0f} MHz; per workgroup: entry -> exit {v[5] / wgs:.0f} cycles = {v[5] / wgs / mhz:.1f} us, stage loops {v[0] / wgs:.0f}, "
          f"epilogues {v[6] / wgs:.0f}; loader wave 0 waits on vmcnt {v[7] / wgs:.0f} cycles per workgroup")
