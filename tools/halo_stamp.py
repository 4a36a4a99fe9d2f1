#!/usr/bin/env python3
"""In-kernel cycle stamps of conv_halo_kernel (tools/_probe/libhifihr_halo_stamp.so, built by tools/build_halo_probe.sh): cycles per
tap of an MFMA wave (ideal: 112 MFMAs x 32 = 3584 on a 7-block tile), time at the per-chunk barrier, epilogue, loader waits."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_halo_stamp.so"))
read = lib.c.hifihr_halo_stamp_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 56
x = torch.randn(B, H, W, 64, device="cuda"); w = torch.randn(64, 3, 3, 64, device="cuda") / 24; out = torch.empty_like(x)
fn = lambda: lib.conv2d_fwd(x, w, None, out, B, H, W, 64, 64, 3, 3, 1, 1)
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
print(f"B={B}: {us:.1f} us/launch; {wgs // n} workgroups; {v[0] / ch:.0f} cycles per tap in the loops (ideal 3584 on a full tile), {v[3] / ch:.0f} of them at "
      f"the barrier; clock {mhz:.0f} MHz; per workgroup: entry -> exit {v[5] / wgs:.0f} cycles = {v[5] / wgs / mhz:.1f} us, chunk loops {v[0] / wgs:.0f}, "
      f"epilogues {v[6] / wgs:.0f}; loader wave 0 waits on vmcnt {v[7] / wgs:.0f} cycles per workgroup")
