#!/usr/bin/env python3
"""Registers / spills / scratch of every kernel of one csrc file, from the device assembly's metadata (no GPU needed).
usage: python tools/kernel_regs.py hifihr_amd/csrc/conv_halo.hip [name filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/_kregs.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out, src],
               check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{dem[:90]:92s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>7s}")
