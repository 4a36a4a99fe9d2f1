#!/usr/bin/env python3
"""Every convolution shape of a bench line (`roofline_kernels[*].shapes`) against ONE pass over its tensors at 5 TB/s: time per step above that
floor, sorted -- the table that showed VGG19's conv2_1 on the implicit GEMM (three 858 us launches for 92 us of tensors) and conv1_1's
backward-data re-reading its input nine times from L2 in round 4.  usage: python tools/shape_floor_table.py profiles/r04_bench_cfg3.json"""
import json, re, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.get("roofline_kernels", {}).items():
    for s in v["shapes"]:
        m = re.match(r"(\w+) N(\d+) (\d+)x(\d+) C(\d+)->K(\d+) (\d)x\d s(\d)", s["shape"])
        if not m:
            continue
        dirn, (N, H, W, C, K, R, st) = m.group(1), map(int, m.groups()[1:])
        OH = (H + (2 if R == 3 else 0) - R) // st + 1
        floor = (N * H * W * C + N * OH * OH * K) * 4 / 5e6                     # us: input + output once at 5 TB/s
        n = s["launches_per_step"]
        rows.append(((s["us"] - floor) * n, k, s["shape"], n, s["us"], floor, s["TFLOPs"]))
rows.sort(reverse=True)
print(f"{sys.argv[1]}: {d['ms_per_step']:.2f} ms/step; time above one 5 TB/s pass over the shape's tensors, per step")
for ex, k, shp, n, us, fl, tf in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{ex:8.1f} us  {k[:30]:30s} {shp:46s} x{n:.0f} {us:8.1f} us (floor {fl:6.1f}) {tf} TF")
print(f"sum over all {len(rows)} shapes: {sum(r[0] for r in rows):.0f} us")
