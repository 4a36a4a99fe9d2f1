#!/usr/bin/env python3
"""Aggregates the FETCH_SIZE / WRITE_SIZE passes of tools/kernel_traffic.sh: HBM bytes per launch of every hifihr kernel in the
last complete (eager) training step of the trace, keyed the way bench.py names its roofline lines."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest():
    h = hashlib.sha256()
    d = os.path.join(REPO, "hifihr_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("hifihr::", "")


per = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(f"{out}/{ctr}.csv")) if r["Counter_Name"] == ctr]
    key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else "Dispatch_ID"
    rows.sort(key=lambda r: int(r[key]))
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    agg = defaultdict(lambda: [0, 0.0])
    for r in step:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    per[ctr] = agg
traffic, detail = {}, {}
groups = defaultdict(lambda: [0, 0.0])
for k in per["FETCH_SIZE"]:
    n, f = per["FETCH_SIZE"][k]
    w = per["WRITE_SIZE"].get(k, [n, 0.0])[1]
    b = f * 1024 * 2 + w * 1024          # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE exact; both in KB
    if "at::" in k or "rocclr" in k:
        continue
    traffic[k] = b / n
    detail[k] = {"launches_per_step": n, "fetch_bytes_per_step": f * 2048, "write_bytes_per_step": w * 1024}
    fam = re.sub(r"<.*", "", k)          # bench.py groups conv_igemm_kernel / conv_wgrad_kernel by family
    groups[fam][0] += n; groups[fam][1] += b
for fam, (n, b) in groups.items():
    traffic.setdefault(fam, b / n)
# the launch-level keys of bench.py's HBM-bound lines (main kernel + its helpers)
# (names are kernel FAMILIES, i.e. the text before the template arguments; a family that matches nothing in the trace is an error:
# r04 kept summing a kernel that had been renamed and under-reported the forward rasteriser's traffic 7 x)
for key, names in (("render_fwd", ("render_fwd3_kernel", "render_vertex_kernel", "render_bin_kernel")),
                   ("render_bwd", ("render_bwd_kernel", "render_vertex_bwd_kernel"))):
    absent = [n for n in names if n not in groups]
    if absent:
        sys.exit(f"kernel_traffic_summary: {key}: no launch of {absent} in the traced step (renamed kernel?) -- have "
                 f"{sorted(g for g in groups if g.startswith('render'))}")
    tot = sum(groups[n][1] for n in names)
    traffic[key] = tot / max(groups[names[0]][0], 1)
whole = sum(v[1] for v in per["FETCH_SIZE"].values()) * 2048 + sum(v[1] for v in per["WRITE_SIZE"].values()) * 1024
print(json.dumps({"csrc_digest": digest(), "workload": "BASELINE configs[1] training step, B = 32, eager (one counter pass each)",
                  "collected_with": "rocprofv3 --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/kernel_traffic.sh); last complete step",
                  "correction": "gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
                  "traffic_bytes_per_launch": traffic, "detail": detail, "whole_step_traffic_bytes": whole}, indent=1))
