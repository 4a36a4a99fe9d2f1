#!/usr/bin/env python3
"""Aggregates the FETCH_SIZE / WRITE_SIZE passes of tools/conv_traffic.sh: HBM bytes of conv_igemm_kernel per launch and per step."""
import csv
import json
import sys

out = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{out}/{ctr}.csv")))
    rows = [r for r in rows if r["Counter_Name"] == ctr]
    # dispatches in launch order; a training step ends with adam_kernel: take the last complete step
    key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else "Dispatch_ID"
    rows.sort(key=lambda r: int(r[key]))
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    conv = [float(r["Counter_Value"]) for r in step if "conv_igemm_kernel" in r["Kernel_Name"]]
    allk = sum(float(r["Counter_Value"]) for r in step)
    res[ctr] = {"conv_igemm_launches": len(conv), "conv_igemm_KB_per_step": sum(conv), "all_kernels_KB_per_step": allk}
fetch = res["FETCH_SIZE"]["conv_igemm_KB_per_step"] * 1024 * 2          # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
write = res["WRITE_SIZE"]["conv_igemm_KB_per_step"] * 1024
n = res["FETCH_SIZE"]["conv_igemm_launches"]
print(json.dumps({
    "kernel": "hifihr::conv_igemm_kernel (all instantiations)", "workload": "BASELINE configs[1] training step, B = 32, eager",
    "collected_with": "rocprofv3 --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/conv_traffic.sh); last complete step",
    "raw": res,
    "correction": "gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
    "launches_per_step": n, "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
    "traffic_bytes_per_launch": (fetch + write) / max(n, 1),
    "whole_step_traffic_bytes": res["FETCH_SIZE"]["all_kernels_KB_per_step"] * 2048 + res["WRITE_SIZE"]["all_kernels_KB_per_step"] * 1024,
}, indent=1))
