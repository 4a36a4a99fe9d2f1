#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv of conv_ablate_pmc.py: per variant (groups of 10 dispatches of
conv_igemm_kernel in order) the mean counter values and mean duration."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_igemm" in r["Kernel_Name"]]
by_disp = collections.OrderedDict()
for r in rows:
    d = by_disp.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(by_disp.values())
names = ["prod", "p1", "p2", "p3", "p4"]
for i, n in enumerate(names):
    grp = disp[i * 10 + 3:(i + 1) * 10]           # skip 3 warm launches of each variant
    if not grp: break
    keys = [k for k in grp[0] if k != "dur"]
    m = {k: sum(g[k] for g in grp) / len(grp) for k in ["dur"] + keys}
    extra = ""
    if "GRBM_GUI_ACTIVE" in m:
        extra = f"  clock={m['GRBM_GUI_ACTIVE'] / 8 / m['dur'] / 1e3:.2f} GHz"
    print(f"{n:5s} dur={m['dur']:7.1f}us " + " ".join(f"{k}={m[k]:.4g}" for k in keys) + extra)
