#!/usr/bin/env python3
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    d = by.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(by.values())[3:]
keys = [k for k in disp[0] if k != "dur"]
print(f"n={len(disp)} dur={sum(d['dur'] for d in disp) / len(disp):.1f}us " + " ".join(f"{k}={sum(d[k] for d in disp) / len(disp):.4g}" for k in keys))
