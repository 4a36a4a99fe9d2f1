#!/usr/bin/env python3
"""Steady-state per-step kernel summary from a rocprofv3 --kernel-trace CSV.

Steps are delimited by the fused Adam kernel (one launch per training step); only the last N complete steps
are aggregated, so MIOpen's solver search during warm-up does not pollute the numbers.
usage: tools/trace_summary.py <kernel_trace.csv> [N=5] [timeline.txt] > profiles/<name>.md
With a third argument the launches of the LAST step are also written in order (start offset, duration, idle gap before the launch).
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)_MI(\d+x\d+x\d+)", name)
        return f"rocBLAS/Tensile GEMM {name[:14]} MT{m.group(1)} MI{m.group(2)}" if m else name[:60]
    if "ck::" in name or name.startswith("_ZN2ck"):
        kind = "bwd_weight" if "bwd_weight" in name else "bwd_data" if "bwd_data" in name else "fwd" if "fwd" in name else "other"
        return f"CK conv/gemm ({kind})"
    return name[:90]


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(ends) < nsteps + 1:
        sys.exit(f"only {len(ends)} adam launches in trace")
    lo, hi = ends[-nsteps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
    agg = defaultdict(lambda: [0, 0.0, 0])
    busy = 0.0
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += d; a[2] = max(a[2], int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
        busy += d
    wall = (t1 - t0) / nsteps / 1e3
    print(f"# steady-state kernel summary: last {nsteps} training steps of `{path.split('/')[-1]}`\n")
    print(f"- wall per step (first kernel start to Adam end): {wall:.1f} us; sum of kernel durations per step: {busy / nsteps / 1e3:.1f} us; "
          f"kernel launches per step: {len(sel) / nsteps:.0f}\n")
    print("| kernel | launches/step | us/step | avg us | % of kernel time | regs |")
    print("|---|---|---|---|---|---|")
    for k, (n, d, regs) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"| `{k}` | {n / nsteps:.1f} | {d / nsteps / 1e3:.1f} | {d / n / 1e3:.1f} | {100 * d / busy:.1f} | {regs} |")


    if len(sys.argv) > 3:
        last = rows[ends[-2] + 1:ends[-1] + 1]
        base = int(last[0]["Start_Timestamp"])
        prev_end = base
        with open(sys.argv[3], "w") as f:
            f.write("# start_us  dur_us  gap_us  kernel   (last step of the trace; gap = idle time on the device before the launch)\n")
            for r in last:
                st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                f.write(f"{(st - base) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {max(0, st - prev_end) / 1e3:7.1f}  {short(r['Kernel_Name'])[:70]}"
                        f"  grid={r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}\n")
                prev_end = max(prev_end, en)


if __name__ == "__main__":
    main()
