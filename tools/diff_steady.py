"""Per-kernel difference of two steady-state tables of tools/trace_summary.py: python tools/diff_steady.py old.md new.md"""
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'\| `(.+?)` \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m:
            d[m.group(1)[:80]] = (float(m.group(2)), float(m.group(3)))
        m2 = re.match(r'- wall per step.*?: ([\d.]+) us; sum of kernel durations per step: ([\d.]+) us; kernel launches per step: (\d+)', l)
        if m2:
            d["__total__"] = (float(m2.group(3)), float(m2.group(2)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in sorted(set(a) | set(b)):
    ca, ua = a.get(k, (0, 0.0)); cb, ub = b.get(k, (0, 0.0))
    rows.append((ub - ua, k, ca, ua, cb, ub))
rows.sort()
print(f"{'kernel':80s} {'old n':>6s} {'old us':>9s} {'new n':>6s} {'new us':>9s} {'delta':>8s}")
for d, k, ca, ua, cb, ub in rows:
    if abs(d) >= 2.0 or k == "__total__":
        print(f"{k:80s} {ca:6.0f} {ua:9.1f} {cb:6.0f} {ub:9.1f} {d:+8.1f}")
