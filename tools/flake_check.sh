#!/bin/bash
# The full -m gpu suite N times, each in a fresh process, on ONE box: a tolerance that sits inside float-atomic
# noise passes most of the time -- one green run says little.  Writes gpurun_out/flake_check/run<i>.log and a summary with the pass /
# fail counts of every run and the smallest margins the tests printed ("[margin] ...: observed/bound r").
# usage (on the GPU box): tools/flake_check.sh [runs=3]
set -u
cd "$(dirname "$0")/.."
N=${1:-3}
OUT=gpurun_out/flake_check
mkdir -p "$OUT"
rc_all=0
# pytest caches its rewritten test modules (tests/__pycache__/*-pytest-*.pyc) without regard to enable_assertion_pass_hook: start clean,
# and leave nothing behind that would make a later plain run pay for the hook
find tests -name "*-pytest-*.pyc" -delete
for i in $(seq 1 "$N"); do
  python -m pytest tests -m gpu -q -s -p no:cacheprovider -o enable_assertion_pass_hook=true > "$OUT/run$i.log" 2>&1
  rc=$?
  cp gpurun_out/margins.txt "$OUT/margins_run$i.txt" 2>/dev/null
  [ $rc -ne 0 ] && rc_all=1
  echo "run $i: rc $rc: $(grep -E '^[0-9]+ (passed|failed)|passed|failed' "$OUT/run$i.log" | tail -n 1)"
done > "$OUT/summary.txt"
{
  echo
  echo "failures over all runs:"
  grep -hE "^FAILED|^ERROR" "$OUT"/run*.log | sort | uniq -c || true
  echo
  echo "largest observed/bound per assertion tag over all runs (1.0 = at the bound):"
  grep -h "^\[margin\]" "$OUT"/run*.log | sed -E 's/^\[margin\] (.*): observed ([^ ]+) +bound ([^ ]+) +observed\/bound ([^ ]+)$/\4\t\1/' \
    | sort -t$'\t' -k2,2 -k1,1gr | awk -F'\t' '!seen[$2]++' | sort -gr | head -n 60
} >> "$OUT/summary.txt"
{
  echo
  echo "passing inequality assertions closest to their bounds, worst over all runs (tests/conftest.py pytest_assertion_pass):"
  cat "$OUT"/margins_run*.txt 2>/dev/null | grep -E "^ +[0-9]" | sort -k6 -k1,1gr | awk '{k=$0; sub(/^ +[0-9.]+ +[^ ]+ vs [^ ]+ +/, "", k); if (!(k in seen)) {seen[k]=1; print}}' | sort -gr | head -n 40
} >> "$OUT/summary.txt"
find tests -name "*-pytest-*.pyc" -delete
cat "$OUT/summary.txt"
exit $rc_all
