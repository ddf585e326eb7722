#!/bin/bash
# decode / check time of fresh processes against the check kernel's workgroups per CU (X3HIP_CHECK_WGS), interleaved
out=${1:-gpurun_out/r6/modes_wgs_stats.txt}
reps=${2:-8}
{
for r in $(seq 1 $reps); do
  for w in ${WGS:-2 3 4}; do
    echo -n "wgs $w rep $r: "
    X3HIP_CHECK_WGS=$w python3 tools/kbench.py --steps 20 2>&1 | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*//'
  done
done
} 2>&1 | tee $out
python3 - $out <<'PY'
import re, sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"wgs (\d+) rep \d+: encode=([\d.]+) decode=([\d.]+) check=([\d.]+)", l)
    if m: acc[int(m.group(1))].append((float(m.group(3)), float(m.group(4))))
for w, v in sorted(acc.items()):
    d = sorted(x[0] for x in v); c = sorted(x[1] for x in v); ph = sorted(max(x) for x in v)
    print("wgs %d: decode min %.3f median %.3f max %.3f | check min %.3f max %.3f | phase (max of both) min %.3f median %.3f max %.3f  (n=%d)" %
          (w, d[0], d[len(d) // 2], d[-1], c[0], c[-1], ph[0], ph[len(ph) // 2], ph[-1], len(v)))
PY
