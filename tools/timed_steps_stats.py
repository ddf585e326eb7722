#!/usr/bin/env python3
"""rocprofv3's --stats averages every launch of a kernel in the process; since round 6 bench.py runs its placement probe first
(64 pairs x 13 round trips on buffers that are mostly NOT the ones it keeps), so the process-wide average no longer is the
timed steps' average.  This reads the kernel trace of the same run and averages each kernel over its LAST K launches -- the
timed steps -- and prints them beside what the run's own JSON line said (HIP events).
usage: timed_steps_stats.py <dir with *_kernel_trace.csv> <K> [<log whose last line is the bench line>]"""
import csv, glob, json, os, sys
from collections import defaultdict
d, K = sys.argv[1], int(sys.argv[2])
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
line = None
if len(sys.argv) > 3:
    try:
        line = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    except Exception:
        line = None
print("%-44s %8s %12s %14s" % ("kernel", "launches", "all: avg ms", "last %d: avg ms" % K))
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if not k.startswith(("x3_", "void x3_")) or len(v) < K:
        continue
    v.sort()
    a = [x[1] for x in v]
    print("%-44s %8d %12.4f %14.4f" % (k[:44], len(a), sum(a) / len(a), sum(a[-K:]) / K))
if line:
    print("the same run's JSON line (HIP events on the timed steps): kernels_ms =", json.dumps(line.get("kernels_ms")),
          " placement =", json.dumps((line.get("placement") or {}).get("step_ms")))
