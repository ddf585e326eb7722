#!/usr/bin/env python3
"""Kernel timeline of the round trip from a rocprofv3 --kernel-trace CSV: start, duration and the gap to the
previous kernel's end for the last few steps.  usage: timeline.py <dir with *_kernel_trace.csv> [n_rows]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); last_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if last_end is None else "%8.1f" % ((s - last_end) / 1e3)
    print("%10.1f us  dur %8.1f  gap-to-latest-end %8s  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
    last_end = e if last_end is None else max(last_end, e)
