#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel dispatch.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [kernel-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if filt and not any(s in k for s in filt):
                continue
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-28s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
