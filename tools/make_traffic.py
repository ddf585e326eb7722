#!/usr/bin/env python3
"""Build profiles/traffic.json (HBM bytes per launch, per kernel) from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as MI355X_MICROARCH.md prescribes).

Corrections applied (MI355X_MICROARCH.md, section HBM):
  * both counters are reported in KiB -> x 1024;
  * on gfx950 FETCH_SIZE tallies each 128-byte request of a wide (16 B/lane) read as 64 bytes
    -> x 2 for our kernels, all of which read with 16-byte-per-lane loads;
  * WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
usage: make_traffic.py <pmc_fetch_dir> <pmc_write_dir> > profiles/traffic.json"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("x3_"):
        continue
    fb = fetch.get(k, 0.0) * 1024 * 2
    wb = write.get(k, 0.0) * 1024
    out[k] = {"hbm_bytes_per_launch": int(fb + wb), "fetch_bytes": int(fb), "write_bytes": int(wb),
              "note": "FETCH_SIZE KiB x1024 x2 (gfx950 128B-request correction) + WRITE_SIZE KiB x1024"}
print(json.dumps(out, indent=1))
