"""per-dispatch UTCL1 counters of the split decoder, grouped the way mode_scan runs them"""
import csv, glob, sys, collections
d = sys.argv[1]; per = int(sys.argv[2])
rows = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_split" not in r["Kernel_Name"]: continue
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
for i in range(0, len(ids), per):
    chunk = ids[i:i + per]
    names = sorted(rows[chunk[0]])
    print(i // per, " ".join("%s=%.3e" % (k, sum(rows[j].get(k, 0) for j in chunk) / len(chunk)) for k in names))
