#!/bin/bash
# The two paces of the decode phase under the counters: address translation (UTCL1 / UTCL2) and the L2's traffic, for the
# allocation that runs slow (default order) and one that runs fast (the stream 4 096 bytes into its allocation).
out=${1:-gpurun_out/r6/modes_pmc}
export TMPDIR=/tmp
root=$PWD
mkdir -p $out
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  for mode in slow fast; do
    extra=""; [ $mode = fast ] && extra="--out-shift 4096"
    (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $root/$out/${mode}$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 $extra > $root/$out/${mode}$i.log 2>&1)
    tail -1 $out/${mode}$i.log | cut -c1-80
  done
done
for mode in slow fast; do
  echo "== $mode"
  for j in 1 2 3 4; do python3 tools/pmc_summary.py $out/${mode}$j x3_decode_split x3_frame_check; done
  python3 - $out $mode <<'PY'
import csv, glob, sys, collections
d, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob("%s/%s1/**/*kernel_trace.csv" % (d, mode), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(acc.items()):
    print("   duration %-40s mean %.4f ms (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
rm -rf $out/*/*.db $out/*/*/*.db
