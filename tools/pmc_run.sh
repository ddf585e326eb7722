#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir> "<counters...>" ["<counters...>" ...]
# One rocprofv3 --pmc pass per counter group over tools/kbench.py (kernel trace only, as the pool requires).
out=$1; shift
export TMPDIR=/tmp
root=$PWD
i=0
for grp in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $root/$out/pass$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 > $root/$out/pass$i.log 2>&1)
done
python3 tools/pmc_summary.py $out x3_encode_stream x3_decode_fast x3_frame_check
