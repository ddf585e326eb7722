#!/bin/bash
# usage (GPU box, repo root): tools/pmc_groups.sh <outdir> "<kbench args>" "<counters...>" ["<counters...>" ...]
# One rocprofv3 --pmc pass (kernel trace only) per counter group over tools/kbench.py; mean per dispatch of the encode kernels.
out=$1; shift
kargs=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $out
i=0
for grp in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $root/$out/pass$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 $kargs > $root/$out/pass$i.log 2>&1)
done
python3 tools/pmc_summary.py $out x3_encode
rm -rf $out/pass*/*.db $out/pass*/*/*.db
