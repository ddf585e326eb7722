#!/bin/bash
# tools/r6/run_exp.sh OUT REPS name...  -- kbench with the placement probe over experiment builds in tools/scratch/exp/, interleaved
out=$1; reps=$2; shift 2
mkdir -p $(dirname $out); : > $out
for r in $(seq 1 $reps); do
  for v in "$@"; do
    echo -n "$v rep$r: " >> $out
    X3HIP_LIB=$PWD/tools/scratch/exp/libx3hip_$v.so timeout 300 python3 tools/kbench.py --steps 30 --place 4 $KBENCH_ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*//' >> $out
  done
done
cat $out
