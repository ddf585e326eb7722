#!/bin/bash
# tools/run_variants.sh OUT STEPS REPEATS name...   -- kbench over variant libraries (tools/variants.py), interleaved so
# that drift of the box hits all of them alike.  One line per run in OUT.
out=$1; steps=$2; reps=$3; shift 3
mkdir -p $(dirname $out)
: > $out
for r in $(seq 1 $reps); do
  for v in "$@"; do
    echo -n "$v rep$r: " >> $out
    X3HIP_LIB=$PWD/x3-rust_amd/lib/variants/libx3hip_$v.so timeout 300 python3 tools/kbench.py --steps $steps $KBENCH_ARGS 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
  done
done
cat $out
