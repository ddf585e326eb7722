#!/bin/bash
# step wall time / decode / check of kbench over check-kernel priority builds x workgroups per CU, fresh processes
# usage: tools/scratch/sweep_check.sh "hip hip_p0 hip_p3" "2 4 8" REPS
for v in $1; do for w in $2; do
  for r in $(seq 1 $3); do
    X3_WALL=1 X3HIP_LIB=$PWD/x3-rust_amd/lib/libx3$v.so python tools/kbench.py --steps 100 --opt check_wgs=$w 2>&1 | tail -1 | sed -e 's/ms; stream.*//' -e "s/^/$v wgs $w: /"
  done
done; done
