#!/bin/bash
# What selects the decode phase's pace?  Every line a fresh process; S = the default allocation (slow on every box so far),
# F = the stream 4 096 bytes into its allocation (fast).  tools/scratch/exp/: -DX3_PROFILING and -DX3S_PACE_OFF=1 builds --
#   JOBS=3 python3 tools/variants.py prof="-DX3_PROFILING" nopace="-DX3S_PACE_OFF=1"; mkdir -p tools/scratch/exp;
#   cp x3-rust_amd/lib/variants/libx3hip_{prof,nopace}.so tools/scratch/exp/   (lib/variants/ itself does not travel with gpurun)
out=${1:-gpurun_out/r6/modes_experiments.txt}
run() { echo -n "$LABEL $*: "; python3 tools/kbench.py --steps 20 "$@" 2>&1 | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*rep 0//'; }
{
for rep in 1 2; do
LABEL="base S" run
LABEL="base F" run --out-shift 4096
done
LABEL="decode-only S" run --decode-only
LABEL="decode-only F" run --decode-only --out-shift 4096
export X3HIP_LIB=$PWD/tools/scratch/exp/libx3hip_nopace.so
LABEL="no pacing S" run
LABEL="no pacing F" run --out-shift 4096
export X3HIP_LIB=$PWD/tools/scratch/exp/libx3hip_prof.so
LABEL="profiling build S" run
LABEL="profiling build F" run --out-shift 4096
export X3HIP_PROFILE_NO_CHECK=1 X3_NOCHECK=1
LABEL="no check kernel S" run
LABEL="no check kernel F" run --out-shift 4096
unset X3HIP_PROFILE_NO_CHECK X3_NOCHECK
export X3HIP_CHECK_SERIAL=1
LABEL="check in front, same stream S" run
LABEL="check in front, same stream F" run --out-shift 4096
unset X3HIP_CHECK_SERIAL X3HIP_LIB
export X3HIP_DECODE_DYN_LDS=8192
LABEL="four groups per CU S" run
LABEL="four groups per CU F" run --out-shift 4096
unset X3HIP_DECODE_DYN_LDS
} 2>&1 | tee $out
