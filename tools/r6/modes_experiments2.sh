#!/bin/bash
# more of modes_experiments.sh: what the check kernel beside the decoder does to it, by how the check kernel is run
out=${1:-gpurun_out/r6/modes_experiments2.txt}
run() { echo -n "$LABEL $*: "; python3 tools/kbench.py --steps 20 "$@" 2>&1 | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*rep 0//'; }
{
for rep in 1 2; do
LABEL="base S" run
LABEL="base F" run --out-shift 4096
LABEL="base pad 2474" run --pad 2474
LABEL="base pad 1237" run --pad 1237
done
for w in 1 2 8; do
export X3HIP_CHECK_WGS=$w
LABEL="check wgs/CU $w S" run
LABEL="check wgs/CU $w F" run --out-shift 4096
done
unset X3HIP_CHECK_WGS
export X3HIP_CHECK_FIRST=1
LABEL="check enqueued first S" run
LABEL="check enqueued first F" run --out-shift 4096
unset X3HIP_CHECK_FIRST
export X3HIP_CHECK_MAIN=1
LABEL="streams swapped S" run
LABEL="streams swapped F" run --out-shift 4096
unset X3HIP_CHECK_MAIN
for pr in 0 2 3; do
export X3HIP_CHECK_PRIO=$pr
LABEL="check prio $pr S" run
LABEL="check prio $pr F" run --out-shift 4096
done
unset X3HIP_CHECK_PRIO
LABEL="decode twice per step? (decode-only) S" run --decode-only
LABEL="white noise S" run --kind 1
LABEL="white noise F" run --kind 1 --out-shift 4096
LABEL="zeros S" run --kind 0
LABEL="zeros F" run --kind 0 --out-shift 4096
} 2>&1 | tee $out
