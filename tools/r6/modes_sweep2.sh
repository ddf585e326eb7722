#!/bin/bash
# decode / check time against (a) the order the buffers are allocated in, (b) where in its allocation the stream begins,
# (c) where the decoded samples begin -- all with nothing allocated in front (the slow mode of modes_pad_sweep.sh)
out=${1:-gpurun_out/r6/modes}
mkdir -p $out
run() { echo -n "$*: "; python3 tools/kbench.py --steps 20 "$@" 2>&1 | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*rep 0//'; }
{
run
run --order owfb
run --order bowf
run --order fbow
run --order wbfo
run --out-shift 4096
run --out-shift 65536
run --out-shift 1048576
run --out-shift 2097152
run --out-shift 6291456
run --back-shift 4096
run --back-shift 65536
run --back-shift 2097152
run --shift 2097152
run --pad 2048
run --pad 2048 --out-shift 2097152
run --pad 2048 --back-shift 2097152
} | tee $out/sweep2.txt
