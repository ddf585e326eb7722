#!/bin/bash
# Kernel timeline of bench.py's steps: what sits between the encoder's and the decoder's kernels.
out=gpurun_out/r5p; mkdir -p $out; root=$PWD; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d $root/$out/tl -o tl --output-format csv -- python3 $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-measure-traffic --no-extras --no-configs > $root/$out/tl.log 2>&1)
python3 tools/timeline.py $out/tl 40 | tee $out/timeline.txt
rm -rf $out/tl/*.db
