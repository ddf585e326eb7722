#!/bin/bash
# decode time against what was allocated first (tools/kbench.py --pad KiB): the addresses of wav / out / back move
out=${1:-gpurun_out/r6/modes}
mkdir -p $out
for pad in 0 1237 2474 3711 4948 6185 2048 4096 8192 1024 512 3072 16384 65536 1048576; do
  echo -n "pad $pad KiB: "
  python3 tools/kbench.py --steps 20 --pad $pad 2>&1 | tail -1 | sed -e 's/sizes=.*check=/check=/' -e 's/dense=.*rep 0//'
done | tee $out/pad_sweep.txt
