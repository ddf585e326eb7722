#!/bin/bash
# The check kernel's packed look-ups (X3_CHECK_STEP_ASM): A/B on config 3 and on white noise, then the suite and the fuzz
# families that damage CRCs on the main build.
out=gpurun_out/r5m; mkdir -p $out
bash tools/run_variants.sh $out/chk_cfg3.txt 100 3 chk0 chk1
KBENCH_ARGS="--kind 1" bash tools/run_variants.sh $out/chk_white.txt 60 3 chk0 chk1
KBENCH_ARGS="--loud 0.01" bash tools/run_variants.sh $out/chk_mixed.txt 60 2 chk0 chk1
cut -c1-150 $out/chk_cfg3.txt $out/chk_white.txt $out/chk_mixed.txt
python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4 | tee $out/tests.txt
timeout 600 python3 tools/fuzz_parity.py --seed 571 --minutes 4 --families deagm 2>&1 | grep -v amdgpu.ids | tail -2 | tee $out/fuzz.txt
