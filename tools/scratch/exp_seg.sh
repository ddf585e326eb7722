mkdir -p gpurun_out/r5g
o=gpurun_out/r5g/seg.txt
date > $o
python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 >> $o
for seg in 0 64 128 252; do
 for n in 691200000 26460000; do
  echo -n "seg=$seg n=$n: " >> $o
  timeout 300 python3 tools/kbench.py --steps 20 --samples $n --seg $seg 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110 >> $o
 done
done
cat $o
