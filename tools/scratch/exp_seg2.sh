mkdir -p gpurun_out/r5g
o=gpurun_out/r5g/seg_lds.txt
date > $o
for pad in 0 9000; do
for seg in 0 64 128; do
  echo -n "pad=$pad seg=$seg: " >> $o
  X3HIP_DECODE_DYN_LDS=$pad timeout 300 python3 tools/kbench.py --steps 20 --seg $seg 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110 >> $o
done
done
cat $o
