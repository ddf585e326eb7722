mkdir -p gpurun_out/r5k
o=gpurun_out/r5k/dense_ahead.txt
date > $o
for rep in 1 2; do
for v in dense0 dense1; do
 for args in "--kind 2" "--loud 0.01" "--loud 0.1" "--kind 1" "--kind 3"; do
  echo -n "$v $args: " >> $o
  X3HIP_LIB=$PWD/x3-rust_amd/lib/variants/libx3hip_$v.so timeout 300 python3 tools/kbench.py --steps 30 $args 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-105 >> $o
 done
done
done
cat $o
