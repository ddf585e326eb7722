mkdir -p gpurun_out/r5b
o=gpurun_out/r5b/rounds.txt
: > $o
for v in base paceoff; do
 for n in 691200000 1382400000 2764800000; do
  echo -n "$v n=$n: " >> $o
  X3HIP_LIB=$PWD/x3-rust_amd/lib/variants/libx3hip_$v.so timeout 300 python3 tools/kbench.py --steps 20 --samples $n 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150 >> $o
 done
done
cat $o
