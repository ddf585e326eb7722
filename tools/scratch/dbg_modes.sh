# what does a co-running kernel do to the decoder?  (profiling build: the check pass off, a do-nothing kernel beside it)
L=$PWD/x3-rust_amd/lib/libx3hip_prof.so
for cfg in "X3HIP_SPIN_WGS=0" "X3HIP_SPIN_WGS=1" "X3HIP_SPIN_WGS=4" "X3HIP_SPIN_WGS=8 X3HIP_SPIN_WAVES=2" "X3HIP_SPIN_WGS=4 X3HIP_SPIN_SLEEP=1" "X3HIP_SPIN_WGS=16 X3HIP_SPIN_SLEEP=1"; do
  echo "== $cfg"
  for i in 1 2 3; do env X3_NOCHECK=1 X3HIP_PROFILE_NO_CHECK=1 X3HIP_LIB=$L $cfg python tools/kbench.py --steps 30 | sed 's/sizes=0.000 scan=0.000 //; s/; stream.*//'; done
done
