mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
root=$PWD
date
python3 tools/kbench.py --steps 10 --opt decode_single=1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
python3 tools/kbench.py --steps 10 --opt decode_single=1 --samples 655360000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY -d $root/gpurun_out/r5f/p1 -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 --opt decode_single=1 > $root/gpurun_out/r5f/p1.log 2>&1)
python3 tools/pmc_summary.py gpurun_out/r5f x3_decode_fast
rm -rf gpurun_out/r5f/p1/*.db gpurun_out/r5f/p1/*/*.db
