#!/bin/bash
# usage (GPU box, repo root): tools/pmc_enc.sh <outdir> [kbench args...]   -- SQ counters of the kernels of one kbench run
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $root/$out/p1 -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 "$@" > $root/$out/p1.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT -d $root/$out/p2 -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 "$@" > $root/$out/p2.log 2>&1)
python3 tools/pmc_summary.py $out x3_encode x3_decode_split
rm -rf $out/p*/*.db $out/p*/*/*.db
