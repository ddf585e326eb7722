#!/bin/bash
# usage (GPU box, repo root): tools/pmc_l2req.sh <outdir>  -- L2 request counts per kernel (one rocprofv3 --pmc pass per group)
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $out
i=0
for grp in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $root/$out/pass$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 "$@" > $root/$out/pass$i.log 2>&1)
  tail -2 $root/$out/pass$i.log
done
rm -rf $out/pass*/*.db $out/pass*/*/*.db
python3 tools/pmc_summary.py $out x3_encode_stream2 x3_decode_split x3_frame_check | tee $out/l2req.txt
