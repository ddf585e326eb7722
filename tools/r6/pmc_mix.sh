# instruction mix and busy/wait cycles of the decode phase (one rocprofv3 --pmc pass per group, kernel trace only)
# usage: tools/r6/pmc_mix.sh <outdir>
out=$1
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $root/$out/pass$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 > $root/$out/pass$i.log 2>&1)
done
python3 tools/pmc_summary.py $out x3_decode_blocks x3_decode_split x3_frame_check x3_encode_wave > $out/summary.txt 2>&1
rm -rf $out/pass*/*.db $out/pass*/*/*.db $out/pass*
cat $out/summary.txt
