#!/bin/bash
# Collect everything profiles/rN/ holds, on the GPU box, from the repo root:
#   tools/profile_round.sh gpurun_out/r1
# 1. the default bench line; 2. rocprofv3 --kernel-trace --stats of the same command; 3. separate PMC
# passes (FETCH_SIZE, WRITE_SIZE, SQ counters) as MI355X_MICROARCH.md prescribes (no other trace domains).
# (the stats run skips the extras -- cold context, frame walk, extreme contents -- so that its per-kernel averages are those
# of the timed steps and agree with the HIP-event averages in the bench line)
out=$1
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
python3 bench.py --details $out/bench_details.json > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/stats -o bench --output-format csv -- python3 $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-measure-traffic --no-extras > $root/$out/stats.log 2>&1)
cp $out/stats/bench_kernel_stats.csv $out/bench_kernel_stats.csv 2>/dev/null
cp $out/stats/bench_domain_stats.csv $out/bench_domain_stats.csv 2>/dev/null
# (--stats averages every launch of the process, the placement probe's 64 x 13 round trips on other buffers among them: the
# timed passes (three of 100 steps each) on their own, from the same run.s kernel trace, beside what its JSON line said)
python3 tools/timed_steps_stats.py $out/stats 300 $out/stats.log > $out/bench_kernel_stats_timed_steps.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $root/$out/pmc$i -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 > $root/$out/pmc$i.log 2>&1)
done
python3 tools/pmc_summary.py $out/pmc1 x3_ > $out/pmc_fetch_size.txt
python3 tools/pmc_summary.py $out/pmc2 x3_ > $out/pmc_write_size.txt
python3 tools/pmc_summary.py $out/pmc3 x3_ > $out/pmc_sq.txt
python3 tools/make_traffic.py $out/pmc1 $out/pmc2 > $out/traffic.json
rm -rf $out/stats/*.db $out/pmc*/*.db
cat $out/bench_default.json
