#!/bin/bash
# Which property of a box / a process selects the decoder's pace?  
# 1. what the box says about itself; 2. six fresh processes, same command; 3. the same with the check kernel's stream idle
out=${1:-gpurun_out/r6/modes}
mkdir -p $out
rocm-smi --showclocks --showpower --showvbios --showmemorypartition --showcomputepartition --showperflevel --showtemp --showfwinfo > $out/rocm_smi.txt 2>&1
rocm-smi --showmclkrange --showsclkrange --showmaxpower >> $out/rocm_smi.txt 2>&1
for i in 1 2 3 4 5 6; do
  python3 tools/kbench.py --steps 20 2>&1 | tail -1 | cut -c1-130
done > $out/six_processes.txt
for i in 1 2 3; do
  python3 tools/kbench.py --steps 20 --pad $((i * 1237)) 2>&1 | tail -1 | cut -c1-130
done > $out/three_processes_padded.txt
cat $out/six_processes.txt $out/three_processes_padded.txt
grep -i "mclk\|sclk\|partition\|perf\|power\|vbios" $out/rocm_smi.txt | head -40
