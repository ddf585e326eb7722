#!/bin/bash
# bench.py in fresh processes with and without the placement probe: value, kernels_ms.decode, what the probe saw
out=${1:-gpurun_out/r6/bench_placement_stats.txt}
reps=${2:-4}
{
for r in $(seq 1 $reps); do
  for pl in ${PLACES:-6 1}; do
    echo -n "place $pl rep $r: "
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-measure-traffic --no-extras --no-configs --place $pl --details /tmp/bd.json 2>/dev/null | tail -1 | python3 -c '
import json,sys
j=json.loads(sys.stdin.readline())
print("value %.0f ms_per_step %.4f decode %.4f encode %.4f check %.4f p90/min %s placement %s" % (j["value"], j["ms_per_step"], j["kernels_ms"]["decode"], j["kernels_ms"]["encode"], j["kernels_ms"].get("frame_check",0), j.get("kernels_ms_p90_over_min",{}).get("decode"), j.get("placement",{}).get("step_ms")))'
  done
done
} 2>&1 | tee $out
