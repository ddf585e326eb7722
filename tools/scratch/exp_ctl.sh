#!/bin/bash
# The encoders' control block cleared by the previous call's last kernel instead of a memset in front of the next call's
# first (ctl0 = before, ctl1 = after): step time of config 3 and the small streams.
out=gpurun_out/r5q; mkdir -p $out
for r in 1 2 3; do for v in ctl0 ctl1; do
  X3HIP_LIB=$PWD/x3-rust_amd/lib/variants/libx3hip_$v.so python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-extras 2>/dev/null | tail -1 > $out/b_${v}_$r.json
  python3 - $out/b_${v}_$r.json $v $r <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["configs"]["config2"]
print(sys.argv[2],sys.argv[3],"value",d["value"],"ms/step",d["ms_per_step"],"kern",d["kernels_ms"]["encode"],d["kernels_ms"]["decode"],
      "| cfg2 enc",c["encode"]["ms"],"dec",c["decode"]["ms"],"rt",c["round_trip_ms"],"| seg enc",c["with_segment_index"]["encode"]["ms"],"dec",c["with_segment_index"]["decode"]["ms"],"rt",c["with_segment_index"]["round_trip_ms"])
PY
done; done | tee $out/ctl_ab.txt
python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep "passed\|failed" | tee $out/tests.txt
timeout 400 python3 tools/fuzz_parity.py --seed 591 --minutes 3 --families ebasm 2>&1 | grep -v amdgpu.ids | tail -1 | tee $out/fuzz.txt
