#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <outdir> [kbench args...]  -- FETCH_SIZE / WRITE_SIZE in separate passes
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $root/$out/pmc1 -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 "$@" > $root/$out/pmc1.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $root/$out/pmc2 -o pmc --output-format csv -- python3 $root/tools/kbench.py --steps 3 "$@" > $root/$out/pmc2.log 2>&1)
python3 tools/make_traffic.py $out/pmc1 $out/pmc2 > $out/traffic.json
rm -rf $out/pmc*/*.db $out/pmc*/*/*.db
python3 - <<PY
import json
t=json.load(open("$out/traffic.json"))
for k,v in t.items():
    if 'synth' in k: continue
    print("%-28s fetch %12d  write %12d" % (k, v['fetch_bytes'], v['write_bytes']))
PY
