#!/bin/bash
# The GPU tests and the fuzz families with every library / x3_dev_alloc buffer between unmapped pages (x3_fence.h):
# buffers end 16 bytes-aligned at the end of their mapping and start out filled with 0xA5.
out=${OUT:-gpurun_out/r5k}; mkdir -p $out
export X3HIP_FENCE=${FENCE:-16} X3HIP_FENCE_FILL=${FILL:-165}
( timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -15 ) > $out/fence_tests.txt
cat $out/fence_tests.txt
X3_FUZZ_TRACE=$PWD/$out/trace.txt timeout $(( ${MIN:-10} * 60 + 300 )) python3 tools/fuzz_parity.py --seed ${SEED:-551} --minutes ${MIN:-10} --families egdbafms > $out/fence_soak.txt 2>&1
echo "fence soak exit $? [$(cat $out/trace.txt)]"; grep -v amdgpu.ids $out/fence_soak.txt | tail -3
