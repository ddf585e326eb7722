// x3_mgpu.hip -- frames sharded over the GPUs of one node (x3_shard_*, x3_mgpu_*; librccl through dlopen).  Host code
// only (C ABI: include/x3hip.h; units: x3_internal.h).
#include "x3_internal.h"

#include "x3_mgpu.h"
