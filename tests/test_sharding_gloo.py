"""world_size-2/3 gloo tests (CPU) of the multi-GPU path's sharding + exchange logic
(x3-rust_amd/x3hip/shard.py, used unchanged by bench.py with the nccl/RCCL backend).
Each rank encodes its frame range with the CPU oracle (there is no GPU here); rank 0 checks that the
reassembled stream is byte-identical to the oracle's encoding of the whole signal."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, result_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
    import oracle_lib as O
    import x3hip
    from x3hip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spf = 10000
        s_lo, s_n = shard.sample_range(n, spf, rank, world)
        wav = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 77, s_lo, s_n)
        rc, sub, stats = O.encode(wav) if s_n else (0, np.zeros(0, dtype=np.uint8), None)
        assert rc == 0
        local = torch.from_numpy(np.ascontiguousarray(sub))
        lens = shard.exchange_lengths(local.numel())
        assert lens.tolist()[rank] == local.numel()
        # the overlapped form bench.py uses: same result once the work handle has been waited for
        lens2, work = shard.exchange_lengths(local.numel(), async_op=True)
        work.wait()
        assert lens2.tolist() == lens.tolist()
        starts = shard.global_offsets(lens)
        assert all(s % 2 == 0 for s in starts)  # sub-streams concatenate without padding
        whole = shard.gather_stream(local, lens, dst=0)
        if rank == 0:
            full = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 77, 0, n)
            rc, ref, _ = O.encode(full)
            ok = rc == 0 and whole.numel() == ref.size and np.array_equal(whole.numpy(), ref)
            # and the reassembled stream decodes back to the signal
            rc2, back, fok, ferr = O.decode_stream(whole.numpy(), wav_cap=n)
            ok = ok and rc2 == 0 and ferr == 0 and np.array_equal(back, full)
            open(result_path, "w").write("ok" if ok else "mismatch")
        else:
            assert whole is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 123457), (2, 10000), (3, 70001), (2, 5)])
def test_sharded_encode_reassembles_to_the_reference_stream(tmp_path, world, n):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), n, str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"


def test_frame_ranges_cover_everything():
    sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
    from x3hip import shard
    for F in [0, 1, 2, 7, 8, 9, 69120, 552960]:
        for world in [1, 2, 3, 4, 8]:
            ranges = [shard.frame_range(F, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == F
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
