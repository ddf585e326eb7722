"""world_size-2/3 gloo tests (CPU) of the multi-GPU path's sharding arithmetic and exchange pattern.

What ships is the C ABI: x3_shard_frame_range / x3_shard_sample_range / x3_shard_offsets (host arithmetic, no GPU needed)
and, on the GPUs, the RCCL calls of x3_shard_exchange_lengths / x3_shard_gather.  Here every rank takes ITS ranges and
offsets from those C functions (through libx3hip.so, as bench.py does), encodes its frame range with the CPU oracle --
there is no GPU in this container -- and gloo only moves the bytes the way RCCL does on the GPUs: one all-gather of the
lengths, then point-to-point sends to the root at the offsets x3_shard_offsets gave.  Rank 0 checks that the reassembled
stream is byte-identical to the oracle's encoding of the whole signal."""
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


def _worker(rank, world, port, n, result_path, sharded=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
    import ctypes as C
    import oracle_lib as O
    import x3hip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = x3hip.Params.default()
        s_lo, s_n = x3hip.shard_sample_range(n, p, rank, world)          # x3_shard_sample_range
        f_lo, f_n = x3hip.shard_frame_range((n + p.spf - 1) // p.spf, rank, world)   # x3_shard_frame_range
        assert s_lo == min(n, f_lo * p.spf) and s_n == max(0, min(n, (f_lo + f_n) * p.spf) - s_lo)
        wav = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 77, s_lo, s_n)
        rc, sub, stats = O.encode(wav) if s_n else (0, np.zeros(0, dtype=np.uint8), None)
        assert rc == 0
        local = torch.from_numpy(np.ascontiguousarray(sub))
        # step 1 (x3_shard_exchange_lengths on the GPUs: ncclAllGather of one uint64 per rank)
        mine = torch.tensor([local.numel()], dtype=torch.int64)
        got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, mine)
        lens = [int(t.item()) for t in got]
        assert lens[rank] == local.numel()
        starts = x3hip.shard_offsets(lens)                                # x3_shard_offsets
        assert len(starts) == world + 1 and starts[-1] == sum(lens)
        assert all(s % 2 == 0 for s in starts)  # sub-streams concatenate without padding
        if sharded:
            # step 2, SHARDED (x3_shard_write_at on the GPUs: every rank brings its own sub-stream down and writes it at
            # base + starts[rank] of ONE file; nobody takes in the whole stream).  Here: the same pwrite, with the archive
            # header in front of the frames as base -- the file is then a complete .x3a archive.
            rc_h, hdr = x3hip.archive_header_write(192000, p)
            assert rc_h == 0
            base = hdr.size
            path = result_path + ".x3a"
            if rank == 0:
                fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
                os.pwrite(fd, hdr.tobytes(), 0)
            dist.barrier()
            if rank != 0:
                fd = os.open(path, os.O_RDWR)
            if lens[rank]:
                assert os.pwrite(fd, local.numpy().tobytes(), base + starts[rank]) == lens[rank]
            os.close(fd)
            dist.barrier()
            if rank == 0:
                full = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 77, 0, n)
                rc, ref, _ = O.x3a_encode(full, 192000)                    # the oracle's wav_to_x3a minus the files
                got = np.fromfile(path, dtype=np.uint8)
                ok = rc == 0 and got.size == ref.size and np.array_equal(got, ref)
                os.unlink(path)
                open(result_path, "w").write("ok" if ok else "mismatch")
            dist.barrier()
            return
        # step 2 (x3_shard_gather on the GPUs: grouped ncclRecv on the root at starts[r], ncclSend on the peers)
        whole = None
        if rank == 0:
            whole = torch.zeros(starts[-1], dtype=torch.uint8)
            whole[starts[0]:starts[1]] = local
            reqs = [dist.irecv(whole[starts[r]:starts[r + 1]], src=r) for r in range(1, world) if lens[r]]
            for q in reqs:
                q.wait()
        elif lens[rank]:
            dist.send(local, dst=0)
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


@pytest.mark.parametrize("world,n", [(2, 123457), (3, 70001), (2, 5)])
def test_sharded_write_makes_the_reference_archive(tmp_path, world, n):
    """--gather sharded / x3_shard_write_at: every rank writes its sub-stream at its own offset of one file behind the
    archive header; the file is byte-identical to the oracle's .x3a of the whole signal (encodefile.rs:48-138)"""
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), n, str(result), True), nprocs=world, join=True)
    assert result.read_text() == "ok"


def test_frame_ranges_cover_everything():
    sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
    import x3hip
    for F in [0, 1, 2, 7, 8, 9, 69120, 552960]:
        for world in [1, 2, 3, 4, 8]:
            ranges = [x3hip.shard_frame_range(F, r, world) for r in range(world)]   # (first, count)
            assert ranges[0][0] == 0 and ranges[-1][0] + ranges[-1][1] == F
            assert all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
