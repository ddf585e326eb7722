"""Frame sharding of one X3 stream over the ranks of a torch.distributed job (one process per GPU).

Frames are independent (each re-seeds the predictor with a raw sample and carries its own CRCs --
encoder.rs:189, decoder.rs:42-46), so rank r encodes a contiguous range of whole frames into its own
sub-stream.  Every frame is 20 + even bytes, so sub-streams concatenate without padding and the ONLY
coupling between ranks is the byte offset of each sub-stream: an exclusive scan of the sub-stream
lengths.  `exchange_lengths` is that exchange (one tiny all-gather); `gather_stream` is the optional
reassembly of the whole .x3a byte stream on one rank (grouped send/recv: with RCCL every peer uses
its own xGMI link to the root, which is why this is not a ring all-gather).

Works with any backend: "nccl" (= RCCL on ROCm) with device tensors, "gloo" with CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def frame_range(n_frames, rank, world):
    """contiguous frame range [lo, hi) of `rank` (remainder frames go to the first ranks)"""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_range(n_samples, samples_per_frame, rank, world):
    """(first sample, number of samples) of `rank`: whole frames, the last frame may be short"""
    n_frames = (n_samples + samples_per_frame - 1) // samples_per_frame
    lo, hi = frame_range(n_frames, rank, world)
    s_lo = lo * samples_per_frame
    s_hi = min(hi * samples_per_frame, n_samples)
    return s_lo, max(0, s_hi - s_lo)


def exchange_lengths(local_len, device=None, group=None, out=None, async_op=False):
    """all-gather of the sub-stream lengths -> tensor[world] (int64) on `device`.
    `local_len` may be a 1-element int64 tensor already on the device (no host sync).
    async_op=True returns (out, work): the collective runs beside whatever is enqueued next (nothing on the
    decode path needs the other ranks' lengths); call work.wait() before `out` is used."""
    world = dist.get_world_size(group)
    if not torch.is_tensor(local_len):
        local_len = torch.tensor([int(local_len)], dtype=torch.int64, device=device)
    if out is None:
        out = torch.empty(world, dtype=torch.int64, device=local_len.device)
    work = dist.all_gather_into_tensor(out, local_len.reshape(1), group=group, async_op=async_op)
    return (out, work) if async_op else out


def global_offsets(lens):
    """exclusive scan of the lengths: byte offset of every rank's sub-stream (+ total at the end)"""
    starts = [0]
    for v in (lens.tolist() if torch.is_tensor(lens) else lens):
        starts.append(starts[-1] + int(v))
    return starts


def gather_stream(local, lens, dst=0, group=None):
    """Reassemble the whole stream on `dst`: returns the uint8 tensor there, None elsewhere.
    `local` is this rank's sub-stream (uint8, at least lens[rank] bytes)."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    starts = global_offsets(lens)
    mine = int(starts[rank + 1] - starts[rank])
    if rank == dst:
        whole = torch.empty(starts[-1], dtype=torch.uint8, device=local.device)
        whole[starts[rank]:starts[rank + 1]].copy_(local[:mine])
        ops = [dist.P2POp(dist.irecv, whole[starts[r]:starts[r + 1]], r, group)
               for r in range(world) if r != dst and starts[r + 1] > starts[r]]
    else:
        whole = None
        ops = [dist.P2POp(dist.isend, local[:mine].contiguous(), dst, group)] if mine else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return whole
