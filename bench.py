#!/usr/bin/env python3
"""bench.py -- encode+decode round trip of BASELINE config 3 (1 h of 192 kHz mono, 691.2 M samples)
on N MI355X GPUs, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic audio already resident in HBM:
x3_encode_dev (frame sizes -> scan -> encode+compact into the final stream) followed by
x3_decode_dev (header + payload CRC check + decode) of the stream just produced.  With N > 1 the
frames are sharded (weak scaling: every rank owns its own hour of audio = BASELINE config 4 at
N = 8); the only data-path exchange per step is the RCCL all-gather of the sub-stream lengths that
places each rank's sub-stream in the global .x3a byte range.  The full reassembly gather to rank 0
is timed separately and reported under "gather" (DESIGN.md section "Multi-GPU").

Rank 0 prints ONE JSON line (see the contract in the task statement), with
  roofline     -- the encode kernel: algorithmic bytes (2 B/sample read + stream bytes written)
                  / mean launch time from HIP events on the launch stream, vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (C port of the reference algorithm, 1 thread) on a bounded
                  sample of the same signal, timed in this run on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SAMPLES = 691_200_000  # config 3: 1 h @ 192 kHz mono
SEED = 0x58330003


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--samples", type=int, default=N_SAMPLES, help="samples per GPU (default: config 3)")
    ap.add_argument("--kind", type=int, default=2, help="synthetic signal (2 = hydrophone noise)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="samples timed on the CPU baseline")
    ap.add_argument("--cpu-reps", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import x3hip
    from x3hip import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # X3_BENCH_FORCE_DIST=1: take the distributed path (RCCL init, length exchange, gather, barriers) with one rank
    # too -- a way to exercise it on a one-GPU box (launch through torch.distributed.run --nproc-per-node 1)
    if world > 1 or os.environ.get("X3_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = args.samples
    p = x3hip.Params.default()
    L = x3hip.lib()
    F = L.x3_num_frames(n, C.byref(p))
    cap = L.x3_encode_bound(n, C.byref(p))

    stream = torch.cuda.current_stream(dev)
    ctx = x3hip.Context(local_rank, stream=stream.cuda_stream)

    wav = torch.empty(n, dtype=torch.int16, device=dev)
    out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
    off = torch.empty(F + 1, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.int16, device=dev)
    # every rank owns a different hour of the same seeded signal
    ctx.synth_dev(args.kind, SEED, rank * n, n, wav.data_ptr())
    torch.cuda.synchronize(dev)

    lens = torch.zeros(world, dtype=torch.int64, device=dev)

    def step():
        rc = ctx.encode_dev(wav.data_ptr(), n, p, out.data_ptr(), cap, 0, off.data_ptr())
        assert rc == 0, (rc, ctx.last_error())
        work = None
        mode = "after" if os.environ.get("X3_BENCH_EXCHANGE") == "after" else "beside"  # (diagnostic switch)
        if dist is not None and mode == "beside":
            # the exchange step of the sharded path: sub-stream lengths -> global byte offsets.  Every rank decodes
            # its own frames, so the 8-byte all-gather runs beside the decoder and is waited for at the end of the step
            _, work = shard.exchange_lengths(off[F:F + 1], out=lens, async_op=True)
        rc = ctx.decode_dev(out.data_ptr(), cap, off.data_ptr(), F, p, back.data_ptr(), n, n_per_clip=n)
        assert rc == 0, (rc, ctx.last_error())
        if dist is not None and mode == "after":
            _, work = shard.exchange_lengths(off[F:F + 1], out=lens, async_op=True)
        if work is not None:
            work.wait()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    rc, pos, stats = ctx.encode_result()
    assert rc == 0, (rc, ctx.last_error())
    rc, first_bad, st, before = ctx.decode_result()
    assert (rc, first_bad, st, before) == (0, F, 0, n), (rc, first_bad, st, before)

    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_time()
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rc, pos, stats = ctx.encode_result()
    assert rc == 0, (rc, ctx.last_error())
    rc, first_bad, st, before = ctx.decode_result()
    assert (rc, first_bad, st, before) == (0, F, 0, n), (rc, first_bad, st, before)
    ktimes = {}
    for name, which in (("encode", 0), ("decode", 1), ("frame_sizes", 2), ("scan", 3), ("frame_check", 4)):
        ms, cnt = ctx.kernel_time(which)
        ktimes[name] = ms / max(cnt, 1)
    ctx.enable_kernel_timing(False)

    # ---- bit-exactness of the timed output
    assert torch.equal(back, wav), "decode(encode(x)) != x"
    import oracle_lib as O
    offs = off.cpu().numpy()
    assert int(offs[-1]) == pos
    for f in [0, F // 3, F - 1]:
        s = wav[f * p.spf:(f + 1) * p.spf].cpu().numpy()
        enc = out[int(offs[f]):int(offs[f + 1])].cpu().numpy()
        assert np.array_equal(enc, O.encode(s)[1]), "frame %d differs from the CPU oracle" % f

    # ---- optional: reassembly gather of the sub-streams to rank 0 (timed on its own)
    gather = None
    if dist is not None and not args.no_gather:
        lens_h = lens.cpu()
        torch.cuda.synchronize(dev)
        barrier()
        g0 = time.perf_counter()
        whole = shard.gather_stream(out, lens_h, dst=0)
        torch.cuda.synchronize(dev)
        barrier()
        g1 = time.perf_counter()
        starts = shard.global_offsets(lens_h)
        del whole
        gather = {"ms": round((g1 - g0) * 1e3, 3), "bytes": int(starts[-1]),
                  "pattern": "grouped ncclSend/ncclRecv to rank 0 (one xGMI link per peer)"}

    # ---- CPU baseline: the oracle (port of the reference algorithm), 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        m = min(args.cpu_sample, n)
        sample = wav[:m].cpu().numpy()
        OL = O.lib(native=True)
        po = O.Params.default()
        es, ds, sl = C.c_double(0), C.c_double(0), C.c_uint64(0)
        rc = OL.x3o_time_roundtrip(sample.ctypes.data, m, C.byref(po), args.cpu_reps, C.byref(es), C.byref(ds),
                                   C.byref(sl))
        assert rc == 0, rc
        tot = es.value + ds.value
        cpu = {"value": round(m * args.cpu_reps / tot / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": "first %d samples of the same signal, %d reps, encode %.2f s + decode %.2f s; "
                         "oracle/x3_oracle.c -O3 -march=native -flto" % (m, args.cpu_reps, es.value, ds.value),
               "encode_msamples_s": round(m * args.cpu_reps / es.value / 1e6, 2),
               "decode_msamples_s": round(m * args.cpu_reps / ds.value / 1e6, 2)}
        # the generous baseline: the same port, frame-parallel over all host cores -- every
        # thread round-trips its own slice of whole frames (ctypes releases the GIL)
        import threading
        ncores = os.cpu_count() or 1
        if ncores > 1 and args.cpu_reps > 0:
            spf = 10000
            per = max(spf, (m // ncores) // spf * spf)
            parts = [sample[i * per:(i + 1) * per] for i in range(ncores) if (i + 1) * per <= m]
            rcs = [0] * len(parts)
            # a few seconds of wall time: assume the whole machine manages ~16x the single-thread rate just
            # measured (memory-bound port, SMT siblings), never more than the thread count
            agg = cpu["value"] * 1e6 * min(ncores, 16)
            reps_mt = max(1, int(4.0 * agg / (len(parts) * per) + 0.5))

            def work(i):
                e, d, l = C.c_double(0), C.c_double(0), C.c_uint64(0)
                rcs[i] = OL.x3o_time_roundtrip(parts[i].ctypes.data, parts[i].size, C.byref(po), reps_mt,
                                               C.byref(e), C.byref(d), C.byref(l))
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            assert not any(rcs), rcs
            cpu["all_cores"] = {"value": round(sum(q.size for q in parts) * reps_mt / wall / 1e6, 2),
                                "unit": "Msamples/s", "cores": len(parts),
                                "sample": "%d slices of %d samples, %d reps, %.2f s wall" % (len(parts), per, reps_mt, wall)}

    # ---- the host-buffer entry points (x3_encode / x3_decode_stream): pageable host memory in, PCIe both ways,
    # internal staging; reported beside the headline, never as `value`
    host_api = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # caller-owned pageable buffers, allocated and touched beforehand (a caller that streams audio has
        # them already); first call = cold (device scratch is allocated), second = steady state
        L = x3hip.lib()
        hwav = wav.cpu().numpy()
        hcap = L.x3_encode_bound(n, C.byref(p))
        hout = np.zeros(hcap, dtype=np.uint8); hout[::4096] = 1
        hback = np.zeros(n, dtype=np.int16); hback[::2048] = 1
        hpos, hn, hok, herr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        hstats = np.zeros(6, dtype=np.uint64)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            rc = L.x3_encode(ctx._h, hwav.ctypes.data, n, 1, C.byref(p), hout.ctypes.data, hcap, 0, C.byref(hpos),
                             hstats.ctypes.data)
            t1 = time.perf_counter()
            rc2 = L.x3_decode_stream(ctx._h, hout.ctypes.data, hpos.value, C.byref(p), hback.ctypes.data, n, C.byref(hn),
                                     C.byref(hok), C.byref(herr))
            t2 = time.perf_counter()
            assert rc == 0 and rc2 == 0 and hn.value == n, (rc, rc2, hn.value)
            times.append((t1 - t0, t2 - t1))
        assert np.array_equal(hback, hwav)
        te, td = times[-1]
        host_api = {"samples": n, "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
                    "encode_msamples_s": round(n / te / 1e6, 1), "decode_msamples_s": round(n / td / 1e6, 1),
                    "msamples_s": round(n / (te + td) / 1e6, 1),
                    "cold_call_ms": [round(times[0][0] * 1e3, 2), round(times[0][1] * 1e3, 2)],
                    "pcie_gb_s": round((2 * n + hpos.value) / te / 1e9, 1),
                    "note": "x3_encode + x3_decode_stream on caller-owned pageable host buffers: H2D, kernels, D2H "
                            "and the host-side frame walk, second call (the first also allocates device scratch)"}
        del hwav, hout, hback

    if rank == 0:
        total_samples = n * world
        value = total_samples * args.steps / elapsed / 1e6
        # algorithmic HBM bytes per launch (DESIGN.md "Kernels"): 2 B per sample + P stream bytes for
        # the encoder and the decoder; the size pass re-reads the samples; the check pass reads the stream
        alg = {"encode": 2 * n + pos, "decode": 2 * n + pos, "frame_sizes": 2 * n, "frame_check": pos}
        kname = {"encode": "x3_encode_stream_kernel" if ktimes.get("frame_sizes", 0.0) == 0.0 else "x3_encode_frames_kernel<false>",
                 "decode": "x3_decode_split_kernel",
                 "frame_sizes": "x3_encode_frames_kernel<true>", "frame_check": "x3_frame_check_kernel"}
        alg = {k: v for k, v in alg.items() if ktimes.get(k, 0.0) > 0.0}  # the two-pass fallback kernels may not run
        # the dominant kernel of the step's critical path: the frame check runs BESIDE the decoder on a second
        # stream (its co-running time is stretched by the decoder's waves), so it is reported but not a candidate
        dominant = max((k for k in alg if k != "frame_check"), key=lambda k: ktimes[k])
        traffic = {}
        try:  # HBM bytes per launch from rocprofv3 PMC passes (profiles/, see DESIGN.md "Measurement")
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        except Exception:
            pass

        def roof(k):
            t = ktimes[k] / 1e3
            ach = alg[k] / t / 1e9
            return {"bound": "hbm", "kernel": kname[k], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get(kname[k], {}).get("hbm_bytes_per_launch"),
                    "algorithmic_bytes": int(alg[k]), "avg_launch_ms": round(ktimes[k], 4)}
        res = {
            "metric": "Msamples/s encode+decode (bit-exact), 1h 192kHz mono; % HBM-read roofline",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i16",
            "data": "synthetic",
            "config": {"workload": "config 3: 1 h 192 kHz mono hydrophone-like noise, encode+decode round trip per GPU",
                       "samples_per_gpu": n, "frames_per_gpu": int(F), "stream_bytes_per_gpu": int(pos),
                       "bytes_per_sample": round(pos / n, 4), "block_len": 20, "blocks_per_frame": 500,
                       "sharding": "frames sharded across ranks; all-gather of sub-stream lengths per step"},
            "roofline": roof(dominant),
            "roofline_all": {k: roof(k) for k in alg},
            "encode_read_frac": round(2 * n / (ktimes["encode"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernels_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "cpu_baseline": cpu,
        }
        if host_api is not None:
            res["host_buffer_api"] = host_api
        if gather is not None:
            res["gather"] = gather
        print(json.dumps(res), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
