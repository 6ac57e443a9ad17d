#!/usr/bin/env python3
"""bench.py -- partition-assignments/sec of the lag-based assignor hot path on MI355X.

One "step" = one pass of the whole hot path (lag compute -> sort by lag desc -> greedy
assignment) over one batch of synthetic topics that is already resident in HBM when the
timed region starts.  Default workload at N=1 is the configuration BASELINE.json quotes
its metric on: 100 000 topics x 256 partitions x 32 consumers, Zipf(1.1) lags.

    python bench.py                         # N=1, 2 000 timed steps after 200 warm-up steps (about 15 s in all)
    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: topics are independent, so they shard across ranks with no data-path
collective (weak scaling: every rank owns a full per-GPU batch).  `--gather` adds the
north star's RCCL all-gather of the result arrays inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:35
# read begin 8 + end 8 + committed 8 + partition id 4, write id-in-assignment-order 4 +
# member rank 4 = 36 B/partition.  auto.offset.reset=latest never reads `begin`: 28 B/partition there.
BYTES_PER_PARTITION = {"earliest": 36, "latest": 28}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # written by tools/pmc_parse.py (separate --pmc passes)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="the target batch runs the package at its 1 400 W cap; the power controller needs ~50 ms of "
                         "back-to-back launches to settle (100 timed steps: 183 us/step, 2 000: 164, 10 000: 163)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--topics", type=int, default=100000)
    ap.add_argument("--partitions", type=int, default=256)
    ap.add_argument("--consumers", type=int, default=32)
    ap.add_argument("--dist", choices=["zipf", "uniform40"], default="zipf",
                    help="lag distribution: Zipf(1.1) shuffled per topic (cfg3 / target) or uniform on [0, 2^40) (cfg4)")
    ap.add_argument("--reset-mode", choices=["latest", "earliest"], default="earliest",
                    help="earliest reads all four marshalled arrays (the 36 B/partition of SURVEY 8d)")
    ap.add_argument("--algo", choices=["auto", "wide", "argmin"], default="auto")
    ap.add_argument("--gather", action="store_true", help="all-gather the result arrays (RCCL) in the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def make_device_workload(torch, dev, T, P, C, seed, dist="zipf"):
    """Target-config inputs generated on the device: Zipf(1.1) lags shuffled over the
    partitions of each topic, shuffled partition ids, offsets built from the lag."""
    from kafka_lag_based_assignor_amd import synth
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if dist == "zipf":
        base = torch.from_numpy(synth.zipf_lags(P)).to(dev)                       # [P] int64
        order = torch.rand(T, P, device=dev, generator=g).argsort(dim=1)
        lag = base[order].reshape(-1).contiguous()
    else:                                                                         # uniform on [0, 2^40)
        order = None
        lag = torch.randint(0, 1 << 40, (T * P,), device=dev, generator=g, dtype=torch.int64)
    pid = torch.rand(T, P, device=dev, generator=g).argsort(dim=1).to(torch.int32).reshape(-1).contiguous()
    del order
    com = torch.randint(0, 1 << 20, (T * P,), device=dev, generator=g, dtype=torch.int64)
    end = com + lag
    none = torch.rand(T * P, device=dev, generator=g) < 0.01
    com = torch.where(none, torch.full_like(com, -1), com)
    begin = torch.zeros(T * P, device=dev, dtype=torch.int64)
    part_off = torch.arange(T + 1, device=dev, dtype=torch.int64) * P
    cons_off = torch.arange(T + 1, device=dev, dtype=torch.int64) * C
    cons_rank = torch.arange(C, device=dev, dtype=torch.int32).repeat(T).contiguous()
    return dict(part_off=part_off, pid=pid, begin=begin, end=end, committed=com, lag=lag,
                cons_off=cons_off, cons_rank=cons_rank)


def alloc_outputs(torch, dev, T, P, C):
    return dict(pid=torch.empty(T * P, device=dev, dtype=torch.int32),
                rank=torch.empty(T * P, device=dev, dtype=torch.int32),
                total=torch.empty(T * C, device=dev, dtype=torch.int64))


def make_batch(N, w, outs, T, P, C, latest, algo):
    """la_device_batch over device-resident tensors; returns (batch, objects to keep alive)."""
    b = N.DeviceBatch()
    b.n_topics = T
    b.reset_mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
    b.algo = {"auto": N.LA_ALGO_AUTO, "wide": N.LA_ALGO_ROUNDS_WIDE, "argmin": N.LA_ALGO_ARGMIN}[algo]
    b.n_partitions = T * P
    b.n_consumers = T * C
    b.max_partitions_per_topic = P
    b.max_consumers_per_topic = C
    b.d_part_off = w["part_off"].data_ptr()
    b.d_partition_id = w["pid"].data_ptr()
    b.d_begin_off = None if latest else w["begin"].data_ptr()
    b.d_end_off = w["end"].data_ptr()
    b.d_committed_off = w["committed"].data_ptr()
    b.d_lag = None
    b.d_cons_off = w["cons_off"].data_ptr()
    b.d_cons_rank = w["cons_rank"].data_ptr()
    b.d_out_partition = outs["pid"].data_ptr()
    b.d_out_member_rank = outs["rank"].data_ptr()
    b.d_out_total_lag = outs["total"].data_ptr()
    keep = []
    if P > 1024 or C > 64:
        h_part = w["part_off"].cpu().numpy()
        h_cons = w["cons_off"].cpu().numpy()
        b.h_part_off = h_part.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        b.h_cons_off = h_cons.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        keep = [h_part, h_cons]
    return b, keep


def measured_traffic(T, P, C, mode, algo):
    """HBM bytes per launch from the PMC passes (tools/pmc_probe.py + tools/pmc_parse.py), if a
    summary for exactly this workload is committed; rocprofv3 cannot wrap itself from inside here."""
    try:
        with open(TRAFFIC_FILE) as fh:
            t = json.load(fh)
        for e in t.get("entries", []):
            if (e["topics"], e["partitions"], e["consumers"], e["reset_mode"], e["algo"]) == (T, P, C, mode, algo):
                return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from kafka_lag_based_assignor_amd import _native as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LA_BENCH_FORCE_DIST=1 runs the RCCL legs (init, barrier, max-reduce) at world size 1 too, so that the N>1
    # code path can be exercised under torchrun on a one-GPU box
    use_dist = world > 1 or os.environ.get("LA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    T, P, C = args.topics, args.partitions, args.consumers
    n_part = T * P
    w = make_device_workload(torch, dev, T, P, C, seed=0x5EED + rank, dist=args.dist)
    outs = alloc_outputs(torch, dev, T, P, C)
    out_pid, out_rank, out_total = outs["pid"], outs["rank"], outs["total"]
    if args.gather and use_dist:
        gathered_pid = torch.empty(world * n_part, device=dev, dtype=torch.int32)
        gathered_rank = torch.empty(world * n_part, device=dev, dtype=torch.int32)

    ctx = N.Context(local_rank)
    latest = args.reset_mode == "latest"
    b, _keep = make_batch(N, w, outs, T, P, C, latest, args.algo)

    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.assign_batch_device(b, stream)
        if args.gather and use_dist:
            dist.all_gather_into_tensor(gathered_pid, out_pid)
            dist.all_gather_into_tensor(gathered_rank, out_rank)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.sync(stream)

    # timed region: exactly K steps, bracketed by barrier + synchronize on both sides
    # HIP events around about 200 of the steps (every step when K <= 200): an event record is a marker packet
    # in the queue, and two per step would be a measurable part of a 160 us step
    stride = max(1, args.steps // 200)
    ev = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for s in range(0, args.steps, stride)}
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        pair = ev.get(s)
        if pair is None:
            step()
        else:
            pair[0].record()
            step()
            pair[1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.sync(stream)

    # HIP-event duration of the assign launch on the stream it runs on (the kernels of one step)
    kern_ms = float(np.mean([a.elapsed_time(z) for a, z in ev.values()]))

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_units = world * n_part * args.steps
    value = total_units / elapsed

    bpp = BYTES_PER_PARTITION[args.reset_mode]
    achieved = bpp * n_part / (kern_ms * 1e-3) / 1e9
    tr = measured_traffic(T, P, C, args.reset_mode, args.algo)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                "kernel": ("wave_tile_packed_kernel (+ the wide-record kernel over its deferred-tile list, empty here)"
                           if (P <= 1024 and C <= 64) else
                           "block_topic_kernel (one workgroup per topic; + the list copy)"
                           if (P <= 8192 and C <= 2048) else "large-topic path (all kernels)"),
                "kernel_ms": round(kern_ms, 4),
                "algorithmic_bytes_per_partition": bpp,
                "algorithmic_bytes_per_launch": bpp * n_part}
    if tr:
        roofline["traffic_source"] = tr.get("source")

    # ---- quality metric of BASELINE.json: max/min per-consumer total lag per topic (min clamped to 1) ----
    lag_ratio = None
    if C > 0 and T * C == out_total.numel():
        tot = out_total.reshape(T, C).to(torch.float64)
        ratio = tot.max(dim=1).values / tot.min(dim=1).values.clamp(min=1.0)
        lag_ratio = {"mean": round(float(ratio.mean()), 4), "p99": round(float(torch.quantile(ratio[:1000000], 0.99)), 4),
                     "max": round(float(ratio.max()), 4),
                     "what": "max/min per-consumer assigned lag per topic of the last step's assignment (README.md:54-69 quotes 1.10 for its example)"}

    # ---- parity spot check + cpu_baseline (oracle; test infrastructure, timed on host cores) ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle
        h = {k: v.cpu().numpy() for k, v in w.items()}
        g_pid, g_rank, g_tot = out_pid.cpu().numpy(), out_rank.cpu().numpy(), out_total.cpu().numpy()
        chunk = max(1, min(T, 2000))
        done = 0
        spent = 0.0
        ok = True
        while done < T and spent < args.cpu_seconds:
            t1 = min(T, done + chunk)
            sl = slice(done * P, t1 * P)
            c0 = time.perf_counter()
            lag = oracle.compute_lags(h["begin"][sl], h["end"][sl], h["committed"][sl], latest)
            e_pid, e_rank, e_tot = oracle.assign_flat(h["part_off"][done:t1 + 1] - done * P, h["pid"][sl], lag,
                                                      h["cons_off"][done:t1 + 1] - done * C,
                                                      h["cons_rank"][done * C:t1 * C])
            spent += time.perf_counter() - c0
            ok &= bool(np.array_equal(e_pid, g_pid[sl]) and np.array_equal(e_rank, g_rank[sl]) and
                       np.array_equal(e_tot, g_tot[done * C:t1 * C]))
            done = t1
        parity = {"checked_topics": done, "bit_exact": ok}
        # the same batch through the host-buffer entry point (la_assign_batch: H2D of the five input arrays,
        # kernels, D2H of the results); reported beside the bench value, never as it
        host_leg = None
        try:
            # result buffers are the caller's and reused across calls (a Java host's direct ByteBuffers): the first
            # call touches them, the second is timed
            reuse = ctx.assign_batch(h["part_off"], h["pid"], None if latest else h["begin"], h["end"], h["committed"],
                                     N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST, h["cons_off"], h["cons_rank"])
            for o in reuse:
                o.fill(0)
            c0 = time.perf_counter()
            hp, hm, ht = ctx.assign_batch(h["part_off"], h["pid"], None if latest else h["begin"], h["end"],
                                          h["committed"], N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST,
                                          h["cons_off"], h["cons_rank"], out=reuse)
            dt = time.perf_counter() - c0
            host_leg = {"ms": round(dt * 1e3, 2), "value": round(n_part / dt, 1), "unit": "partition-assignments/sec",
                        "what": "one la_assign_batch call on pageable host buffers (results into reused, already "
                                "touched buffers), PCIe copies included",
                        "pcie_gbs": round((n_part * (bpp) + out_total.numel() * 8) / dt / 1e9, 1),
                        "bit_exact_vs_device_path": bool(np.array_equal(hp, g_pid) and np.array_equal(hm, g_rank))}
        except Exception as exc:  # noqa: BLE001 -- a reported extra, not part of the contract
            host_leg = {"error": str(exc)}
        cpu = {"value": round(done * P / spent, 1), "unit": "partition-assignments/sec", "cores": 1,
               "kind": "port",
               "sample": "first %d of %d topics of the same batch, C oracle (oracle/lag_oracle.c, literal "
                         "per-step min), 1 thread, %.1f s" % (done, T, spent),
               "host_cpus": os.cpu_count()}
        if not ok:
            print("PARITY FAILURE against the oracle", file=sys.stderr)

    line = {
        "metric": "partition-assignments/sec (whole node)",
        "value": round(value, 1),
        "unit": "partition-assignments/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "%d topics x %d partitions x %d consumers per GPU, %s lags, "
                               "shuffled partition ids, 1%% no committed offset, auto.offset.reset=%s"
                               % (T, P, C, "Zipf(1.1)" if args.dist == "zipf" else "uniform [0, 2^40)", args.reset_mode),
                   "topics_per_gpu": T, "partitions_per_topic": P, "consumers_per_topic": C,
                   "gather": bool(args.gather and world > 1), "algo": args.algo},
        "roofline": roofline,
        "lag_ratio": lag_ratio,
        "cpu_baseline": cpu,
        "parity": parity,
        "host_boundary": host_leg if (not args.no_cpu_baseline and world == 1) else None,
    }
    print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    if parity is not None and not parity["bit_exact"]:
        sys.exit(2)


if __name__ == "__main__":
    main()
