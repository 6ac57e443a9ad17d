#!/usr/bin/env python3
"""bench.py -- partition-assignments/sec of the lag-based assignor hot path on MI355X.

One "step" = one pass of the whole hot path (lag compute -> sort by lag desc -> greedy
assignment) over one batch of synthetic topics that is already resident in HBM when the
timed region starts.  Default workload at N=1 is the configuration BASELINE.json quotes
its metric on: 100 000 topics x 256 partitions x 32 consumers, Zipf(1.1) lags -- the vectors
of `synth.config("target")`, the same seeded SplitMix64 generator the parity tests and the
golden digests use (one generator is the single source of truth).

    python bench.py                                   # N=1: settle, 50 warm-up steps, 1 000 timed steps
    python bench.py --gpus 1 --steps 20 --warmup 5    # what the driver runs
    python bench.py --phase sort                      # the radix-sort phase of the large path, one 33.5 M-partition topic
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling strong --workload cfg4]

Timing.  The target batch runs the package at its power cap and the power controller needs
~50 ms of back-to-back launches to settle, so before the W counted
warm-up steps there is an UNTIMED settle phase sized in time, not steps (`--settle-ms`, default
150 ms of launches).  Then exactly K steps are timed, bracketed by barrier + synchronize on both
sides, max over ranks.  What ONE rebalance sees -- the first call after the GPU has idled for
a second -- is timed separately and reported as `cold_call_ms`; it is never `value`.

Multi-GPU.  Topics are independent (Main.java:177-184), so they shard across ranks with no
data-path collective.
  --scaling strong (default when WORLD_SIZE > 1: the north star's workload) ONE batch -- the
                   100 000-topic target, or BASELINE config 4 with `--workload cfg4` -- is split
                   over the ranks by the library's own planner (la_plan_shards of the C ABI), every
                   rank runs the hot path on its shard, and the global assignment is reassembled on
                   every rank by ONE RCCL all-gather per step inside the timed region.  What is gathered
                   is the narrow wire format of include/lagassign.h (la_pack_results_on): one element of
                   2 bytes per assigned partition at the target and cfg4 -- ((member rank + 1) << id_bits)
                   | partition id -- instead of two int32 arrays (`--wire int32`: the round-3 form, one
                   [2, cap] int32 buffer).  The gathered map stays in the wire format -- every (partition,
                   member) pair of the global assignment, on every GPU, decodable with
                   la_unpack_results_on -- and `--unpack` puts that expansion into the step; either way
                   its time is in the line (wire.unpack_ms).
  --scaling weak   (default at one rank) every rank owns a full copy of the workload; `--gather`
                   adds the same single all-gather.

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
# radix-sort phase.  Single-kernel passes (decoupled look-back, the default): a pass reads the 8 B key +
# 4 B id of every partition once and writes them once = 24 B/partition.  Four-kernel passes (--sort-form multi): the
# count kernel re-reads the array that carries the digit: 4 B more for an id digit (28 B), 8 B more for a key digit (32 B).
SORT_BYTES = {"single": (24, 24), "multi": (28, 32)}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # written by tools/pmc_parse.py (separate --pmc passes)
SORT_PHASE_PARTITIONS = 1 << 25                                 # past the 256 MiB Infinity Cache


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed back-to-back launches before the warm-up, until this much time has passed: the power "
                         "controller's dip after the first ~9 launches lasts ~50 ms (DESIGN.md section 6)")
    ap.add_argument("--workload", choices=["target", "cfg3", "cfg4", "custom"], default="target",
                    help="target = 100 000 x 256 x 32 Zipf (BASELINE metric); cfg4 = 100 000 x 64 x 8 uniform [0, 2^40); "
                         "custom = --topics/--partitions/--consumers/--dist from the same generator")
    ap.add_argument("--topics", type=int, default=None)
    ap.add_argument("--partitions", type=int, default=None)
    ap.add_argument("--consumers", type=int, default=None)
    ap.add_argument("--dist", choices=["zipf", "uniform40", "pareto"], default=None)
    ap.add_argument("--reset-mode", choices=["latest", "earliest"], default="earliest",
                    help="earliest reads all four marshalled arrays (the 36 B/partition of SURVEY 8d)")
    ap.add_argument("--algo", choices=["auto", "wide", "argmin"], default="auto")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong when WORLD_SIZE > 1 (ONE batch sharded over the ranks + one all-gather per step: "
                         "the north star's multi-GPU workload), weak at one rank")
    ap.add_argument("--gather", action="store_true",
                    help="weak scaling: also all-gather the packed result buffer (RCCL) in the timed region; strong scaling always does")
    ap.add_argument("--wire", choices=["fused", "packed", "int32"], default="fused",
                    help="what the all-gather moves: fused (default) = the narrow wire format written by the assignment kernels "
                         "themselves (LA_FLAG_WIRE_OUT: no pack launch; falls back to `packed` where the batch is not eligible), "
                         "packed = the narrow wire format (2 B per partition at the target) from la_pack_results_on, "
                         "int32 = the two int32 result arrays as one [2, cap] buffer (8 B per partition)")
    ap.add_argument("--unpack", action="store_true",
                    help="packed wire: also expand the gathered map into the two int32 arrays on every rank inside the step "
                         "(la_unpack_results_on); by default the map stays in the wire format -- every (partition, member) pair is "
                         "there, 2 bytes each -- and the expansion is timed separately after the timed region (wire.unpack_ms)")
    ap.add_argument("--no-unpack", action="store_true", help="(the default since round 4; kept for older command lines)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-BASELINE-config block of the default line")
    ap.add_argument("--phase", choices=["assign", "sort"], default="assign",
                    help="sort: time the radix-sort phase of the large path on one topic of --partitions partitions "
                         "(default 33 554 432, no consumers) and report it against the HBM roofline")
    ap.add_argument("--sort-form", choices=["single", "multi"], default="single",
                    help="radix passes of the large path: single = one kernel per pass (decoupled look-back), multi = four")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (parity check, cpu_baseline, host_boundary)")
    ap.add_argument("--no-sort-phase", action="store_true", help="skip the sort_phase leg of the default line")
    ap.add_argument("--rotate", type=int, default=3,
                    help="distinct resident copies of the batch (inputs AND result buffers) the steps rotate over, so that no "
                         "step finds its bytes in the 256 MiB Infinity Cache because the previous step left them there: 3 x 950 MB "
                         "at the target (1 = the round 1-4 form: every step on the same buffers)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC summary instead of two rocprofv3 child passes in this run")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
def make_workload(args):
    """The host-side vectors (synth.Workload) and a description."""
    from kafka_lag_based_assignor_amd import synth
    custom = any(v is not None for v in (args.topics, args.partitions, args.consumers, args.dist))
    if args.workload != "custom" and not custom:
        w = synth.config(args.workload)
        dist = {"target": "Zipf(1.1)", "cfg3": "Zipf(1.1)", "cfg4": "uniform [0, 2^40)"}[args.workload]
        return w, args.workload, dist
    base = {"target": (100000, 256, 32, "zipf"), "cfg3": (1000, 256, 32, "zipf"), "cfg4": (100000, 64, 8, "uniform40"),
            "custom": (100000, 256, 32, "zipf")}[args.workload]
    t = args.topics if args.topics is not None else base[0]
    p = args.partitions if args.partitions is not None else base[1]
    c = args.consumers if args.consumers is not None else base[2]
    d = args.dist or base[3]
    w = synth.make_uniform("custom", 11, t, p, c, d)
    return w, "custom", {"zipf": "Zipf(1.1)", "uniform40": "uniform [0, 2^40)", "pareto": "Pareto(1.5)"}[d]


def sort_phase_workload(n, torch=None, dev=None, ids=None):
    """One topic of n partitions, no consumers, lags uniform on [0, 2^40) from the SplitMix64 generator; ids are a random
    permutation of [0, n): the stable argsort of a second SplitMix64 stream (sorted on the device when one is given: an
    argsort of 33.5 M keys costs ten seconds of one host core).
    Until round 3 the ids were the affine permutation i -> (a*i + c) mod 2^k restricted to [0, n) (LA_SORT_IDS=affine
    still gives it).  With it every tile of an id pass holds EXACTLY the same number of elements of each of the 256
    digits, so a tile's 256 output runs sit at exact multiples of n/256 elements from each other -- 512 KB / 1 MB apart
    for n = 2^25: one set of memory channels.  The id passes then take 241 us where the key passes take 179
    (profiles/archive/r03_large_timeline.txt); partition ids of a real topic do not arrive as an arithmetic progression."""
    from kafka_lag_based_assignor_amd import synth
    lag = (synth.splitmix64(0x9E3779B97F4A7C15 ^ 12, n, 1) >> np.uint64(24)).astype(np.int64)
    if (ids or os.environ.get("LA_SORT_IDS")) == "affine":
        m = 1
        while m < n:
            m <<= 1
        i = np.arange(m, dtype=np.int64)
        perm = (i * 0x9E3779B1 + 0x7F4A7C15) & (m - 1)          # odd multiplier: a bijection on [0, 2^k)
        pid = perm[perm < n].astype(np.int32)
    else:
        keys = (synth.splitmix64(0x9E3779B97F4A7C15 ^ 12, n, 2) >> np.uint64(1)).astype(np.int64)   # 63 bits: order as int64
        if torch is not None and dev is not None:
            pid = torch.argsort(torch.from_numpy(keys).to(dev), stable=True).to(torch.int32).cpu().numpy()
        else:
            pid = np.argsort(keys, kind="stable").astype(np.int32)
    return synth.Workload("sort_phase", 1, np.array([0, n], np.int64), pid, np.zeros(n, np.int64), lag.copy(),
                          np.zeros(n, np.int64), lag, np.array([0, 0], np.int64), np.zeros(0, np.int32), n, 0)


class DeviceShard:
    """Topics [t0, t1) of a workload, resident on one device, with its result buffers and its la_device_batch."""

    def __init__(self, torch, N, dev, w, t0, t1, latest, algo, out_cap=None, flags=0):
        from kafka_lag_based_assignor_amd import sharding
        self.t0, self.t1 = t0, t1
        po, co, ps, cs = sharding.shard_slices(w.part_off, w.cons_off, t0, t1)
        self.h_part_off = np.ascontiguousarray(po)
        self.h_cons_off = np.ascontiguousarray(co)
        self.part_slice, self.cons_slice = ps, cs
        self.n = int(po[-1])
        self.k = int(co[-1])
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
        self.d = dict(part_off=up(po), cons_off=up(co), pid=up(w.partition_id[ps]), begin=up(w.begin[ps]),
                      end=up(w.end[ps]), committed=up(w.committed[ps]), cons_rank=up(w.cons_rank[cs]))
        cap = max(self.n if out_cap is None else out_cap, 1)
        # The two result arrays are the two halves of ONE [2, cap] int32 buffer, so that the global assignment is
        # reassembled by a single all-gather per step.  Zero-filled once: in strong scaling the tail beyond this
        # shard's partitions is all-gathered as padding.
        self.cap = cap
        self.out2 = torch.zeros(2 * cap, device=dev, dtype=torch.int32)
        self.out_pid = self.out2[:cap]
        self.out_rank = self.out2[cap:]
        self.out_total = torch.zeros(max(self.k, 1), device=dev, dtype=torch.int64)
        # (max end offset, max partition id) of the shard, or None when an offset or an id is negative
        e, p_ = w.end[ps], w.partition_id[ps]
        self.bounds = ((int(e.max()), int(p_.max())) if e.size and int(e.min()) >= 0 and int(w.begin[ps].min()) >= 0 and
                       int(p_.min()) >= 0 else None)
        self.max_p = int(np.diff(po).max()) if po.size > 1 else 0
        self.max_c = int(np.diff(co).max()) if co.size > 1 else 0
        self._N = N
        self.batch = self.make_batch(latest, algo, flags)

    def make_batch(self, latest, algo="auto", flags=0, bounds=True):
        """A la_device_batch over this shard's resident arrays (`latest`: auto.offset.reset=latest, `begin` not even passed;
        bounds=False: without LA_FLAG_BOUNDS, the two-launch form of the tile path)."""
        N = self._N
        b = N.DeviceBatch()
        b.n_topics = self.t1 - self.t0
        b.reset_mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
        b.algo = {"auto": N.LA_ALGO_AUTO, "wide": N.LA_ALGO_ROUNDS_WIDE, "argmin": N.LA_ALGO_ARGMIN}[algo]
        b.flags = flags
        b.n_partitions = self.n
        b.n_consumers = self.k
        b.max_partitions_per_topic = self.max_p
        b.max_consumers_per_topic = self.max_c
        b.d_part_off = self.d["part_off"].data_ptr()
        b.d_partition_id = self.d["pid"].data_ptr()
        b.d_begin_off = None if latest else self.d["begin"].data_ptr()
        b.d_end_off = self.d["end"].data_ptr()
        b.d_committed_off = self.d["committed"].data_ptr()
        b.d_lag = None
        b.d_cons_off = self.d["cons_off"].data_ptr()
        b.d_cons_rank = self.d["cons_rank"].data_ptr()
        b.d_out_partition = self.out_pid.data_ptr()
        b.d_out_member_rank = self.out_rank.data_ptr()
        b.d_out_total_lag = self.out_total.data_ptr()
        if b.max_partitions_per_topic > 1024 or b.max_consumers_per_topic > 64:
            b.h_part_off = self.h_part_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
            b.h_cons_off = self.h_cons_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        if bounds and self.bounds is not None and not os.environ.get("LA_BENCH_NO_BOUNDS"):
            # what a marshaller knows without looking at a lag: the largest end offset (no lag exceeds it) and the largest
            # partition id.  When they prove that every tile packs, the tile path is ONE launch (LA_FLAG_BOUNDS)
            b.flags |= N.LA_FLAG_BOUNDS
            b.max_lag_hint, b.max_partition_id_hint = self.bounds
        return b

    def make_wire_batch(self, latest, algo, fmt, d_wire):
        """The same batch with LA_FLAG_WIRE_OUT: the kernels write wire elements ((rank + 1) << id_bits | id) into d_wire instead of
        the two int32 arrays.  None where the library would refuse (no bounds, shapes beyond a tile, wide format)."""
        N = self._N
        if self.bounds is None or fmt is None or int(fmt.elem_bytes) not in (2, 4) or self.max_p > 1024 or self.max_c > 64 or algo != "auto":
            return None
        b = self.make_batch(latest, algo)
        if not (b.flags & N.LA_FLAG_BOUNDS):
            return None
        b.flags |= N.LA_FLAG_WIRE_OUT
        b.d_out_wire = d_wire
        b.wire_elem_bytes, b.wire_id_bits = int(fmt.elem_bytes), int(fmt.id_bits)
        return b


def measured_sort_traffic(n, form="single"):
    """HBM bytes of the radix-sort phase on an n-partition topic from the committed PMC summary (tools/gpu_session.sh TAG sort pmc)."""
    try:
        with open(TRAFFIC_FILE) as fh:
            for e in json.load(fh).get("entries", []):
                if e.get("kind") == "sort_phase" and e.get("partitions") == n and e.get("form", "multi") == form:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def measured_traffic(T, P, C, mode, algo):
    """HBM bytes per launch from the PMC passes (tools/pmc_probe.py + tools/pmc_parse.py), if a
    summary for exactly this workload is committed; rocprofv3 cannot wrap itself from inside here."""
    try:
        with open(TRAFFIC_FILE) as fh:
            t = json.load(fh)
        for e in t.get("entries", []):
            if e.get("kind") == "sort_phase":
                continue
            if (e["topics"], e["partitions"], e["consumers"], e["reset_mode"], e["algo"]) == (T, P, C, mode, algo):
                return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def _pmc_passes(probe_args, seconds=60):
    """Two child processes under rocprofv3 (--pmc FETCH_SIZE, then --pmc WRITE_SIZE: separate passes, as
    MI355X_MICROARCH.md's HBM section prescribes) over tools/pmc_probe.py -- a calibration kernel with known bytes, then
    the launches to measure -- summarised and calibrated by tools/pmc_parse.py.  rocprofv3 cannot wrap the process it is
    called from, hence the children.  None when rocprofv3 is missing or a pass fails or times out."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    # already under a profiler (rocprofv3 -- python bench.py ...)?  Its environment would follow the children: stand down
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    out = tempfile.mkdtemp(prefix="la_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    probe = [sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")] + probe_args
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            r = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d",
                                os.path.join(out, counter), "--"] + probe, cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=seconds)
            if r.returncode != 0:
                return None
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_parse.py"), os.path.join(out, "FETCH_SIZE"),
                            os.path.join(out, "WRITE_SIZE")], capture_output=True, text=True, timeout=60)
        d = json.loads(r.stdout)
        d["how"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child processes over "
                    "tools/pmc_probe.py on this box), calibrated on lag_kernel_vec2's known bytes (fetch x%.1f B/count, "
                    "write x%.1f B/count)" % (d["calibration"]["fetch_bytes_per_count"], d["calibration"]["write_bytes_per_count"]))
        return d
    except Exception:  # noqa: BLE001 -- a reported extra: the committed summary stands in
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic(reset_mode, algo, none_frac=None):
    """HBM bytes per launch of the target batch's kernels, measured now on this box (see _pmc_passes)."""
    d = _pmc_passes(["--reset-mode", reset_mode, "--algo", algo, "--launches", "3"] +
                    (["--none-frac", str(none_frac)] if none_frac is not None else []))
    if not d:
        return None
    rd = wr = 0.0
    for k, e in d["kernels"].items():
        if "wave_tile_packed_kernel" in k or "wave_tile_wide_kernel" in k:
            rd += e.get("fetch_bytes_calibrated", 0.0)
            wr += e.get("write_bytes_calibrated", 0.0)
    if rd <= 0 or wr <= 0:
        return None
    return {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "written_bytes": round(wr), "source": d["how"]}


def live_sort_traffic(n, form):
    """HBM bytes of the radix-sort phase of one n-partition topic, measured now: every pass kernel's mean bytes per
    launch times the 12 launches of a sort (the skipped passes' launches move next to nothing and are in the mean)."""
    if form == "multi":
        os.environ["LA_SORT_MULTIKERNEL"] = "1"
    try:
        d = _pmc_passes(["--topics", "0", "--large-partitions", str(n), "--large-consumers", "0"], seconds=90)
    finally:
        os.environ.pop("LA_SORT_MULTIKERNEL", None)
    if not d:
        return None
    # every launch of the sort phase's kernels in the probe (how many pass slots a sort launches depends on the caller's bounds
    # and on the plan since round 6), over the probe's two sorts: mean bytes per launch x launches / 2
    sorts_in_probe = 2                                           # (tools/pmc_probe.py runs the large topic twice)
    names = (("onesweep_pass_kernel", "onesweep_redo_kernel", "tie_scan_kernel", "tie_repair_kernel") if form == "single" else
             ("tile_count_kernel", "scan_group_sums_kernel", "scan_offsets_kernel", "tile_scatter_kernel", "tie_scan_kernel", "tie_repair_kernel"))
    rd = wr = 0.0
    for k, e in d["kernels"].items():
        if any(x in k for x in names):
            launches = float(e.get("dispatches", 0)) / sorts_in_probe
            rd += launches * e.get("fetch_bytes_calibrated", 0.0)
            wr += launches * e.get("write_bytes_calibrated", 0.0)
    if rd <= 0 or wr <= 0:
        return None
    return {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "written_bytes": round(wr), "source": d["how"]}


def kernel_name(max_p, max_c):
    if max_p <= 1024 and max_c <= 64:
        return "wave_tile_packed_kernel (+ the wide-record kernel over its deferred-tile list, empty here)"
    if (max_p <= 8192 and max_c <= 2048) or (max_p <= 16384 and max_c <= 1024):
        return "block_topic_kernel<%d> (one workgroup per topic)" % (16 if max_p > 8192 else 8)
    return "large path: greedy_rounds_kernel (one workgroup per topic) behind build_keys + onesweep_pass_kernel x passes"


FROZEN_FULL = os.path.join(ROOT, "tests", "golden", "oracle_frozen_full.json")   # tests/golden/make_golden.py --full


CACHE_PROOF_BYTES = 3 * 256 * (1 << 20)         # three Infinity Caches' worth between two touches of the same bytes
MAX_ROTATION = 24


def rotation_for(set_bytes):
    """Distinct resident copies of a batch that the timed calls of a leg rotate over so that no call finds its bytes in the
    256 MiB Infinity Cache: as many as put 768 MB between two touches of the same copy, at most 24."""
    return int(max(1, min(MAX_ROTATION, -(-CACHE_PROOF_BYTES // max(int(set_bytes), 1)))))


def timed_calls(torch, ctx, batch, stream, settle_ms, min_calls=10, max_calls=400, window_ms=60.0):
    """ms per la_assign_batch_device call at steady state: an untimed settle phase sized in time, then ONE pair of HIP events
    around a back-to-back block of calls on the stream they run on.  `batch` may be a LIST of batches over distinct resident
    copies of the same workload: the calls then take them in turn (see rotation_for)."""
    batches = list(batch) if isinstance(batch, (list, tuple)) else [batch]
    nb = len(batches)
    for b in batches:
        ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.assign_batch_device(batches[0], stream)
    e1.record()
    ctx.sync(stream)
    est = max(float(e0.elapsed_time(e1)), 1e-3)
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(max(1, min(50, int(5.0 / est)))):
            ctx.assign_batch_device(batches[i % nb], stream)
            i += 1
        ctx.sync(stream)
    calls = int(max(min_calls, min(max_calls, window_ms / est)))
    e0.record()
    for _ in range(calls):
        ctx.assign_batch_device(batches[i % nb], stream)
        i += 1
    e1.record()
    ctx.sync(stream)
    return float(e0.elapsed_time(e1)) / calls, calls


def run_configs(torch, N, ctx, dev, stream):
    """Every BASELINE.json configuration that runs on one GPU (configs[1..4]; configs[0] is the README triple on the CPU path),
    at full size, device-resident, auto.offset.reset=earliest: settled ms per call by HIP events, the 36 B/partition roofline
    reading, the dominant kernel, and bit-exactness of the LAST timed call's results -- cfg2b / cfg3 / cfg4 against the literal
    oracle run here, cfg5 (15 s of literal oracle) against oracle/round_form.py; all four also against the sha256 the literal
    oracle froze into tests/golden/oracle_frozen_full.json."""
    import hashlib
    from kafka_lag_based_assignor_amd import synth
    from oracle import oracle
    from oracle.round_form import round_form
    try:
        with open(FROZEN_FULL) as fh:
            frozen = json.load(fh)
    except (OSError, ValueError):
        frozen = {}
    out = {}
    for name in ("cfg2b", "cfg3", "cfg4", "cfg5"):
        try:
            w = synth.config(name)
            sh = DeviceShard(torch, N, dev, w, 0, w.n_topics, False, "auto")
            set_bytes = 36 * sh.n + 8 * sh.k
            copies = [sh] + [DeviceShard(torch, N, dev, w, 0, w.n_topics, False, "auto") for _ in range(rotation_for(set_bytes) - 1)]
            ms, calls = timed_calls(torch, ctx, [x.batch for x in copies], stream, settle_ms=40.0)
            if len(copies) > 1:                                     # the results read below are sets[0]'s: its call last
                ctx.assign_batch_device(sh.batch, stream)
                ctx.sync(stream)
            n = sh.n
            g_pid, g_rank = sh.out_pid[:n].cpu().numpy(), sh.out_rank[:n].cpu().numpy()
            g_tot = sh.out_total[: sh.k].cpu().numpy()
            lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
            t0 = time.perf_counter()
            if name == "cfg5":
                e_pid, e_rank, e_tot = round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
                against = "oracle/round_form.py (independent numpy round form)"
            else:
                e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
                against = "oracle/lag_oracle.c (literal per-step min)"
            cpu_s = time.perf_counter() - t0
            ok = bool(np.array_equal(g_pid, e_pid) and np.array_equal(g_rank, e_rank) and np.array_equal(g_tot, e_tot))
            h = hashlib.sha256()
            for a in (g_pid.astype("<i4"), g_rank.astype("<i4"), g_tot.astype("<i8")):
                h.update(np.ascontiguousarray(a).tobytes())
            fz = frozen.get("%s@1/earliest" % name, {}).get("sha256")
            ratio = synth.lag_ratio(g_tot, w.cons_off)
            out[name] = {
                "workload": "%d topic(s) x %d partitions x %d consumers" % (w.n_topics, w.max_partitions, w.max_consumers),
                "partitions": int(n), "ms_per_call": round(ms, 5), "calls_timed": calls,
                "value": round(n / (ms * 1e-3), 1), "frac": round(36.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "kernel": kernel_name(w.max_partitions, w.max_consumers),
                "bit_exact": ok, "against": against, "checker_seconds": round(cpu_s, 2),
                "sha256_matches_frozen_literal_oracle": (h.hexdigest() == fz) if fz else None,
                "lag_ratio_mean": round(float(ratio.mean()), 4),
                "rotation": {"sets": len(copies), "bytes_resident": int(len(copies) * set_bytes),
                             "cache_resident": bool(len(copies) * set_bytes < 2 * 256 * (1 << 20))},
            }
            del sh, copies
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            out[name] = {"error": str(exc)}
    # The same two single-topic configs as a consumer group that has CAUGHT UP would hand them over: 99 % of the partitions with
    # committed == end (lag 0).  Not a BASELINE row: the state big topics are in most of the time, and the one in which the rounds
    # behind the last lag only repeat the final order (round 6).
    caught = {}
    for name in ("cfg2b", "cfg5"):
        try:
            w0 = synth.config(name)
            rng = np.random.default_rng(99)
            idle = rng.random(w0.n_partitions) < 0.99
            committed = np.where(idle, w0.end, w0.committed)
            w = synth.Workload(name + " caught up", w0.n_topics, w0.part_off, w0.partition_id, w0.begin, w0.end, committed,
                               np.where(idle, 0, w0.lag), w0.cons_off, w0.cons_rank, w0.max_partitions, w0.max_consumers)
            sh = DeviceShard(torch, N, dev, w, 0, w.n_topics, False, "auto")
            set_bytes = 36 * sh.n + 8 * sh.k
            copies = [sh] + [DeviceShard(torch, N, dev, w, 0, w.n_topics, False, "auto") for _ in range(rotation_for(set_bytes) - 1)]
            ms, calls = timed_calls(torch, ctx, [x.batch for x in copies], stream, settle_ms=40.0)
            if len(copies) > 1:
                ctx.assign_batch_device(sh.batch, stream)
                ctx.sync(stream)
            n = sh.n
            got = (sh.out_pid[:n].cpu().numpy(), sh.out_rank[:n].cpu().numpy(), sh.out_total[: sh.k].cpu().numpy())
            lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
            exp = round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
            caught[name] = {"ms_per_call": round(ms, 5), "calls_timed": calls, "lagging_partitions": int((lag > 0).sum()),
                            "bit_exact": bool(all(np.array_equal(g, e) for g, e in zip(got, exp))),
                            "against": "oracle/round_form.py", "config_ms_per_call": out.get(name, {}).get("ms_per_call")}
            del sh, copies
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            caught[name] = {"error": str(exc)}
    caught["what"] = ("cfg2b / cfg5 with 99 % of the partitions caught up (committed == end): the rounds behind the last lag repeat "
                      "the final order instead of being played; config_ms_per_call = the same shape with every partition lagging")
    out["caught_up"] = caught
    out["what"] = ("every single-GPU BASELINE.json config at full size, device-resident, earliest mode: ms per "
                   "la_assign_batch_device call from one HIP-event pair around a settled back-to-back block of calls that ROTATE "
                   "over `rotation.sets` distinct resident copies of the config (inputs and result buffers; 768 MB between two "
                   "touches of the same copy, at most 24 copies: `cache_resident` says where that is not reached -- cfg2b is "
                   "360 KB and a dependent chain, not a bandwidth figure either way; the large path's sort scratch is the context's "
                   "and shared by the copies); frac = "
                   "36 B x partitions / time / 8 TB/s (cfg2b / cfg5 are ONE topic -- a dependent chain on one workgroup -- and "
                   "cfg3 is 1 000 small topics: launch / latency bound, the fraction says so)")
    return out


def _small_calls_c_abi():
    """The same comparison without an interpreter on either side: tools/latency_c.c (plain C99 against include/lagassign.h)
    times la_assign_batch_grouped / la_assign_batch at the C ABI and, with oracle/liblagoracle.so dlopen'ed beside it, the C
    oracle + a stable counting sort by member for the same call on one host core.  None when no C compiler is around."""
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    oracle_so = os.path.join(ROOT, "oracle", "liblagoracle.so")
    if cc is None or not os.path.exists(oracle_so):
        return None
    pkg = os.path.join(ROOT, "kafka_lag_based_assignor_amd")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "latency_c")
        try:
            subprocess.check_call([cc, "-O2", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "latency_c.c"),
                                   "-L" + pkg, "-llagassign", "-ldl", "-Wl,-rpath," + pkg, "-o", exe], timeout=120)
            out = subprocess.run([exe, "--json", oracle_so], capture_output=True, text=True, timeout=180)
            rows = json.loads(out.stdout)
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            return {"error": str(exc)}
        # ... and what a REAL rebalance pays (tools/cold_c.c): the same call after the process and the device idled for a second,
        # without and with la_wake 5 ms before it (what a host issues when it enters assign(), before its broker round trips)
        cold = None
        try:
            exe2 = os.path.join(d, "cold_c")
            subprocess.check_call([cc, "-O2", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cold_c.c"),
                                   "-L" + pkg, "-llagassign", "-ldl", "-Wl,-rpath," + pkg, "-o", exe2], timeout=120)
            out2 = subprocess.run([exe2, "--json", "100", "20", "4"], capture_output=True, text=True, timeout=120)
            cold = json.loads(out2.stdout)
            cold["what"] = ("tools/cold_c.c at the C ABI: la_assign_batch_grouped back to back, after 1 s of idle (a rebalance is "
                            "never back to back), and after 1 s of idle with la_wake 5 ms before the call; medians of 5")
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            cold = {"error": str(exc)}
    cross = next((r["partitions"] for r in rows if r["cpu_oracle_us"] > 0 and r["grouped_us"] < r["cpu_oracle_us"]), None)
    return {"rows": rows, "gpu_faster_from_partitions": cross, "cold": cold,
            "what": "tools/latency_c.c: median wall time of ONE la_assign_batch_grouped call (pageable buffers in, every member's list "
                    "out) at the C ABI, and of the C oracle + a stable counting sort by member for the same call on one host core"}


def run_small_calls(N, ctx):
    """What ONE real rebalance costs: la_assign_batch_grouped (assignment + every member's list, host buffers in, host buffers
    out) against the C oracle on the SAME call on one host core, so the crossover is on the record."""
    from kafka_lag_based_assignor_amd import synth
    from oracle import oracle
    rows = []
    for (t, p, c) in [(10, 10, 3), (40, 50, 5), (100, 100, 8), (1000, 50, 5), (1000, 256, 32)]:
        w = synth.make_uniform("lat", 20, t, p, c, "uniform40")
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        for _ in range(20):
            got = ctx.assign_batch_grouped(*a, c)
        reps = 200 if w.n_partitions <= 10000 else 40
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            got = ctx.assign_batch_grouped(*a, c)
            ts.append(time.perf_counter() - t0)
        tc = []
        for _ in range(max(5, reps // 8)):
            t0 = time.perf_counter()
            lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
            e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
            order = np.argsort(e_rank, kind="stable")              # the per-member lists (what the grouped call returns)
            tc.append(time.perf_counter() - t0)
        ok = bool(np.array_equal(got[2], e_pid[order]) and np.array_equal(got[3], e_tot))
        rows.append({"topics": t, "partitions_per_topic": p, "consumers": c, "partitions": int(w.n_partitions),
                     "gpu_call_us": round(float(np.median(ts)) * 1e6, 1), "cpu_oracle_us": round(float(np.median(tc)) * 1e6, 1),
                     "bit_exact": ok})
    cross = next((r["partitions"] for r in rows if r["gpu_call_us"] < r["cpu_oracle_us"]), None)
    return {"rows": rows, "gpu_faster_from_partitions": cross, "c_abi": _small_calls_c_abi(),
            "what": "median wall time of ONE la_assign_batch_grouped call (pageable host buffers in, every member's list out: "
                    "inputs packed into mapped host memory, the kernels read and write it in place -- ONE launch up to ~2 500 "
                    "partitions -- and the calling thread spins on a flag word) against the C oracle + a stable sort by member "
                    "for the same call on one host core (ctypes call overhead included on both sides; c_abi = the same without "
                    "an interpreter).  Below the crossover a rebalance is cheaper on the CPU: the GPU path's floor is ~20 us of "
                    "launch, PCIe round trips and completion"}


def run_sort_phase(torch, N, ctx, dev, n, reps, stream, form="single", live=False, ids=None):
    """The radix-sort phase of the large path on one topic of n partitions (no consumers: keys, sort, ids).  Times come
    from HIP events the library records around its phases (LA_FLAG_PROFILE / la_last_phase_times)."""
    w = sort_phase_workload(n, torch, dev, ids)
    torch.cuda.empty_cache()
    sh = DeviceShard(torch, N, dev, w, 0, 1, False, "auto",
                     flags=N.LA_FLAG_PROFILE | (N.LA_FLAG_SORT_MULTIKERNEL if form == "multi" else 0))
    ctx.assign_batch_device(sh.batch, stream)                      # scratch allocation, first touch
    ctx.sync(stream)
    sort_ms, keys_ms, ids_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.assign_batch_device(sh.batch, stream)
        ctx.sync(stream)
        t = ctx.last_phase_times()
        sort_ms.append(t.sort_ms); keys_ms.append(t.keys_ms); ids_ms.append(t.greedy_ms)
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    ms = float(np.mean(sort_ms))
    bytes_id, bytes_key = SORT_BYTES[form]
    keys_first = bool(getattr(t, "keys_first", 0))
    # a keys-first sort (no id passes) reads every key once more to find and repair the runs of equal lags
    algo_bytes = n * (t.id_passes * bytes_id + t.key_passes * bytes_key + (8 if keys_first else 0))
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    # sortedness of what came out: ids in (lag desc, id asc) order
    pid = sh.out_pid[:n].to(torch.int64)
    lag_dev = torch.from_numpy(w.lag).to(dev)
    perm_lag = torch.empty(n, device=dev, dtype=torch.int64)
    inv = torch.empty(n, device=dev, dtype=torch.int64)
    inv[sh.d["pid"].to(torch.int64)] = torch.arange(n, device=dev)
    perm_lag = lag_dev[inv[pid]]
    ok = bool(((perm_lag[:-1] > perm_lag[1:]) | ((perm_lag[:-1] == perm_lag[1:]) & (pid[:-1] < pid[1:]))).all()) if n > 1 else True
    del pid, lag_dev, perm_lag, inv
    tr = (live_sort_traffic(n, form) if live else None) or measured_sort_traffic(n, form)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr["hbm_bytes_per_launch"] if tr else None,
            "traffic_source": tr["source"] if tr else None,
            "kernel": ("onesweep_pass_kernel (one launch per active 8-bit pass: stable scatter with decoupled look-back)"
                       if form == "single" else "tile_count+scan_group_sums+scan_offsets+tile_scatter (every active 8-bit pass)"),
            "form": form, "rank": "ds_add_rtn" if ctx.device_features(0) & N.LA_FEATURE_ATOMIC_RANK else "wave match",
            "kernel_ms": round(ms, 4), "partitions": n, "id_passes": int(t.id_passes), "key_passes": int(t.key_passes),
            "ids": ids or os.environ.get("LA_SORT_IDS") or "random", "bytes_per_partition": round(algo_bytes / n, 1),
            "keys_first": keys_first, "redone": bool(getattr(t, "redone", 0)),
            "order_of_passes": ("key digits only, then tie_repair_kernel puts runs of equal lags in id order (the sample of the "
                                "lags showed no frequent one)" if keys_first else "id digits, then key digits (LSD)"),
            "traffic_bytes_per_partition": round(tr["hbm_bytes_per_launch"] / n, 1) if tr else None,
            "frac_moved": round(tr["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr else None,
            "algorithmic_bytes_per_launch": int(algo_bytes),
            "algorithmic_bytes": "%d B per id pass + %d B per key pass, per partition" % (bytes_id, bytes_key),
            "keys_ms": round(float(np.mean(keys_ms)), 4), "ids_ms": round(float(np.mean(ids_ms)), 4),
            "call_ms": round(call_ms, 4), "reps": reps, "sorted_ok": ok,
            "source": "measured in this run: HIP events inside liblagassign (la_last_phase_times), %d calls" % reps}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from kafka_lag_based_assignor_amd import _native as N
    from kafka_lag_based_assignor_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # LA_BENCH_BACKEND=gloo: a test hook for boxes with fewer GPUs than ranks -- the ranks share the devices there are
    # (RCCL refuses two ranks on one GPU) and the collectives run over gloo on host copies.  It exercises everything of
    # the N > 1 path but the RCCL transport: the la_plan_shards split, per-rank upload, padded all-gather, reassembly,
    # the oracle check of the gathered arrays, barrier and max-reduce.  Never a performance number.
    backend = os.environ.get("LA_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LA_BENCH_FORCE_DIST=1 runs the RCCL legs (init, barrier, all-gather, max-reduce) at world size 1 too, so that the
    # N>1 code path can be exercised under torchrun on a one-GPU box
    use_dist = world > 1 or os.environ.get("LA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import datetime
        patience = datetime.timedelta(minutes=30)                 # rank 0 measures its extras while the others wait
        if backend == "gloo":
            dist.init_process_group("gloo", timeout=patience)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=patience)

    def all_gather(dst, src):
        if backend == "gloo":                                     # through host copies (see above)
            h = torch.empty(dst.numel(), dtype=dst.dtype)
            dist.all_gather_into_tensor(h, src.cpu())
            dst.copy_(h)
        else:
            dist.all_gather_into_tensor(dst, src)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    ctx = N.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    latest = args.reset_mode == "latest"

    # ---- --phase sort: its own line -------------------------------------------------------------------------
    if args.phase == "sort":
        n = args.partitions or SORT_PHASE_PARTITIONS
        reps = max(1, min(args.steps, 20))
        sp = run_sort_phase(torch, N, ctx, dev, n, reps, stream, args.sort_form, live=not args.no_live_traffic)
        if rank == 0:
            print(json.dumps({
                "metric": "radix-sort phase of the large path, partitions sorted/sec", "value": round(n / (sp["kernel_ms"] * 1e-3), 1),
                "unit": "partitions/sec", "n_gpus": 1, "steps": reps, "warmup": 1, "ms_per_step": sp["kernel_ms"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                "config": {"workload": "1 topic x %d partitions x 0 consumers, uniform [0, 2^40) lags, randomly permuted ids" % n,
                           "phase": "sort"},
                "roofline": sp}))
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- workload: ONE generator for bench, tests and golden digests ----------------------------------------
    w, wname, dist_name = make_workload(args)
    T = w.n_topics
    # N > 1 defaults to the north star's multi-GPU workload: ONE batch sharded over the ranks, one all-gather per step
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    strong = scaling == "strong"
    if strong:
        # la_plan_shards (what la_create_multi uses itself); ncclAllGather needs equal counts: buffers padded to `cap`
        ranges, counts, cap = sharding.strong_plan(w.part_off, world)
        cap = (cap + 7) // 8 * 8              # every rank's block of the gathered buffer starts 16-byte aligned
        bounds = [r[0] for r in ranges] + [ranges[-1][1]]
        t0, t1 = ranges[rank]
        sh = DeviceShard(torch, N, dev, w, t0, t1, latest, args.algo, out_cap=cap)
        n_total = w.n_partitions
        gather = True
    else:
        t0, t1 = 0, T
        sh = DeviceShard(torch, N, dev, w, 0, T, latest, args.algo)
        counts = [sh.n] * world
        n_total = world * sh.n
        gather = bool(args.gather and use_dist)
    # The steps ROTATE over `--rotate` distinct resident copies of the (shard of the) batch -- inputs and result buffers at
    # different addresses, the same values -- so that what a step reads was last touched two steps (1.9 GB of traffic) ago and
    # cannot sit in the 256 MiB Infinity Cache (MI355X_MICROARCH.md: cache hits appear inside FETCH_SIZE).
    # sets[0] is `sh`: parity, lag_ratio and the host legs read its buffers, which hold the same assignment as every copy's.
    rot = max(1, int(args.rotate))
    sets = [sh] + [DeviceShard(torch, N, dev, w, t0, t1, latest, args.algo, out_cap=(sh.cap if strong else None))
                   for _ in range(rot - 1)]
    cap = sh.cap
    do_gather = gather and use_dist
    n_part = sh.n
    b = sh.batch
    # What the ONE all-gather of a step moves.  packed (default): the wire format of include/lagassign.h -- one unsigned
    # element per assigned partition, ((member rank + 1) << id_bits) | partition id, 2 bytes at the target and at cfg4 (the
    # library picks the width from the largest id and the member count: la_wire_format_for); int32: the two result arrays as
    # one [2, cap] int32 buffer (8 B per partition, round 3's form).
    packed = do_gather and args.wire in ("packed", "fused")
    unpack = packed and args.unpack and not args.no_unpack
    fused = False
    fmt = None
    if do_gather:
        max_id = int(w.partition_id.max()) if w.partition_id.size and int(w.partition_id.min()) >= 0 else -1
        n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
        fmt = N.wire_format_for(max_id, n_members)
        if packed:
            # raw bytes: RCCL (like NCCL) has no 16-bit integer type, and an all-gather does no arithmetic
            wire_send = torch.zeros(cap * fmt.elem_bytes, device=dev, dtype=torch.uint8)          # the tail beyond n_part stays padding
            wire_recv = torch.empty(world * cap * fmt.elem_bytes, device=dev, dtype=torch.uint8)  # [world][cap] elements
            if unpack:
                gathered = torch.empty(2 * world * cap, device=dev, dtype=torch.int32)  # [2][world][cap]: ids | ranks
        else:
            gathered = torch.empty(world * 2 * cap, device=dev, dtype=torch.int32)     # [world][2][cap]
    gather_bytes_per_rank = (cap * fmt.elem_bytes if packed else 2 * cap * 4) if do_gather else 0
    if packed and args.wire == "fused":
        # the assignment kernels write the wire elements themselves, straight into the gather's send buffer: no pack launch.
        # Every rank must agree (a rank whose shard is not eligible would make the step's shape differ): all or none.
        for x in sets:
            x.batch_wire = x.make_wire_batch(latest, args.algo, fmt, wire_send.data_ptr())
        ok = torch.tensor([1.0 if all(x.batch_wire is not None for x in sets) else 0.0], device="cpu" if backend == "gloo" else dev)
        if use_dist and world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        fused = bool(ok.item() == 1.0)

    def gather_step(cur, trio=None):
        """pack -> THE collective of a step -> unpack, on the results of the copy `cur` the step's kernels just wrote;
        trio[2..4] are recorded behind each phase when given."""
        if packed:
            if not fused:
                ctx.pack_results(n_part, cur.out_pid.data_ptr(), cur.out_rank.data_ptr(), fmt, wire_send.data_ptr(), stream)
            if trio:
                trio[2].record()
            all_gather(wire_recv, wire_send)
            if trio:
                trio[3].record()
            if unpack:
                ctx.unpack_results(world * cap, wire_recv.data_ptr(), fmt, gathered.data_ptr(),
                                   gathered.data_ptr() + 4 * world * cap, stream)
            if trio:
                trio[4].record()
        else:
            if trio:
                trio[2].record()
            all_gather(gathered, cur.out2)
            if trio:
                trio[3].record()
                trio[4].record()

    step_no = [0]

    def next_set():
        """the copy of the batch the next step works on: the copies in turn"""
        i = step_no[0] % rot
        step_no[0] += 1
        return sets[i]

    def step():
        cur = next_set()
        ctx.assign_batch_device(cur.batch_wire if fused else cur.batch, stream)
        if do_gather:
            gather_step(cur)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle (untimed, sized in time), warm-up (untimed, W steps) ----------------------------------------
    step()
    ctx.sync(stream)                                               # first touch: scratch, code objects
    settle_steps = 0
    t_settle = time.perf_counter()
    while True:
        # every rank runs the same number of settle steps (a collective sits in each): rank 0's clock decides
        go = torch.tensor([1.0 if (time.perf_counter() - t_settle) * 1e3 < args.settle_ms else 0.0],
                          device="cpu" if backend == "gloo" else dev)
        if use_dist and world > 1:
            dist.broadcast(go, 0)
        if go.item() == 0.0:
            break
        for _ in range(20):
            step()
        settle_steps += 20
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize on both sides --------------------
    # HIP events around a sample of the steps: an event record is a marker packet in the queue, ~1.5 us each, and a pair
    # around every one of the driver's 20 steps took 2 % off `value` (the wall clock over the K steps).  About 200 of the
    # steps of a long run, every other step of a short one (at least 10 samples).
    stride = max(1, args.steps // 200) if args.steps > 40 else (2 if args.steps >= 20 else 1)
    ev = {s: tuple(torch.cuda.Event(enable_timing=True) for _ in range(5)) for s in range(0, args.steps, stride)}
    barrier()
    t0c = time.perf_counter()
    for s in range(args.steps):
        trio = ev.get(s)
        if trio is None:
            step()
        else:
            trio[0].record()
            cur = next_set()
            ctx.assign_batch_device(cur.batch_wire if fused else cur.batch, stream)
            trio[1].record()
            if do_gather:
                gather_step(cur, trio)
    barrier()
    elapsed = time.perf_counter() - t0c
    ctx.sync(stream)
    if fused:
        # the timed steps wrote wire elements only: one plain call, so that the two int32 result arrays (and the totals) of
        # sets[0] hold the same assignment for the legs below that read them
        ctx.assign_batch_device(b, stream)
        ctx.sync(stream)

    # HIP-event durations on the stream the work runs on: the assign launch (the kernels of one step), then pack, the
    # collective, unpack
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev.values()]))
    pack_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev.values()])) if do_gather else 0.0
    gather_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev.values()])) if do_gather else 0.0
    unpack_ms = float(np.mean([e[3].elapsed_time(e[4]) for e in ev.values()])) if do_gather else 0.0

    cdev = "cpu" if backend == "gloo" else dev
    t = torch.tensor([elapsed, kern_ms, gather_ms, pack_ms, unpack_ms], device=cdev, dtype=torch.float64)
    per_rank = torch.zeros(2 * world, device=cdev, dtype=torch.float64)
    per_rank[rank] = kern_ms
    per_rank[world + rank] = gather_ms
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    elapsed, kern_ms_max, gather_ms_max = float(t[0].item()), float(t[1].item()), float(t[2].item())
    pack_ms_max, unpack_ms_max = float(t[3].item()), float(t[4].item())
    per_rank = per_rank.cpu().numpy()

    # strong scaling: the gathered buffer, stripped of its padding, is the global assignment -- checked below
    gathered_host = None
    if strong and rank == 0:
        if do_gather and packed and not unpack:
            # the map was left in the wire format: expand it here, after the timed region, with the library's own decoder
            # (and time the expansion: what a consumer that wants the two int32 arrays pays on top of a step)
            tmp = torch.empty(2 * world * cap, device=dev, dtype=torch.int32)
            ctx.unpack_results(world * cap, wire_recv.data_ptr(), fmt, tmp.data_ptr(), tmp.data_ptr() + 4 * world * cap, stream)
            ctx.sync(stream)
            u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            u0.record()
            for _ in range(10):
                ctx.unpack_results(world * cap, wire_recv.data_ptr(), fmt, tmp.data_ptr(), tmp.data_ptr() + 4 * world * cap, stream)
            u1.record()
            ctx.sync(stream)
            unpack_ms_max = float(u0.elapsed_time(u1)) / 10
            g = tmp.cpu().numpy().reshape(2, world * cap)
            gathered_host = (sharding.strip_padding(g[0], counts, cap), sharding.strip_padding(g[1], counts, cap))
        elif do_gather and packed:
            g = gathered.cpu().numpy().reshape(2, world * cap)
            gathered_host = (sharding.strip_padding(g[0], counts, cap), sharding.strip_padding(g[1], counts, cap))
        elif do_gather:
            g = gathered.cpu().numpy().reshape(world, 2, cap)
            gathered_host = (sharding.strip_padding(np.ascontiguousarray(g[:, 0, :]).reshape(-1), counts, cap),
                             sharding.strip_padding(np.ascontiguousarray(g[:, 1, :]).reshape(-1), counts, cap))
        else:
            gathered_host = (sh.out_pid[:n_part].cpu().numpy(), sh.out_rank[:n_part].cpu().numpy())

    if rank != 0:
        if use_dist:
            dist.barrier()                     # rank 0 is still measuring its extras and checking the oracle: leave together
            dist.destroy_process_group()
        return

    # ---- what one rebalance sees: the first call after a second of idle (rank 0; the other ranks wait at the barrier) ----
    torch.cuda.synchronize()
    time.sleep(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0 = time.perf_counter()
    e0.record()
    ctx.assign_batch_device(b, stream)
    e1.record()
    ctx.sync(stream)
    cold_wall = (time.perf_counter() - c0) * 1e3
    cold_ms = float(e0.elapsed_time(e1))
    cold = {"ms": round(cold_ms, 4), "wall_ms": round(cold_wall, 4), "value": round(n_part / (cold_ms * 1e-3), 1),
            "what": "ONE la_assign_batch_device call (HIP events around it; wall = enqueue + la_sync) after the GPU "
                    "idled for 1 s: clocks and power state as a rebalance finds them%s"
                    % (" (rank 0's shard, no gather)" if world > 1 else "")}

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed

    lens_p, lens_c = np.diff(w.part_off), np.diff(w.cons_off)
    P = int(lens_p.max()) if lens_p.size else 0
    C = int(lens_c.max()) if lens_c.size else 0
    uniform = bool(lens_p.size and (lens_p == P).all() and (lens_c == C).all())
    bpp = BYTES_PER_PARTITION[args.reset_mode]
    achieved = bpp * n_part / (kern_ms * 1e-3) / 1e9
    tr = None
    if uniform and world == 1 and wname == "target" and not args.no_live_traffic and not args.no_cpu_baseline:
        tr = live_traffic(args.reset_mode, args.algo)              # measured in this run (child processes, this box)
    if tr is None:
        tr = measured_traffic(b.n_topics, P, C, args.reset_mode, args.algo) if uniform else None
        if tr:
            tr = dict(tr, source="%s (committed summary of the same workload: the live PMC passes were skipped or failed)" % tr.get("source"))
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                "kernel": kernel_name(P, C),
                "kernel_ms": round(kern_ms, 4),
                "kernel_ms_source": "HIP events around %d of the %d timed launches, on the stream they run on (rank 0%s)"
                                    % (len(ev), args.steps, "; max over ranks %.4f" % kern_ms_max if world > 1 else ""),
                "algorithmic_bytes_per_partition": bpp,
                "algorithmic_bytes_per_launch": bpp * n_part,
                # the same bytes over the time of ONE call after a second of idle: the regime a real rebalance lives in
                "frac_cold": round(bpp * n_part / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    roofline["frac_rotating"] = roofline["frac"] if rot > 1 else None
    roofline["rotation"] = {"sets": rot, "bytes_resident": int(rot * (bpp * n_part + 8 * sh.k)),
                            "what": "the timed steps rotate over %d distinct resident copies of the batch (inputs and result buffers): "
                                    "`frac` / `frac_rotating` cannot owe anything to the 256 MiB Infinity Cache holding a previous "
                                    "step's bytes; frac_same_buffers is the round 1-4 form (every step on ONE copy)" % rot}
    if world == 1 and not args.no_cpu_baseline:
        try:
            sms, scalls = timed_calls(torch, ctx, sh.batch, stream, settle_ms=60.0)
            roofline["same_buffers_ms"] = round(sms, 4)
            roofline["frac_same_buffers"] = round(bpp * n_part / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if sh.bounds is not None:
                # the same batch without the marshaller's bounds: packed-record kernel + the wide-record kernel over its (empty)
                # deferred list -- what a caller that gives no hint (la_hint_next_call / LA_FLAG_BOUNDS) gets
                nms, ncalls = timed_calls(torch, ctx, [x.make_batch(latest, args.algo, bounds=False) for x in sets], stream,
                                          settle_ms=60.0)
                roofline["no_bounds_ms"] = round(nms, 4)
                roofline["no_bounds_frac"] = round(bpp * n_part / (nms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            ctx.assign_batch_device(b, stream)
            ctx.sync(stream)
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            roofline["same_buffers_error"] = str(exc)
    if world == 1 and not do_gather and not args.no_cpu_baseline and uniform:
        # what the N > 1 step's kernels cost with and without the fused wire output, measurable on one GPU: assignment + pack
        # (two int32 arrays written, read again, 2 B written) against the assignment kernels writing the wire elements themselves
        try:
            max_id = int(w.partition_id.max()) if w.partition_id.size and int(w.partition_id.min()) >= 0 else -1
            fmt1 = N.wire_format_for(max_id, int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0)
            # one wire buffer per resident copy: both forms rotate over the copies like the headline steps
            wbufs = [torch.zeros(max(n_part, 1) * int(fmt1.elem_bytes), device=dev, dtype=torch.uint8) for _ in sets]
            wbuf = wbufs[0]
            bws = [x.make_wire_batch(latest, args.algo, fmt1, wb.data_ptr()) for x, wb in zip(sets, wbufs)]
            bw = bws[0]
            if all(x is not None for x in bws):
                fms, fcalls = timed_calls(torch, ctx, bws, stream, settle_ms=60.0)
                ctx.assign_batch_device(bw, stream)
                ctx.sync(stream)
                got_w = wbuf.clone()

                def plain_and_pack(i):
                    x, wb = sets[i % len(sets)], wbufs[i % len(sets)]
                    ctx.assign_batch_device(x.batch, stream)
                    ctx.pack_results(n_part, x.out_pid.data_ptr(), x.out_rank.data_ptr(), fmt1, wb.data_ptr(), stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(21):
                    plain_and_pack(i)
                ctx.sync(stream)
                e0.record()
                for i in range(fcalls):
                    plain_and_pack(i)
                e1.record()
                plain_and_pack(0)                                   # wbuf holds sets[0]'s packed result for the comparison
                ctx.sync(stream)
                roofline["wire_out"] = {"fused_ms": round(fms, 4), "kernels_plus_pack_ms": round(float(e0.elapsed_time(e1)) / fcalls, 4),
                                        "elem_bytes": int(fmt1.elem_bytes), "id_bits": int(fmt1.id_bits),
                                        "same_wire_bytes": bool(torch.equal(got_w, wbuf)),
                                        "what": "the kernels of one N > 1 step on this GPU's batch: LA_FLAG_WIRE_OUT (the assignment kernels "
                                                "write the all-gather's 2-byte elements themselves) against assignment + la_pack_results_on"}
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            roofline["wire_out"] = {"error": str(exc)}
    if world > 1 or do_gather:
        roofline["per_rank_kernel_ms"] = [round(float(x), 4) for x in per_rank[:world]]
        roofline["per_rank_gather_ms"] = [round(float(x), 4) for x in per_rank[world:]]
    if tr:
        roofline["traffic_source"] = tr.get("source")
        roofline["traffic_read_bytes"], roofline["traffic_written_bytes"] = tr.get("read_bytes"), tr.get("written_bytes")
        # the SAME kernel time over the bytes that really crossed the fabric (PMC): the numerator of `frac` charges `begin`
        # for every partition (SURVEY 8d fixes 36 B) while the kernels read it only where there is no committed offset
        roofline["frac_moved"] = round(tr["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["moved_bytes_per_partition"] = round(tr["hbm_bytes_per_launch"] / max(n_part, 1), 2)
    # ... and the `latest` form of the same batch (28 B contract: no `begin` array exists at all), same settled regime
    if world == 1 and not latest and not args.no_cpu_baseline:
        try:
            lms, lcalls = timed_calls(torch, ctx, [x.make_batch(True, args.algo) for x in sets], stream, settle_ms=60.0)
            ctx.assign_batch_device(b, stream)                      # the result buffers hold the EARLIEST assignment again:
            ctx.sync(stream)                                        # lag_ratio, parity and the host legs below read them
            roofline["latest_mode"] = {"kernel_ms": round(lms, 4), "calls_timed": lcalls, "algorithmic_bytes_per_partition": 28,
                                       "frac": round(28.0 * n_part / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "value": round(n_part / (lms * 1e-3), 1),
                                       "rotation_sets": len(sets),
                                       "what": "auto.offset.reset=latest on the same resident copies, taken in turn: d_begin_off = NULL"}
        except Exception as exc:  # noqa: BLE001
            roofline["latest_mode"] = {"error": str(exc)}

    # ---- quality metric of BASELINE.json: max/min per-consumer total lag per topic (min clamped to 1) ----
    lag_ratio = None
    if uniform and C > 0:
        tot = sh.out_total[: b.n_topics * C].reshape(b.n_topics, C).to(torch.float64)
        ratio = tot.max(dim=1).values / tot.min(dim=1).values.clamp(min=1.0)
        lag_ratio = {"mean": round(float(ratio.mean()), 4), "p99": round(float(torch.quantile(ratio[:1000000], 0.99)), 4),
                     "max": round(float(ratio.max()), 4),
                     "what": "max/min per-consumer assigned lag per topic of the last step's assignment (README.md:54-69 quotes 1.10 for its example)"}

    # ---- parity + cpu_baseline (oracle; test infrastructure, timed on host cores, after the timed region) ----
    cpu = parity = host_leg = None
    if not args.no_cpu_baseline:
        from oracle import oracle
        if strong:
            g_pid, g_rank = gathered_host
            g_tot = None
        else:
            g_pid, g_rank = sh.out_pid[:n_part].cpu().numpy(), sh.out_rank[:n_part].cpu().numpy()
            g_tot = sh.out_total[: sh.k].cpu().numpy()
        # strong scaling: a slice of every rank's shard; otherwise the batch from its first topic on
        starts = [int(bounds[r]) for r in range(world)] if strong else [0]
        ends = [int(bounds[r + 1]) for r in range(world)] if strong else [T]
        budget = args.cpu_seconds / len(starts)
        done_total, spent = 0, 0.0
        ok = True
        for s0, s1 in zip(starts, ends):
            done, local = s0, 0.0
            chunk = max(1, min(s1 - s0, 2000))
            while done < s1 and local < budget:
                e = min(s1, done + chunk)
                p0, p1 = int(w.part_off[done]), int(w.part_off[e])
                k0, k1 = int(w.cons_off[done]), int(w.cons_off[e])
                c0 = time.perf_counter()
                lag = oracle.compute_lags(w.begin[p0:p1], w.end[p0:p1], w.committed[p0:p1], latest)
                e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off[done:e + 1] - p0, w.partition_id[p0:p1], lag,
                                                          w.cons_off[done:e + 1] - k0, w.cons_rank[k0:k1])
                local += time.perf_counter() - c0
                ok &= bool(np.array_equal(e_pid, g_pid[p0:p1]) and np.array_equal(e_rank, g_rank[p0:p1]) and
                           (g_tot is None or np.array_equal(e_tot, g_tot[k0:k1])))
                done_total += e - done
                done = e
            spent += local
        frozen_ok = None
        if world == 1 and wname == "target" and not latest and g_tot is not None:
            # ... and the WHOLE result against the committed digest of the literal oracle on this batch (tests/golden/oracle_frozen_full.json)
            try:
                import hashlib
                with open(FROZEN_FULL) as fh:
                    fz = json.load(fh).get("target@1/earliest", {}).get("sha256")
                h = hashlib.sha256()
                for a_ in (g_pid.astype("<i4"), g_rank.astype("<i4"), g_tot.astype("<i8")):
                    h.update(np.ascontiguousarray(a_).tobytes())
                frozen_ok = (h.hexdigest() == fz) if fz else None
            except (OSError, ValueError):
                frozen_ok = None
        parity = {"checked_topics": done_total, "bit_exact": ok, "sha256_matches_frozen_literal_oracle": frozen_ok,
                  "against": "oracle/lag_oracle.c (literal per-step min)%s" % (
                      "; the all-gathered global arrays, a slice of every rank's shard" if strong else "")}
        if world == 1:
            n_checked = sum(int(w.part_off[min(s1, s0 + done_total)] - w.part_off[s0]) for s0, s1 in zip(starts[:1], ends[:1]))
            cpu = {"value": round(n_checked / spent, 1) if spent > 0 else None, "unit": "partition-assignments/sec", "cores": 1,
                   "kind": "port",
                   "sample": "first %d of %d topics of the same batch, C oracle (oracle/lag_oracle.c, literal "
                             "per-step min), 1 thread, %.1f s" % (done_total, T, spent),
                   "host_cpus": os.cpu_count()}
            # the same batch through the host-buffer entry point (la_assign_batch: chunked H2D / kernels / D2H over the
            # context's lanes); reported beside the bench value, never as it
            try:
                mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
                a = (w.part_off, w.partition_id, None if latest else w.begin, w.end, w.committed, mode, w.cons_off, w.cons_rank)
                # what the Java / C++ hosts' marshalling loops hand over before every assign call (la_hint_next_call): the largest
                # end offset and partition id they walked past.  One-shot, so it is repeated before each timed call.
                hb = N.offset_bounds(None if latest else w.begin, w.end, w.committed, w.partition_id)
                reuse = ctx.assign_batch(*a)            # result buffers are the caller's and reused (a Java host's direct
                for o in reuse:                         # ByteBuffers): the first call touches them, the next are timed
                    o.fill(0)
                times = []
                for _ in range(3):
                    c0 = time.perf_counter()
                    ctx.hint_next_call(hb)
                    hp, hm, ht = ctx.assign_batch(*a, out=reuse)
                    times.append(time.perf_counter() - c0)
                dt = min(times)
                ref_p = g_pid if not strong else gathered_host[0]
                ref_m = g_rank if not strong else gathered_host[1]
                # the same call on pinned arrays (la_host_alloc: what the Java host's direct ByteBuffers are made of): the
                # library then needs no worker threads -- H2D, kernels and D2H on three streams chained by events
                pin = [None if x is None or isinstance(x, int) else ctx.host_alloc(x.shape, x.dtype) for x in a]
                for dst, src in zip(pin, a):
                    if dst is not None:
                        dst[...] = src
                pa = tuple(p if p is not None else x for p, x in zip(pin, a))
                pout = tuple(ctx.host_alloc(o.shape, o.dtype) for o in reuse)
                ptimes = []
                for _ in range(4):
                    c0 = time.perf_counter()
                    ctx.hint_next_call(hb)
                    pp, pm, pt = ctx.assign_batch(*pa, out=pout)
                    ptimes.append(time.perf_counter() - c0)
                pdt = min(ptimes[1:])
                pinned_pipe = ctx.last_pipeline()
                pinned_launches = ctx.last_launches()
                # the same call without the hint (the tile path's second, empty launch is back)
                utimes = []
                for _ in range(3):
                    c0 = time.perf_counter()
                    ctx.assign_batch(*pa, out=pout)
                    utimes.append(time.perf_counter() - c0)
                unhinted_ms, unhinted_launches = min(utimes[1:]) * 1e3, ctx.last_launches()
                pinned_ok = bool(np.array_equal(pp, ref_p) and np.array_equal(pm, ref_m)) and \
                    pinned_pipe in (N.LA_PIPELINE_STREAMS, N.LA_PIPELINE_MAPPED)
                # ... and the round-3 form on the same arrays: three copy streams (LA_NO_MAPPED_PIPELINE)
                os.environ["LA_NO_MAPPED_PIPELINE"] = "1"
                try:
                    s3 = []
                    for _ in range(3):
                        c0 = time.perf_counter()
                        ctx.hint_next_call(hb)
                        ctx.assign_batch(*pa, out=pout)
                        s3.append(time.perf_counter() - c0)
                    streams_ms = min(s3[1:]) * 1e3 if ctx.last_pipeline() == N.LA_PIPELINE_STREAMS else None
                finally:
                    os.environ.pop("LA_NO_MAPPED_PIPELINE", None)
                # the same call with `begin` handed over only where there is no committed offset (la_assign_batch_sparse): 20 B
                # instead of 28 B per partition cross the link
                sparse_ms = sparse_ok = None
                if not latest:
                    s_idx, s_val = N.sparse_begin(w.begin, w.committed)
                    p_idx, p_val = ctx.host_alloc(s_idx.shape, np.int64), ctx.host_alloc(s_val.shape, np.int64)
                    p_idx[...] = s_idx
                    p_val[...] = s_val
                    stimes = []
                    for _ in range(4):
                        c0 = time.perf_counter()
                        ctx.hint_next_call(hb)
                        sp_, sm_, st_ = ctx.assign_batch_sparse(pa[0], pa[1], pa[3], pa[4], mode, p_idx, p_val, pa[6], pa[7], out=pout)
                        stimes.append(time.perf_counter() - c0)
                    sparse_ms = min(stimes[1:])
                    sparse_ok = bool(np.array_equal(sp_, ref_p) and np.array_equal(sm_, ref_m)) and \
                        ctx.last_pipeline() in (N.LA_PIPELINE_STREAMS, N.LA_PIPELINE_MAPPED)
                # what the Java host really does with a batch: the ungrouped result stays on the device, every member's
                # list comes back grouped (la_group_last_by_member: a stable device sort by member rank, then the D2H)
                n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
                gout = (ctx.host_alloc((n_members + 1,), np.int64), ctx.host_alloc((w.n_partitions,), np.int32),
                        ctx.host_alloc((w.n_partitions,), np.int32))
                gtimes = []
                for _ in range(3):
                    c0 = time.perf_counter()
                    ctx.hint_next_call(hb)
                    ctx.assign_batch(*pa, keep_on_device=True, want_totals=False)
                    g_off, g_t, g_p = ctx.group_last_by_member(w.n_partitions, n_members, out=gout)
                    gtimes.append(time.perf_counter() - c0)
                gdt = min(gtimes[1:])
                # the reference's order: member by member, inside a member topic by topic in assignment order = a stable
                # sort of the ungrouped arrays by member rank (checked on the first two million entries of every member range)
                order = np.argsort(ref_m, kind="stable")
                first = int(np.searchsorted(ref_m[order], 0))          # entries of topics without consumers come first
                grouped_ok = bool(g_off[0] == first and g_off[-1] == w.n_partitions and
                                  np.array_equal(g_p, ref_p[order]) and
                                  np.array_equal(g_t, (np.searchsorted(w.part_off, order, side="right") - 1).astype(np.int32)))
                host_leg = {"ms": round(dt * 1e3, 2), "value": round(w.n_partitions / dt, 1), "unit": "partition-assignments/sec",
                            "pinned_ms": round(pdt * 1e3, 2), "pinned_value": round(w.n_partitions / pdt, 1),
                            "pinned_ms_all": [round(x * 1e3, 2) for x in ptimes], "pinned_bit_exact_and_three_streams": pinned_ok,
                            "hint": ({"max_end_offset": hb[0], "max_partition_id": hb[1], "pinned_launches": pinned_launches,
                                      "pinned_unhinted_ms": round(unhinted_ms, 2), "pinned_unhinted_launches": unhinted_launches,
                                      "what": "every timed call of this block is preceded by la_hint_next_call with what a marshalling "
                                              "loop knows (largest end offset, largest partition id), as the Java and C++ hosts do: "
                                              "the tile path is one launch; pinned_unhinted_* = the same call without it"}
                                     if hb else None),
                            "pinned_pipeline": {2: "three copy streams", 4: "mapped: the kernels read / write the pinned arrays in place, no copies"}.get(pinned_pipe, pinned_pipe),
                            "pinned_three_streams_ms": round(streams_ms, 2) if streams_ms else None,
                            "sparse_begin_ms": round(sparse_ms * 1e3, 2) if sparse_ms else None,
                            "sparse_begin_value": round(w.n_partitions / sparse_ms, 1) if sparse_ms else None,
                            "sparse_begin_bit_exact_and_three_streams": sparse_ok,
                            "sparse_begin_what": "la_assign_batch_sparse on pinned arrays: begin offsets only for the partitions "
                                                 "without a committed offset (%d of %d): 20 B + 16 B x 1 %% per partition over PCIe, "
                                                 "floor %.2f ms at 57.2 GB/s" % (
                                                     int((w.committed < 0).sum()), w.n_partitions,
                                                     (w.n_partitions * 20 + int((w.committed < 0).sum()) * 16) / 57.2e9 * 1e3),
                            "grouped_ms": round(gdt * 1e3, 2), "grouped_value": round(w.n_partitions / gdt, 1),
                            "grouped_what": "pinned arrays in, the ungrouped result stays on the device, every member's list back "
                                            "(la_assign_batch + la_group_last_by_member): the Java host's flow; best of the last 2 of 3",
                            "grouped_lists_equal_stable_sort_by_member": grouped_ok,
                            "h2d_floor_ms": round(w.n_partitions * (bpp - 8) / 57.2e9 * 1e3, 2),
                            "mapped_floor_ms": round((w.n_partitions * 20 + int((w.committed < 0).sum()) * (0 if latest else 8) +
                                                      w.cons_rank.size * 4) / 57.2e9 * 1e3, 2),
                            "mapped_floor": "what the kernels of the mapped form read over the link (end, committed, id; `begin` only where "
                                            "there is no committed offset; the consumer ranks) at the same 57.2 GB/s",

                            "h2d_floor": "the input bytes of the call at the 57.2 GB/s one pinned hipMemcpy sustains on this "
                                         "link (tools/pcie_probe.py, profiles/archive/r03_pcie_probe.txt): no host-buffer call can be faster",
                            "what": "one la_assign_batch call on pageable host buffers (results into reused, already touched "
                                    "buffers), PCIe copies included: best of 3; chunks of the batch overlap their H2D, kernels "
                                    "and D2H over the context's lanes",
                            "ms_all": [round(x * 1e3, 2) for x in times],
                            "pcie_gbs": round((w.n_partitions * bpp + w.cons_rank.size * 8) / dt / 1e9, 1),
                            "bit_exact_vs_device_path": bool(np.array_equal(hp, ref_p) and np.array_equal(hm, ref_m))}
            except Exception as exc:  # noqa: BLE001 -- a reported extra, not part of the contract
                host_leg = {"error": str(exc)}
        if not ok:
            print("PARITY FAILURE against the oracle", file=sys.stderr)

    # ---- every other BASELINE config, and what one small rebalance costs (rank 0, N = 1) ----------------------------
    configs = small_call = None
    if world == 1 and not args.no_cpu_baseline and not args.no_configs and wname == "target":
        try:
            small_call = run_small_calls(N, ctx)
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            small_call = {"error": str(exc)}
        configs = run_configs(torch, N, ctx, dev, stream)
        P_t, C_t = int(np.diff(w.part_off).max()), int(np.diff(w.cons_off).max())
        configs["target"] = {"workload": "%d topics x %d partitions x %d consumers" % (T, P_t, C_t), "partitions": int(n_part),
                             "ms_per_call": round(kern_ms, 5), "value": round(n_part / (kern_ms * 1e-3), 1),
                             "frac": roofline["frac"], "kernel": roofline["kernel"],
                             "bit_exact": parity["bit_exact"] if parity else None,
                             "sha256_matches_frozen_literal_oracle": parity.get("sha256_matches_frozen_literal_oracle") if parity else None,
                             "against": "oracle/lag_oracle.c on the first %d topics (the cpu_baseline leg's budget)" % parity["checked_topics"] if parity else None}

    # ---- the workload in which the whole 36 B/partition really moves: NO committed offset anywhere (a brand-new consumer
    # group with auto.offset.reset=earliest, Main.java:393-396): same shape, same drawn lags (begin = what committed would
    # have been); the whole batch is checked against the literal oracle, not a sample.
    if world == 1 and not args.no_cpu_baseline and wname == "target" and not latest and uniform and parity is not None:
        try:
            from kafka_lag_based_assignor_amd import synth
            wn = synth.config("target", none_frac=1.0)
            if not (np.array_equal(wn.lag, w.lag) and np.array_equal(wn.end, w.end) and np.array_equal(wn.partition_id, w.partition_id)):
                raise RuntimeError("the all-none workload does not carry the headline's drawn lags")
            hb_ = torch.from_numpy(wn.begin).to(dev)
            for x in sets:
                x.d["committed"].fill_(-1)
                x.d["begin"].copy_(hb_)
                x.out2.zero_()
                x.out_total.zero_()
            del hb_
            ams, acalls = timed_calls(torch, ctx, [x.batch for x in sets], stream, settle_ms=60.0)
            ctx.assign_batch_device(b, stream)
            ctx.sync(stream)
            # the literal oracle on the whole all-none batch (C = 32: ~2 s of one host core)
            c0 = time.perf_counter()
            n_lag = oracle.compute_lags(wn.begin, wn.end, wn.committed, False)
            e_p, e_m, e_t = oracle.assign_flat(wn.part_off, wn.partition_id, n_lag, wn.cons_off, wn.cons_rank)
            an_cpu_s = time.perf_counter() - c0
            a_pid, a_rank, a_tot = sh.out_pid[:n_part].cpu().numpy(), sh.out_rank[:n_part].cpu().numpy(), sh.out_total[: sh.k].cpu().numpy()
            same = bool(np.array_equal(n_lag, wn.lag) and np.array_equal(a_pid, e_p) and np.array_equal(a_rank, e_m) and
                        np.array_equal(a_tot, e_t))
            an_frozen = None
            try:
                import hashlib
                with open(FROZEN_FULL) as fh:
                    fz = json.load(fh).get("target@1/earliest/none=1", {}).get("sha256")
                h = hashlib.sha256()
                for a_ in (a_pid.astype("<i4"), a_rank.astype("<i4"), a_tot.astype("<i8")):
                    h.update(np.ascontiguousarray(a_).tobytes())
                an_frozen = (h.hexdigest() == fz) if fz else None
            except (OSError, ValueError):
                pass
            del e_p, e_m, e_t, n_lag, a_pid, a_rank, a_tot
            an = {"kernel_ms": round(ams, 4), "calls_timed": acalls, "rotation_sets": len(sets), "none_frac": 1.0,
                  "algorithmic_bytes_per_partition": 36,
                  "frac": round(36.0 * n_part / (ams * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "value": round(n_part / (ams * 1e-3), 1),
                  "bit_exact": same, "against": "oracle/lag_oracle.c on all %d partitions of the all-none batch (%.1f s)" % (n_part, an_cpu_s),
                  "sha256_matches_frozen_literal_oracle": an_frozen,
                  "what": "the target shape with NO committed offset at all (100 % fall back to `begin`, earliest; begin = what "
                          "committed would have been, so every lag is the drawn one): every byte of the 36 B contract is read or written"}
            if not args.no_live_traffic:
                trn = live_traffic(args.reset_mode, args.algo, none_frac=1.0)
                if trn:
                    an["traffic"] = trn["hbm_bytes_per_launch"]
                    an["moved_bytes_per_partition"] = round(trn["hbm_bytes_per_launch"] / max(n_part, 1), 2)
                    an["frac_moved"] = round(trn["hbm_bytes_per_launch"] / (ams * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    an["traffic_source"] = trn["source"]
            roofline["all_none"] = an
            if not same:
                print("PARITY FAILURE: the all-none workload's assignment differs from the headline's", file=sys.stderr)
                parity["bit_exact"] = False
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            roofline["all_none"] = {"error": str(exc)}

    # ---- the north star's other figure: the radix-sort phase against the HBM roofline, measured in this run --------
    sort_phase = None
    if not args.no_sort_phase:                                  # rank 0 (the other ranks wait at the closing barrier)
        try:
            for x in sets:
                if hasattr(x, "d"):
                    del x.d                                      # the batch is done with: make room
            del sets[1:]
            torch.cuda.empty_cache()
            sp = run_sort_phase(torch, N, ctx, dev, SORT_PHASE_PARTITIONS, 5, stream, args.sort_form,
                                live=world == 1 and not args.no_live_traffic and not args.no_cpu_baseline)
            sort_phase = {k: sp[k] for k in ("frac", "frac_moved", "achieved", "unit", "kernel", "form", "rank", "kernel_ms", "partitions", "keys_first",
                                             "redone", "order_of_passes",
                                             "id_passes", "ids", "algorithmic_bytes", "bytes_per_partition", "traffic_bytes_per_partition",
                                             "key_passes", "algorithmic_bytes_per_launch", "traffic", "traffic_source", "sorted_ok", "source")}
            # the round-1/2 workload beside it: ids as the affine permutation i -> (a*i + c) mod 2^k, whose id
            # passes alias their 256 output runs onto one set of memory channels
            if world == 1 and not args.no_cpu_baseline:
                torch.cuda.empty_cache()
                sa = run_sort_phase(torch, N, ctx, dev, SORT_PHASE_PARTITIONS, 3, stream, args.sort_form, live=False, ids="affine")
                sort_phase["affine_ids"] = {k: sa[k] for k in ("kernel_ms", "frac", "id_passes", "key_passes", "keys_first", "sorted_ok")}
                # ... and the round-3 order of passes (ids first) on the headline workload, same box, same process
                os.environ["LA_SORT_KEYS_FIRST"] = "0"
                try:
                    s0 = run_sort_phase(torch, N, ctx, dev, SORT_PHASE_PARTITIONS, 3, stream, args.sort_form, live=False)
                    sort_phase["ids_first"] = {k: s0[k] for k in ("kernel_ms", "frac", "id_passes", "key_passes", "keys_first", "bytes_per_partition", "sorted_ok")}
                finally:
                    os.environ.pop("LA_SORT_KEYS_FIRST", None)
        except Exception as exc:  # noqa: BLE001 -- a reported extra
            sort_phase = {"error": str(exc)}
    # the same figures inside `roofline`, where the driver's parsed block keeps them
    if sort_phase and "error" not in sort_phase:
        roofline["sort_phase"] = {k: sort_phase.get(k) for k in ("frac", "frac_moved", "kernel_ms", "partitions", "bytes_per_partition",
                                                                  "traffic_bytes_per_partition", "keys_first", "redone", "sorted_ok")}
    roofline["frac_cold_what"] = ("`frac` is the steady state of back-to-back steps over rotating copies; `frac_cold` is the same 36 B x "
                                  "partitions over the HIP-event time of ONE call after the GPU idled for 1 s (cold_call): clocks, power "
                                  "state and caches as a real rebalance finds them")

    line = {
        "metric": "partition-assignments/sec (whole node)",
        "value": round(value, 1),
        "unit": "partition-assignments/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "%s: %d topics x %d partitions x %d consumers%s, %s lags, shuffled partition ids, "
                               "1%% no committed offset, auto.offset.reset=%s (synth.config, SplitMix64)"
                               % (wname, T, P, C, " in all, sharded over %d GPUs by la_plan_shards" % world if strong
                                  else " per GPU", dist_name, args.reset_mode),
                   "topics": T, "partitions_per_topic": P, "consumers_per_topic": C,
                   "topics_on_rank0": int(b.n_topics), "gather": bool(do_gather), "algo": args.algo,
                   "collectives_per_step": 1 if do_gather else 0,
                   "collective": (("ONE all_gather_into_tensor of %d wire elements of %d bytes per rank (((member rank + 1) << %d) | "
                                   "partition id; %s%s), inside the timed region: %.4f ms by HIP events "
                                   "(max over ranks)" % (cap, fmt.elem_bytes, fmt.id_bits,
                                                         "written by the assignment kernels themselves (LA_FLAG_WIRE_OUT)" if fused else
                                                         "la_pack_results_on before it",
                                                         ", la_unpack_results_on over the gathered map after it" if unpack else
                                                         "; the gathered map stays in the wire format", gather_ms_max))
                                  if packed else
                                  ("ONE all_gather_into_tensor of the [2, %d] int32 result buffer (partition order | member "
                                   "rank) per step, inside the timed region: %.4f ms by HIP events (max over ranks)"
                                   % (cap, gather_ms_max))) if do_gather else None,
                   "wire": ({"format": "fused" if fused else ("packed" if packed else args.wire), "requested": args.wire,
                             "pack_launch": bool(packed and not fused),
                             "produced_by": ("the assignment kernels themselves (LA_FLAG_WIRE_OUT)" if fused else
                                             ("la_pack_results_on behind the assignment kernels" if packed else "no packing: two int32 arrays")),
                             "elem_bytes": int(fmt.elem_bytes) if packed else 8, "id_bits": int(fmt.id_bits) if packed else None,
                             "gather_bytes_per_rank": int(gather_bytes_per_rank), "unpacked_in_step": bool(unpack),
                             "pack_ms": round(pack_ms_max, 4), "gather_ms": round(gather_ms_max, 4), "unpack_ms": round(unpack_ms_max, 4),
                             "unpack_ms_source": ("inside the step" if unpack else "rank 0, 10 back-to-back expansions after the timed region") if packed else None,
                             "value_if_unpacked_in_step": (round(n_total / ((elapsed / args.steps) + unpack_ms_max * 1e-3), 1)
                                                           if (packed and not unpack) else None),
                             "kernels_ms": round(kern_ms_max, 4)} if do_gather else None),
                   "backend": ("rccl" if backend == "nccl" else "gloo, ranks sharing devices (test hook: not a performance number)") if use_dist else None,
                   "bounds_hint": ({"max_end_offset": sh.bounds[0], "max_partition_id": sh.bounds[1],
                                    "what": "LA_FLAG_BOUNDS: the caller's bounds on lags (the largest end offset) and ids prove that every tile "
                                            "packs into 64-bit records, so the tile path is ONE launch per step (no empty wide-record launch "
                                            "behind it); a violated bound is LA_EINVAL, never a different result"}
                                   if (sh.bounds is not None and not os.environ.get("LA_BENCH_NO_BOUNDS")) else None),
                   "rotate": rot, "settle_ms": args.settle_ms, "settle_steps": settle_steps},
        "roofline": roofline,
        "multi_gpu": {"ranks_in_this_run": world,
                      "note": ("this line is a one-GPU measurement: no N > 1 throughput of this code has been measured anywhere yet (the "
                               "boxes it was developed on have one GPU; the N > 1 path runs in the tests with 2-4 ranks sharing a GPU over "
                               "gloo and with RCCL at one rank); roofline.wire_out is what one GPU can say about the N > 1 step's kernels"
                               if world == 1 else
                               "strong scaling: ONE batch split over the ranks by la_plan_shards, one all-gather per step inside the timed region")},
        "cold_call_ms": cold["ms"],
        "cold_call": cold,
        "sort_phase": sort_phase,
        "lag_ratio": lag_ratio,
        "cpu_baseline": cpu,
        "parity": parity,
        "host_boundary": host_leg,
        "configs": configs,
        "small_call": small_call,
    }
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["bit_exact"]:
        sys.exit(2)


if __name__ == "__main__":
    main()
