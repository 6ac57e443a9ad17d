"""bench.py's N > 1 path on a box with ONE GPU.

* gloo hook (LA_BENCH_BACKEND=gloo): two to four ranks share the GPU and the collectives run over gloo on host copies -- RCCL
  refuses two ranks on one device.  Everything of the strong-scaling form but the RCCL transport is exercised on real
  kernels: the la_plan_shards split, per-rank upload of a shard, the [2, cap] result buffer padded to the largest shard,
  the ONE all-gather per step inside the timed region, reassembly, the oracle check of the GATHERED global arrays on
  rank 0, barrier + max-reduce.  Not a performance number.
* RCCL leg (LA_BENCH_FORCE_DIST=1, one rank): backend "nccl" = RCCL init, barrier, the all-gather and the reductions on the
  device, so that the driver's GPU test record says something about the transport the 8-GPU run will use."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(n, extra, env_extra):
    env = dict(os.environ, **env_extra)
    port = 29600 + os.getpid() % 300 + n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5",
           "--warmup", "2", "--settle-ms", "5", "--no-sort-phase"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


GLOO = {"LA_BENCH_BACKEND": "gloo"}


@pytest.mark.timeout(1200)
def test_default_at_two_ranks_is_the_north_star_workload():
    """A bare `torchrun ... bench.py --gpus 2` -- what the driver's SCALE run issues -- measures ONE 100 000-topic target
    batch split over the ranks with one collective per step, not per-rank replicas."""
    d = _run(2, ["--cpu-seconds", "4"], GLOO)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["gather"] is True and d["config"]["collectives_per_step"] == 1
    assert d["config"]["topics"] == 100000 and d["config"]["partitions_per_topic"] == 256
    assert d["config"]["topics_on_rank0"] == 50000
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] > 0
    assert len(d["roofline"]["per_rank_kernel_ms"]) == 2 and min(d["roofline"]["per_rank_kernel_ms"]) > 0
    assert d["cold_call_ms"] > 0                                 # rank 0's extras survive at N > 1
    # the default wire format: 2 bytes per assigned partition (ids < 256, 32 members), written by the assignment kernels themselves
    # (round 5: LA_FLAG_WIRE_OUT, no pack launch in the step); the map stays packed on every rank and the expansion is timed beside
    # the step
    wire = d["config"]["wire"]
    assert wire["format"] == "fused" and wire["pack_launch"] is False and wire["pack_ms"] < 0.02
    assert wire["elem_bytes"] == 2 and wire["id_bits"] == 8 and wire["unpacked_in_step"] is False
    assert wire["unpack_ms"] > 0 and 0 < wire["value_if_unpacked_in_step"] < d["value"]
    assert wire["gather_bytes_per_rank"] == 2 * 12800000


@pytest.mark.timeout(1200)
def test_strong_scaling_two_ranks_cfg4():
    d = _run(2, ["--scaling", "strong", "--workload", "cfg4"], GLOO)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gather"] is True
    assert d["config"]["topics"] == 100000 and d["config"]["topics_on_rank0"] == 50000      # la_plan_shards: equal halves
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000     # the gathered global arrays
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.timeout(1200)
def test_strong_scaling_four_ranks_cfg4():
    d = _run(4, ["--workload", "cfg4"], GLOO)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["topics_on_rank0"] == 25000
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000
    assert len(d["roofline"]["per_rank_gather_ms"]) == 4
    assert d["config"]["wire"]["elem_bytes"] == 2 and d["config"]["wire"]["gather_bytes_per_rank"] == 2 * 1600000


@pytest.mark.timeout(1200)
def test_strong_scaling_three_ranks_other_wire_forms():
    """The round-3 form ([2, cap] int32) and the packed form expanded inside the step: same global arrays (rank 0's oracle)."""
    d = _run(3, ["--workload", "cfg4", "--wire", "int32"], GLOO)
    assert d["config"]["wire"]["format"] == "int32" and d["config"]["wire"]["elem_bytes"] == 8
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000 and d["config"]["collectives_per_step"] == 1
    d = _run(3, ["--workload", "cfg4", "--unpack"], GLOO)
    assert d["config"]["wire"]["unpacked_in_step"] is True and d["config"]["wire"]["elem_bytes"] == 2
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000
    d = _run(3, ["--workload", "cfg4", "--wire", "packed"], GLOO)                 # round 4's form: a pack launch behind the kernels
    assert d["config"]["wire"]["format"] == "packed" and d["config"]["wire"]["pack_launch"] is True and d["config"]["wire"]["pack_ms"] > 0
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000


@pytest.mark.timeout(1200)
def test_weak_scaling_three_ranks_with_gather():
    d = _run(3, ["--scaling", "weak", "--workload", "cfg3", "--gather"], GLOO)
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["config"]["gather"] is True
    assert d["config"]["topics_on_rank0"] == 1000 and d["parity"]["bit_exact"] is True


@pytest.mark.timeout(1200)
def test_rccl_leg_one_rank_strong_cfg4():
    """backend nccl (= RCCL) on the device: init, barrier, the single all-gather of the packed result buffer inside the
    timed region, max- and sum-reductions; the gathered arrays are what the oracle checks."""
    d = _run(1, ["--scaling", "strong", "--workload", "cfg4"], {"LA_BENCH_FORCE_DIST": "1"})
    assert d["config"]["backend"] == "rccl" and d["config"]["gather"] is True
    assert d["config"]["collectives_per_step"] == 1 and d["scaling"] == "strong"
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000
    assert d["roofline"]["per_rank_gather_ms"][0] > 0
    assert d["config"]["wire"]["format"] == "fused" and d["config"]["wire"]["elem_bytes"] == 2      # bytes through RCCL (ncclUint8)
