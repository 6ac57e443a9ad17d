"""bench.py's N > 1 path on a box with ONE GPU: two ranks share it and the collectives run over gloo on host copies
(LA_BENCH_BACKEND=gloo, a test hook -- RCCL refuses two ranks on one device).  Everything of the strong-scaling form but
the RCCL transport is exercised on real kernels: the la_plan_shards split, per-rank upload of a shard, result buffers
padded to the largest shard, the all-gather inside the timed region, reassembly, the oracle check of the GATHERED global
arrays on rank 0, barrier + max-reduce.  Not a performance number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(n, extra):
    env = dict(os.environ, LA_BENCH_BACKEND="gloo")
    port = 29600 + os.getpid() % 300 + n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5",
           "--warmup", "2", "--settle-ms", "5", "--no-sort-phase"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_strong_scaling_two_ranks_cfg4():
    d = _run(2, ["--scaling", "strong", "--workload", "cfg4"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gather"] is True
    assert d["config"]["topics"] == 100000 and d["config"]["topics_on_rank0"] == 50000      # la_plan_shards: equal halves
    assert d["parity"]["bit_exact"] is True and d["parity"]["checked_topics"] == 100000     # the gathered global arrays
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.timeout(1200)
def test_weak_scaling_three_ranks_with_gather():
    d = _run(3, ["--workload", "cfg3", "--gather"])
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["config"]["gather"] is True
    assert d["config"]["topics_on_rank0"] == 1000 and d["parity"]["bit_exact"] is True
