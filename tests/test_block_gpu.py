"""Block path (csrc/la_block.hip: one workgroup per topic, up to 8 192 x 2 048 / 16 384 x 1 024) through the C ABI, bit-exact against the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import sharding, synth
from oracle import oracle
from oracle.round_form import round_form
from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


# ---- ADVICE r3 ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [1, 7, 64])
def test_block_path_with_exactly_np_cap_partitions(ctx, c):
    """16 384 partitions x <= 64 consumers: the E = 16 class filled to the last slot; the one-wavefront slots greedy reads
    s_key[P] as its zero slot, which has its own 16 bytes now (it used to alias s_tot[0])."""
    rng = np.random.default_rng(c)
    p = 16384
    part_off = np.array([0, p, 2 * p], np.int64)
    cons_off = np.array([0, c, 2 * c], np.int64)
    pid = np.concatenate([rng.permutation(p), rng.permutation(p)]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, 2 * p).astype(np.int64)
    lag[p:] = rng.integers(0, 3, p)                                                # heavy ties in the second topic
    ranks = np.concatenate([np.sort(rng.choice(500, c, replace=False)) for _ in range(2)]).astype(np.int32)
    got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    for g, e, what in zip(got, exp, ("order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


@pytest.mark.parametrize("topics,p,c", [
    (1, 513, 3), (3, 2049, 70), (2, 4097, 200), (1, 8193, 128), (2, 10000, 128), (1, 12289, 256), (40, 700, 128), (40, 1500, 256),
    (600, 300, 128), (600, 300, 200), (5, 9000, 65), (3, 5000, 129),
])
def test_block_sort_skips_sentinel_halves_and_small_multi_wave_greedy(ctx, topics, p, c):
    """Round 4, block path: (i) the records of a topic go to as few wavefronts as hold them and the packed sort skips every
    merge whose upper half is all sentinels -- partition counts just above a class border (513, 2 049, 4 097, 8 193) and in
    the middle of one; (ii) 65 .. 256 consumers with packed bins: one bin per lane on 2 / 4 wavefronts, the others leave the
    workgroup (128 bins only in launches of up to 512 topics: 600 topics take the one-wavefront form).  Mixed lag kinds (the
    "full" topics do not pack: they take the 96-bit forms of both steps), against the oracle."""
    w = _batch_of([(p + (i % 3), c - (i % 2)) for i in range(topics)], 1000 * topics + p + c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(_device_call(ctx, w), exp)


def test_block_sort_forms_in_fresh_processes():
    """The block path sorts packed records by digits inside the workgroup (block_sort_radix: ranks from returning LDS
    atomics) and with the bitonic network where the device lacks LA_FEATURE_ATOMIC_RANK or LA_BLOCK_RADIX says so; the
    choice is made once per process.  Every form, the same topics (all five size classes, partition counts at and between
    the class borders, packed and 96-bit records), the oracle's result."""
    import subprocess
    import sys
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N
from oracle import oracle
import gpu_helpers as t
ctx = N.Context(0)
for topics, p, c in ((3, 100, 65), (2, 512, 3), (2, 513, 70), (2, 2048, 256), (1, 2049, 300), (2, 4096, 1000), (1, 4097, 17),
                     (1, 8192, 2048), (1, 8193, 5), (1, 10000, 128), (1, 16384, 1024), (300, 130, 70), (7, 1, 65)):
    w = t._batch_of([(p - (i %% 2), c) for i in range(topics)], 77 * topics + p + c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    t._same3(t._device_call(ctx, w), exp, what=str((topics, p, c)))
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("0", "1", "2", "3"):                              # 3: topics below the size threshold too
        env = dict(os.environ, LA_BLOCK_RADIX=mode)
        out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"))], env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, (mode, out.stdout[-1500:], out.stderr[-1500:])


# ---- block_sort_radix, a padding sentinel against the record whose examined bits are all ones ------------
def test_block_radix_sentinel_against_all_ones_record_in_fresh_process():
    """The workgroup's digit sort pads with all-ones sentinels and looks at ceil((lbw + sh) / 8) digits.  When lbw + sh is a
    multiple of 8, the record (lag 0, id 2^sh - 1) has the sentinel's digits; a sentinel of an earlier wavefront may then sort
    before it.  P is not a multiple of 64 x 8, the special record sits in the second wavefront, lbw + sh in {8, 16, 24}; every
    topic through the digits (LA_BLOCK_RADIX=3, read once per process), against the oracle."""
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle
import gpu_helpers as t
ctx = N.Context(0)
rng = np.random.default_rng(5)
for P, C, sh, lbw in ((924, 65, 10, 6), (924, 70, 10, 14), (924, 65, 4, 4), (1500, 3, 11, 5), (3000, 100, 12, 12), (9000, 8, 14, 10),
                      (924, 65, 10, 7)):
    for at in (64, 100, 127, P - 1):
        ids = rng.permutation((1 << sh) - 1)[:P] if (1 << sh) - 1 >= P else rng.integers(0, (1 << sh) - 1, P)
        ids = ids.astype(np.int32)
        lag = rng.integers(0, 1 << lbw, P).astype(np.int64)
        lag[0] = (1 << lbw) - 1                       # the OR of the lags has lbw bits
        ids[at] = (1 << sh) - 1                       # the record whose lbw + sh bits are all ones ...
        lag[at] = 0                                   # ... : lag_max - 0 = all ones, id all ones
        part_off = np.array([0, P], np.int64); cons_off = np.array([0, C], np.int64)
        ranks = np.arange(C, dtype=np.int32) * 2
        w = synth.Workload("s", 1, part_off, ids, np.zeros(P, np.int64), lag.copy(), np.zeros(P, np.int64), lag, cons_off, ranks, P, C)
        exp = oracle.assign_flat(part_off, ids, lag, cons_off, ranks)
        t._same3(t._device_call(ctx, w), exp, what=str((P, C, sh, lbw, at)))
print("ok")
"""
    env = dict(os.environ, LA_BLOCK_RADIX="3")
    out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


@pytest.mark.parametrize("P,C", [(10000, 128), (1100, 65), (2049, 100), (4097, 129), (8193, 200), (16384, 256), (300, 70), (5000, 255)])
@pytest.mark.parametrize("kind", ["u40", "bigties", "zero", "pareto", "u20", "nearties"])
def test_block_greedy_through_32_bit_keys(ctx, P, C, kind):
    """greedy_one_wave_key32 against the literal oracle: uniform 40-bit lags (bits dropped from the key, shared truncated totals
    rare), many EQUAL large lags (every round meets tied totals with bits dropped: the exact re-ordering runs), lags that differ
    only below the dropped bits, all-zero and small lags (drop == 0: the key is exact, memberId breaks the ties), a Pareto tail."""
    import gpu_helpers as t4
    rng = np.random.default_rng(P + C)
    if kind == "u40":
        lag = rng.integers(0, 1 << 40, P)
    elif kind == "bigties":
        lag = (1 << 39) + rng.integers(0, 3, P) * (1 << 20)
    elif kind == "nearties":
        lag = (1 << 41) + rng.integers(0, 64, P)
    elif kind == "zero":
        lag = np.zeros(P, np.int64)
    elif kind == "u20":
        lag = rng.integers(0, 1 << 20, P)
    else:
        lag = np.floor(np.minimum(float(1 << 40), 1000.0 * (1.0 - rng.random(P)) ** (-1.0 / 1.5))).astype(np.int64)
    w = _one_topic(P, C, lag, P * 7 + C)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(t4._device_call(ctx, w), exp, "%s %d x %d" % (kind, P, C))


def test_block_greedy_forms_agree_in_a_fresh_process():
    """LA_BLOCK_KEY32=0 (the 64-bit bins through the networks, rounds 3-4) and =2 (32-bit keys for 256 bins too) give what the
    default gives: the oracle's assignment.  With bigties / nearties lags the exact re-ordering of tied rounds runs in mode 2."""
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N
from oracle import oracle
import gpu_helpers as t4, gpu_helpers as t5
ctx = N.Context(0)
rng = np.random.default_rng(3)
for (P, C) in ((10000, 128), (3000, 200), (16000, 256), (1500, 66)):
    for lag in (rng.integers(0, 1 << 40, P), (1 << 39) + rng.integers(0, 3, P) * (1 << 20), (1 << 41) + rng.integers(0, 64, P)):
        w = t5._one_topic(P, C, lag, P + C)
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        t4._same3(t4._device_call(ctx, w), exp, what=str((P, C)))
print("ok")
"""
    for mode in ("0", "2", "dense0"):                              # 2: the 32-bit-key form for 256 bins as well (default: 128 only)
        env = dict(os.environ, LA_BLOCK_KEY32=mode)
        if mode == "dense0":                                       # ... and without the one-scatter placement of dense ids in front of the digits
            env = dict(os.environ, LA_BLOCK_DENSE_IDS="0")
        out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (mode, out.stdout[-1500:], out.stderr[-1500:])


# ---- the rounds behind a topic's last lag (round 6: zero_tail_rounds, la_block.hip) -------------------------------------------------
@pytest.mark.parametrize("p,c", [(8000, 16), (5000, 50), (3000, 64), (1500, 1), (10000, 128), (8000, 100), (6000, 200), (16384, 64), (2500, 7)])
@pytest.mark.parametrize("nonzero", ["none", "one", "c-1", "c", "c+1", "mid", "p-c", "p-1"])
def test_block_rounds_behind_the_last_lag_repeat_the_final_order(ctx, p, c, nonzero):
    """A consumer group that has caught up on most of a topic's partitions: the sorted lags end in zeros, and from the first round
    whose first lag is zero on every round repeats that round's winners (the two one-wavefront greedy forms that leave their
    winners in the slots: up to 64 consumers, and 65 - 256 through 32-bit keys).  The zeros beginning at, before and behind a
    round's border; alone, many topics in one launch, and through the host entry."""
    k = {"none": 0, "one": 1, "c-1": c - 1, "c": c, "c+1": c + 1, "mid": (p // (2 * c)) * c + c // 2 + 1, "p-c": p - c, "p-1": p - 1}[nonzero]
    k = max(0, min(k, p))
    rng = np.random.default_rng(p * 7 + c + k)
    t = 5
    part_off = np.arange(t + 1, dtype=np.int64) * p
    cons_off = np.arange(t + 1, dtype=np.int64) * c
    lag = np.zeros(t * p, np.int64)
    for i in range(t):
        kk = k if i != 3 else max(0, k - 1)                              # (one topic a partition off)
        seg = np.zeros(p, np.int64)
        seg[:kk] = rng.integers(1, (1 << 30) if i % 2 else 40, kk)      # wide lags / heavy ties before the zeros
        lag[i * p:(i + 1) * p] = rng.permutation(seg)
    pid = np.concatenate([rng.permutation(p) for _ in range(t)]).astype(np.int32)
    ranks = np.concatenate([np.sort(rng.choice(3 * c + 5, c, replace=False)) for _ in range(t)]).astype(np.int32)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    w = synth.Workload("caught up", t, part_off, pid, np.zeros(t * p, np.int64), lag.copy(), np.zeros(t * p, np.int64), lag, cons_off, ranks, p, c)
    _same3(_device_call(ctx, w), exp, "device entry")
    _same3(ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "host entry")
