#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_frozen.json: digests of the oracle's answers on the seeded
BASELINE.json workloads (kafka_lag_based_assignor_amd/synth.py is the single generator).

    python tests/golden/make_golden.py

The reference is Java and cannot be run in this image, so these are NOT reference outputs: they are
the outputs of oracle/ (which tests/test_oracle_golden.py pins on the reference's own known-answer
vectors in reference_vectors.json), frozen so that (a) the oracle cannot drift silently and (b) the
HIP path is also checked against committed fixtures, not only against an oracle built at test time.
A digest is sha256 over the little-endian bytes of out_partition | out_member_rank | out_total_lag.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [  # (config name, scale, reset mode)
    ("cfg1", 1.0, "lags"), ("cfg2a", 1.0, "lags"), ("cfg2b", 1.0, "latest"), ("cfg2b", 1.0, "earliest"),
    ("cfg3", 1.0, "latest"), ("cfg3", 1.0, "earliest"), ("cfg4", 0.01, "earliest"), ("cfg5", 1.0 / 64, "earliest"),
    ("target", 0.01, "latest"), ("target", 0.01, "earliest"),
    ("block_a", 1.0, "earliest"), ("block_b", 1.0, "latest"), ("block_c", 1.0, "earliest"),
]


# Full-size BASELINE configurations whose literal oracle run is too long for the per-round CPU suite (cfg5: 8.6e9 comparator
# steps, ~15 s): `make_golden.py --full` freezes them into oracle_frozen_full.json once; bench.py's `configs` block and the
# GPU tests check the HIP path against these digests (the GPU tests ALSO re-run the oracle).
FULL_CASES = [("cfg2b", 1.0, "earliest"), ("cfg3", 1.0, "earliest"), ("cfg4", 1.0, "earliest"), ("cfg5", 1.0, "earliest"),
              # round 6: the headline batch itself and its form without any committed offset (synth.config(..., none_frac=1.0))
              ("target", 1.0, "earliest"), ("target", 1.0, "earliest/none=1")]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    return h.hexdigest()


def case_key(name, scale, mode):
    return "%s@%g/%s" % (name, scale, mode)


def run_oracle(w, mode):
    from oracle import oracle
    if mode == "lags":
        lag = w.lag
    else:
        lag = oracle.compute_lags(w.begin, w.end, w.committed, mode == "latest")
    return oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)


def main():
    from kafka_lag_based_assignor_amd import synth
    out = {}
    full = "--full" in sys.argv
    for name, scale, mode in (FULL_CASES if full else CASES):
        none_frac = float(mode.split("none=")[1]) if "none=" in mode else None
        w = synth.config(name, scale, none_frac=none_frac) if none_frac is not None else synth.config(name, scale)
        p, m, t = run_oracle(w, mode.split("/")[0])
        ratio = synth.lag_ratio(t, w.cons_off)
        out[case_key(name, scale, mode)] = {
            "n_topics": int(w.n_topics), "n_partitions": int(w.n_partitions), "n_consumers": int(w.cons_rank.size),
            "inputs_sha256": digest(w.part_off, w.partition_id, w.begin, w.end, w.committed, w.cons_off, w.cons_rank),
            "sha256": digest(p.astype(np.int32), m.astype(np.int32), t.astype(np.int64)),
            "first_partitions": p[:8].astype(int).tolist(), "first_members": m[:8].astype(int).tolist(),
            "lag_ratio_max": float(ratio.max()), "lag_ratio_mean": float(ratio.mean()),
        }
        print(case_key(name, scale, mode), out[case_key(name, scale, mode)]["sha256"][:16], file=sys.stderr)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_frozen_full.json" if full else "oracle_frozen.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")


if __name__ == "__main__":
    main()
