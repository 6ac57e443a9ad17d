"""Multi-GPU layout: one process per GPU, topics sharded across ranks.

``assignTopic`` reads and writes only its own topic's bins (Main.java:216-225), so topics are
independent units: each rank runs the whole hot path on a contiguous range of topics and
there is NO collective on the data path.  Reassembling the global (partition -> member) map
on every rank -- the north star's "single RCCL all-gather over xGMI" -- is an optional output
step (``gather_results``): with ``torch.distributed`` backend "nccl" it is RCCL, with "gloo"
it runs on CPU (tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_bounds(part_off: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous topic ranges [t0, t1) per rank, balanced by partition count (the kernels'
    cost is per partition).  Every topic lands in exactly one range; ranges may be empty.

    This IS the library's planner (``la_plan_shards`` of include/lagassign.h, pure host code): the split a
    one-process-per-GPU launcher makes here is the split a multi-device context (``la_create_multi``) makes
    inside ``la_assign_batch``."""
    try:
        from . import _native
        b = _native.plan_shards(part_off, world_size)
    except (OSError, ImportError):
        # liblagassign.so (and with it the HIP runtime) cannot be loaded on this host -- a launcher node, a gloo-only
        # test: the planner is pure host arithmetic, so the same formula is restated here.  tests/test_sharding_gloo.py
        # asserts that the two agree.
        b = plan_shards_numpy(part_off, world_size)
    return [(int(b[r]), int(b[r + 1])) for r in range(world_size)]


def plan_shards_numpy(part_off: np.ndarray, n_shards: int) -> np.ndarray:
    """la_plan_shards restated: bounds[r] = first topic boundary at or after r/n_shards of the partitions
    (target = base + (total // S) * r + (total % S) * r // S), searched from the previous bound on."""
    po = np.ascontiguousarray(part_off, dtype=np.int64)
    t = po.size - 1
    if t < 0 or n_shards < 1 or np.any(np.diff(po) < 0):
        raise ValueError("bad offsets or shard count")
    base, total = int(po[0]), int(po[t] - po[0])
    bounds = np.zeros(n_shards + 1, dtype=np.int32)
    for r in range(1, n_shards):
        target = base + (total // n_shards) * r + (total % n_shards) * r // n_shards
        lo = int(bounds[r - 1])
        bounds[r] = min(t, lo + int(np.searchsorted(po[lo:], target, side="left")))
    bounds[n_shards] = t
    return bounds


def shard_slices(part_off: np.ndarray, cons_off: np.ndarray, t0: int, t1: int):
    """(local part_off, local cons_off, partition slice, consumer slice) of topics [t0, t1)."""
    po = np.asarray(part_off[t0:t1 + 1], dtype=np.int64)
    co = np.asarray(cons_off[t0:t1 + 1], dtype=np.int64)
    return po - po[0], co - co[0], slice(int(po[0]), int(po[-1])), slice(int(co[0]), int(co[-1]))


def strong_plan(part_off: np.ndarray, world_size: int):
    """The split of ONE batch over `world_size` ranks (strong scaling): (bounds, counts, cap) with
    bounds[r]..bounds[r+1] the topics of rank r (la_plan_shards), counts[r] its partitions and cap the largest
    count -- ncclAllGather moves equal counts, so every rank's result buffers are `cap` long and the tail is padding."""
    bounds = shard_bounds(part_off, world_size)
    counts = [int(part_off[t1] - part_off[t0]) for t0, t1 in bounds]
    return bounds, counts, (max(counts) if counts else 0)


def strip_padding(gathered, counts: List[int], cap: int):
    """[world * cap] all-gathered array -> the global array (rank order = topic order: shards are contiguous)."""
    import torch
    if isinstance(gathered, np.ndarray):
        return np.concatenate([gathered[r * cap: r * cap + counts[r]] for r in range(len(counts))])
    return torch.cat([gathered[r * cap: r * cap + counts[r]] for r in range(len(counts))])


def wire_format_numpy(max_partition_id: int, n_members: int) -> Tuple[int, int]:
    """la_wire_format_for restated (elem_bytes, id_bits): the narrowest unsigned element that holds
    ((member rank + 1) << id_bits) | partition id for ids in [0, max_partition_id] and ranks in [-1, n_members);
    (8, 32) carries any int32 pair.  tests/test_sharding_gloo.py asserts that the library agrees."""
    if max_partition_id < 0 or max_partition_id > 0x7FFFFFFF or n_members < 0 or n_members > 0x7FFFFFFF:
        return 8, 32
    ib, rb = int(max_partition_id).bit_length(), int(n_members).bit_length()
    if ib + rb <= 16:
        return 2, ib
    if ib + rb <= 32:
        return 4, ib
    return 8, 32


_WIRE_DTYPE = {2: np.uint16, 4: np.uint32, 8: np.uint64}


def pack_results_numpy(pid: np.ndarray, rank: np.ndarray, elem_bytes: int, id_bits: int) -> np.ndarray:
    """The wire format on the host (what la_pack_results_on does on the device): raises if a pair does not fit."""
    p = np.asarray(pid, dtype=np.int32).view(np.uint32).astype(np.uint64)
    r1 = (np.asarray(rank, dtype=np.int64) + 1)
    if (r1 < 0).any():
        raise ValueError("member ranks must be >= -1")
    r1 = r1.astype(np.uint64)
    if elem_bytes != 8:
        if (p >> np.uint64(id_bits)).any() or (r1 >> np.uint64(8 * elem_bytes - id_bits)).any():
            raise ValueError("a partition id or member rank does not fit the wire format")
    return ((r1 << np.uint64(id_bits)) | p).astype(_WIRE_DTYPE[elem_bytes])


def unpack_results_numpy(packed: np.ndarray, elem_bytes: int, id_bits: int):
    w = np.asarray(packed).astype(np.uint64)
    mask = np.uint64(0xFFFFFFFF if id_bits >= 32 else (1 << id_bits) - 1)
    pid = (w & mask).astype(np.uint32).view(np.int32)
    rank = ((w >> np.uint64(id_bits)).astype(np.int64) - 1).astype(np.int32)
    return pid, rank


def gather_results_packed(local_pid, local_rank, counts: List[int], max_partition_id: int, n_members: int, group=None):
    """gather_results through the narrow wire format: ONE all_gather_into_tensor of `cap` elements of 2 / 4 / 8 bytes per
    rank instead of two int32 arrays (8 B per partition).  Host tensors (gloo; the CPU test): packed with the numpy
    restatement; device tensors go through la_pack_results_on / la_unpack_results_on in bench.py."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cap = max(counts) if counts else 0
    eb, ib = wire_format_numpy(max_partition_id, n_members)
    dt = {2: torch.int16, 4: torch.int32, 8: torch.int64}[eb]           # same width; gloo has no unsigned 16 / 32
    send = torch.zeros(cap, dtype=dt)
    packed = pack_results_numpy(local_pid.numpy(), local_rank.numpy(), eb, ib)
    send[: packed.size] = torch.from_numpy(packed.view({2: np.int16, 4: np.int32, 8: np.int64}[eb]))
    recv = torch.empty(world * cap, dtype=dt)
    dist.all_gather_into_tensor(recv, send, group=group)
    g = strip_padding(recv.numpy().view(_WIRE_DTYPE[eb]), counts, cap)
    pid, rank = unpack_results_numpy(g, eb, ib)
    return torch.from_numpy(pid), torch.from_numpy(rank), eb * cap


def gather_results(local_pid, local_rank, counts: List[int], group=None):
    """All-gathers the per-rank result arrays into the global arrays (topic order = rank
    order, because shards are contiguous).  ``counts[r]`` = partitions owned by rank r.
    Shards are padded to the largest so ONE all_gather_into_tensor per array suffices
    (ncclAllGather needs equal counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cap = max(counts) if counts else 0
    dev = local_pid.device

    def one(x):
        send = torch.zeros(cap, dtype=x.dtype, device=dev)
        send[: x.numel()] = x
        recv = torch.empty(world * cap, dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        return strip_padding(recv, counts, cap)

    return one(local_pid), one(local_rank)
