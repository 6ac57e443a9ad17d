"""ctypes binding of the C ABI in include/lagassign.h (liblagassign.so).

This is the same boundary a JNI shim binds.  There is no fallback: if
the shared object is missing or no gfx950 device is usable, calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LA_LIB_PATH") or os.path.join(_HERE, "liblagassign.so")   # LA_LIB_PATH: A/B runs of two builds

LA_OK = 0
LA_EINVAL, LA_ENOMEM, LA_EHIP, LA_ENODEV, LA_ESHAPE = -1, -2, -3, -4, -5
LA_RESET_LATEST, LA_RESET_EARLIEST = 0, 1
LA_ALGO_AUTO, LA_ALGO_ROUNDS, LA_ALGO_ARGMIN, LA_ALGO_ROUNDS_WIDE = 0, 1, 2, 3
LA_FLAG_INDEX64, LA_FLAG_DEFER_WIDE, LA_FLAG_RAGGED, LA_FLAG_SHAPE_CLASSES = 1, 2, 4, 8
LA_FLAG_PROFILE, LA_FLAG_NO_SAMPLE_SORT, LA_FLAG_SAMPLE_TIGHT = 16, 32, 64
LA_FLAG_SORT_MULTIKERNEL = 128
LA_FLAG_NO_RUN_MERGE = 256
LA_FLAG_NO_MOVED_SORT = 2048
LA_FLAG_SERIAL_LARGE = 512
LA_FLAG_BOUNDS = 1024
LA_FLAG_WIRE_OUT = 4096
LA_FEATURE_ATOMIC_RANK = 1
LA_PIPELINE_ONE_COPY, LA_PIPELINE_LANES, LA_PIPELINE_STREAMS, LA_PIPELINE_ZERO_COPY, LA_PIPELINE_MAPPED = 0, 1, 2, 3, 4
LA_CREATE_LANES_MASK, LA_CREATE_SPLIT_ALWAYS = 0xF, 0x10
LA_HINT_BOUNDS = 1

EXPORTED_SYMBOLS = (
    "la_create", "la_destroy", "la_last_error", "la_version", "la_compute_lag",
    "la_assign_batch", "la_assign_batch_lags", "la_assign_batch_device", "la_sync", "la_stream",
    "la_group_by_member", "la_group_by_member_device", "la_group_last_by_member",
    "la_device_count", "la_create_multi", "la_shard_count", "la_shard_device", "la_plan_shards",
    "la_last_shard_bounds", "la_host_alloc", "la_host_free", "la_last_phase_times", "la_device_features",
    "la_shard_stream", "la_assign_batch_device_on", "la_sync_on", "la_group_by_member_device_on", "la_last_pipeline",
    "la_allgather_results", "la_assign_batch_grouped",
    "la_wire_format_for", "la_pack_results_on", "la_unpack_results_on", "la_allgather_packed",
    "la_assign_batch_sparse", "la_assign_batch_grouped_sparse",
    "la_hint_next_call", "la_last_launches", "la_last_phase_times_sized", "la_wake",
)

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p      # what the array parameters are declared as: an address (int), None, or any ctypes pointer
_byte = ctypes.c_char


class LagAssignError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("liblagassign error %d: %s" % (code, message))
        self.code = code


class DeviceBatch(ctypes.Structure):
    """struct la_device_batch"""
    _fields_ = [
        ("n_topics", ctypes.c_int32), ("reset_mode", ctypes.c_int32),
        ("algo", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("n_partitions", ctypes.c_int64), ("n_consumers", ctypes.c_int64),
        ("max_partitions_per_topic", ctypes.c_int64), ("max_consumers_per_topic", ctypes.c_int64),
        ("d_part_off", ctypes.c_void_p), ("d_partition_id", ctypes.c_void_p),
        ("d_begin_off", ctypes.c_void_p), ("d_end_off", ctypes.c_void_p),
        ("d_committed_off", ctypes.c_void_p), ("d_lag", ctypes.c_void_p),
        ("d_cons_off", ctypes.c_void_p), ("d_cons_rank", ctypes.c_void_p),
        ("d_out_partition", ctypes.c_void_p), ("d_out_member_rank", ctypes.c_void_p),
        ("d_out_total_lag", ctypes.c_void_p),
        ("h_part_off", _i64p), ("h_cons_off", _i64p),
        ("max_lag_hint", ctypes.c_int64), ("max_partition_id_hint", ctypes.c_int64),
        ("d_out_wire", ctypes.c_void_p), ("wire_elem_bytes", ctypes.c_int32), ("wire_id_bits", ctypes.c_int32),
    ]


class CallHints(ctypes.Structure):
    """struct la_call_hints"""
    _fields_ = [("struct_size", ctypes.c_int32), ("flags", ctypes.c_int32), ("max_lag", ctypes.c_int64),
                ("max_partition_id", ctypes.c_int64)]


class WireFormat(ctypes.Structure):
    """struct la_wire_format: one element = ((member rank + 1) << id_bits) | partition id, in elem_bytes bytes"""
    _fields_ = [("elem_bytes", ctypes.c_int32), ("id_bits", ctypes.c_int32)]

    @property
    def dtype(self):
        return {2: np.uint16, 4: np.uint32, 8: np.uint64}[int(self.elem_bytes)]


class PhaseTimes(ctypes.Structure):
    """struct la_phase_times"""
    _fields_ = [("n_partitions", ctypes.c_int64), ("id_passes", ctypes.c_int32), ("key_passes", ctypes.c_int32),
                ("keys_ms", ctypes.c_float), ("sort_ms", ctypes.c_float), ("greedy_ms", ctypes.c_float),
                ("keys_first", ctypes.c_int32), ("redone", ctypes.c_int32)]


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Loads liblagassign.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m kafka_lag_based_assignor_amd.build` "
            "(needs hipcc). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  Importing torch
    # first makes liblagassign's NEEDED libamdhip64.so.7 resolve to that already-loaded copy,
    # so torch tensors, torch streams and our kernels share one runtime.  (Two runtimes in one
    # process fail at device discovery.)  Without torch -- e.g. under a JVM -- the system
    # ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.la_version.restype = ctypes.c_int
    L.la_create.restype = ctypes.c_int
    L.la_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_uint]
    L.la_create_multi.restype = ctypes.c_int
    L.la_create_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.c_uint]
    L.la_device_count.restype = ctypes.c_int
    L.la_device_count.argtypes = []
    L.la_shard_count.restype = ctypes.c_int
    L.la_shard_count.argtypes = [ctypes.c_void_p]
    L.la_shard_device.restype = ctypes.c_int
    L.la_shard_device.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.la_device_features.restype = ctypes.c_int
    L.la_device_features.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.la_plan_shards.restype = ctypes.c_int
    L.la_plan_shards.argtypes = [ctypes.c_int32, _vp, ctypes.c_int32, _vp]
    L.la_last_shard_bounds.restype = ctypes.c_int
    L.la_last_shard_bounds.argtypes = [ctypes.c_void_p, _vp, ctypes.c_int32]
    L.la_host_alloc.restype = ctypes.c_void_p
    L.la_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.la_host_free.restype = None
    L.la_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.la_destroy.restype = None
    L.la_destroy.argtypes = [ctypes.c_void_p]
    L.la_last_error.restype = ctypes.c_char_p
    L.la_last_error.argtypes = [ctypes.c_void_p]
    L.la_compute_lag.restype = ctypes.c_int
    L.la_compute_lag.argtypes = [ctypes.c_void_p, ctypes.c_int64, _vp, _vp, _vp, ctypes.c_int32, _vp]
    L.la_assign_batch.restype = ctypes.c_int
    L.la_assign_batch.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp,
                                  ctypes.c_int32, _vp, _vp, _vp, _vp, _vp]
    L.la_assign_batch_lags.restype = ctypes.c_int
    L.la_assign_batch_grouped.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp,
                                          ctypes.c_int32, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp]
    L.la_assign_batch_grouped.restype = ctypes.c_int
    L.la_assign_batch_lags.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp]
    L.la_assign_batch_device.restype = ctypes.c_int
    L.la_assign_batch_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(DeviceBatch), ctypes.c_void_p]
    L.la_sync.restype = ctypes.c_int
    L.la_sync.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.la_last_phase_times.restype = ctypes.c_int
    L.la_last_phase_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(PhaseTimes)]
    L.la_stream.restype = ctypes.c_void_p
    L.la_stream.argtypes = [ctypes.c_void_p]
    L.la_group_by_member.restype = ctypes.c_int
    L.la_group_by_member.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32,
                                     _vp, _vp, _vp]
    L.la_group_last_by_member.restype = ctypes.c_int
    L.la_group_last_by_member.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp]
    L.la_group_by_member_device.restype = ctypes.c_int
    L.la_group_by_member_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.la_allgather_results.restype = ctypes.c_int
    L.la_allgather_results.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p)]
    L.la_last_pipeline.restype = ctypes.c_int
    L.la_last_pipeline.argtypes = [ctypes.c_void_p]
    L.la_wire_format_for.restype = ctypes.c_int
    L.la_wire_format_for.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(WireFormat)]
    L.la_pack_results_on.restype = ctypes.c_int
    L.la_pack_results_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(WireFormat), ctypes.c_void_p, ctypes.c_void_p]
    L.la_unpack_results_on.restype = ctypes.c_int
    L.la_unpack_results_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.POINTER(WireFormat), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.la_allgather_packed.restype = ctypes.c_int
    L.la_allgather_packed.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_void_p)]
    L.la_assign_batch_sparse.restype = ctypes.c_int
    L.la_assign_batch_sparse.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, _vp, ctypes.c_int32,
                                         ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.la_assign_batch_grouped_sparse.restype = ctypes.c_int
    L.la_assign_batch_grouped_sparse.argtypes = [ctypes.c_void_p, ctypes.c_int32, _vp, _vp, _vp, _vp, ctypes.c_int32,
                                                 ctypes.c_int64, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp,
                                                 _vp, _vp]
    L.la_shard_stream.restype = ctypes.c_void_p
    L.la_shard_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.la_assign_batch_device_on.restype = ctypes.c_int
    L.la_assign_batch_device_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.la_sync_on.restype = ctypes.c_int
    L.la_sync_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.la_group_by_member_device_on.restype = ctypes.c_int
    L.la_group_by_member_device_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.la_hint_next_call.restype = ctypes.c_int
    L.la_hint_next_call.argtypes = [ctypes.c_void_p, ctypes.POINTER(CallHints)]
    L.la_last_launches.restype = ctypes.c_int64
    L.la_last_launches.argtypes = [ctypes.c_void_p]
    L.la_wake.restype = ctypes.c_int
    L.la_wake.argtypes = [ctypes.c_void_p]
    L.la_last_phase_times_sized.restype = ctypes.c_int
    L.la_last_phase_times_sized.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    _lib = L
    return L


def _a64(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.int64)


def _a32(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.int32)


def _addr(a: Optional[np.ndarray]):
    """Address of a contiguous array's first element for a c_void_p parameter.  A small rebalance is a dozen pointers per call
    and `a.ctypes.data_as(...)` costs 1-2.5 us apiece (it builds a helper object and a typed pointer); the buffer protocol gives
    the same address in 0.3 us.  The CALLER keeps `a` alive across the call (the wrappers below hold every array in a local).
    Read-only and empty arrays do not export a writable buffer: they take the slow way."""
    if a is None:
        return None
    try:
        return ctypes.addressof(_byte.from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data


_p64 = _addr       # (dtype and contiguity are the business of _a64 / _a32 before)
_p32 = _addr


def _check_out3(out, n: int, k: int):
    """Caller-owned result buffers (out_partition int32[N], out_member_rank int32[N], totals int64[K] or None): the library writes
    N / K contiguous elements at each address, so anything else -- a strided view above all -- is refused here."""
    out_p, out_m, out_t = out
    ok = (isinstance(out_p, np.ndarray) and isinstance(out_m, np.ndarray) and out_p.dtype == np.int32 and out_m.dtype == np.int32 and
          out_p.size == n and out_m.size == n and out_p.flags.c_contiguous and out_m.flags.c_contiguous and
          (out_t is None or (isinstance(out_t, np.ndarray) and out_t.dtype == np.int64 and out_t.size == k and out_t.flags.c_contiguous)))
    if not ok:
        raise ValueError("out buffers must be contiguous int32[N], int32[N], int64[K]")
    return out_p, out_m, out_t


def plan_shards(part_off, n_shards: int) -> np.ndarray:
    """la_plan_shards: the library's own planner (pure host code, no device needed).  Returns int32
    bounds[n_shards + 1]: shard r owns topics [bounds[r], bounds[r+1])."""
    part_off = _a64(part_off)
    bounds = np.zeros(n_shards + 1, dtype=np.int32)
    rc = load().la_plan_shards(part_off.size - 1, _p64(part_off), n_shards, _p32(bounds))
    if rc != LA_OK:
        raise LagAssignError(rc, "la_plan_shards: bad arguments")
    return bounds


def wire_format_for(max_partition_id: int, n_members: int) -> WireFormat:
    """la_wire_format_for (pure host code): the narrowest wire element for ids in [0, max_partition_id] (negative: any
    int32) and member ranks in [-1, n_members)."""
    f = WireFormat()
    rc = load().la_wire_format_for(int(max_partition_id), int(n_members), ctypes.byref(f))
    if rc != LA_OK:
        raise LagAssignError(rc, "la_wire_format_for")
    return f


def sparse_begin(begin, committed):
    """(none_index int64[m], none_begin int64[m]) of a dense `begin` array: the positions without a committed offset, ascending
    -- what a marshaller that knows `md == null` hands to la_assign_batch_sparse instead of the dense array."""
    idx = np.flatnonzero(np.asarray(committed) < 0).astype(np.int64)
    return idx, np.ascontiguousarray(np.asarray(begin, dtype=np.int64)[idx])


def offset_bounds(begin, end, committed, partition_id, lag=None):
    """What a marshaller that walks every partition knows for free (Main.java:344-356): (largest end offset, largest partition
    id) -- a lag never exceeds its end offset when no offset of the batch is negative -- or None when an offset or an id IS
    negative (then there is nothing to promise).  With `lag` (the precomputed-lags seam): (largest lag, largest id)."""
    pid = np.asarray(partition_id)
    if pid.size == 0 or int(pid.min()) < 0:
        return None
    if lag is not None:
        lag = np.asarray(lag)
        return None if int(lag.min()) < 0 else (int(lag.max()), int(pid.max()))
    end = np.asarray(end)
    if int(end.min()) < 0:
        return None
    if begin is not None:
        b = np.asarray(begin)
        if b.size and int(b.min()) < 0:
            return None
    return int(end.max()), int(pid.max())


def device_count() -> int:
    n = load().la_device_count()
    if n < 0:
        raise LagAssignError(n, load().la_last_error(None).decode())
    return n


class Context:
    """RAII wrapper of la_ctx (one per assignor instance; not thread-safe).

    ``device_id`` may be an int (one device) or a sequence of device ids (la_create_multi: one shard per
    entry; an id may repeat -- several logical shards on one GPU); ``devices="all"`` takes every device."""

    def __init__(self, device_id=0, flags: int = 0):
        self._lib = load()
        h = ctypes.c_void_p()
        if isinstance(device_id, str):
            if device_id != "all":
                raise ValueError("device_id must be an int, a sequence of ints or 'all'")
            rc = self._lib.la_create_multi(ctypes.byref(h), 0, None, flags)
        elif isinstance(device_id, (list, tuple)):
            ids = (ctypes.c_int * len(device_id))(*device_id)
            rc = self._lib.la_create_multi(ctypes.byref(h), len(device_id), ids, flags)
        else:
            rc = self._lib.la_create(ctypes.byref(h), int(device_id), flags)
        if rc != LA_OK:
            raise LagAssignError(rc, self._lib.la_last_error(None).decode())
        self._h = h

    @property
    def shard_count(self) -> int:
        return int(self._lib.la_shard_count(self._h))

    def shard_device(self, i: int) -> int:
        return int(self._lib.la_shard_device(self._h, i))

    def device_features(self, i: int = 0) -> int:
        """LA_FEATURE_* bits of shard i's device."""
        return int(self._lib.la_device_features(self._h, i))

    def last_shard_bounds(self) -> np.ndarray:
        """Topic ranges the last host-buffer assign call gave to its shards (int32 [S + 1])."""
        buf = np.zeros(65, dtype=np.int32)
        s = int(self._lib.la_last_shard_bounds(self._h, _p32(buf), buf.size))
        return buf[: s + 1].copy()

    def host_alloc(self, shape, dtype) -> np.ndarray:
        """A numpy array over pinned host memory from la_host_alloc (freed when the array is collected)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        p = self._lib.la_host_alloc(self._h, max(1, n * dtype.itemsize))
        if not p:
            raise LagAssignError(LA_ENOMEM, self._lib.la_last_error(self._h).decode())
        lib = self._lib

        class _Owner:
            def __del__(self_inner):
                lib.la_host_free(None, ctypes.c_void_p(p))

        buf = (ctypes.c_char * max(1, n * dtype.itemsize)).from_address(p)
        buf._owner = _Owner()
        return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.la_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != LA_OK:
            raise LagAssignError(rc, self._lib.la_last_error(self._h).decode())

    def hint_next_call(self, bounds) -> None:
        """la_hint_next_call: `bounds` = (max_lag, max_partition_id) the caller guarantees for its NEXT host-buffer assign call
        (offset_bounds() computes them the way a marshaller would), or None = no hint (clears a pending one)."""
        if bounds is None:
            self._check(self._lib.la_hint_next_call(self._h, None))
            return
        h = CallHints(ctypes.sizeof(CallHints), LA_HINT_BOUNDS, int(bounds[0]), int(bounds[1]))
        self._check(self._lib.la_hint_next_call(self._h, ctypes.byref(h)))

    def last_launches(self) -> int:
        """Kernel launches the last call on this context enqueued (la_last_launches)."""
        return int(self._lib.la_last_launches(self._h))

    def wake(self) -> None:
        """la_wake: one one-partition rebalance through the real small-call path (waited for) + an empty kernel on every other
        stream -- for the moment a host ENTERS assign(), milliseconds before it has offsets to hand over."""
        self._check(self._lib.la_wake(self._h))

    # -- host-buffer entry points ------------------------------------------------
    def compute_lag(self, begin, end, committed, reset_mode: int) -> np.ndarray:
        end, committed = _a64(end), _a64(committed)
        begin = None if begin is None else _a64(begin)
        out = np.empty_like(end)
        self._check(self._lib.la_compute_lag(self._h, end.size, _p64(begin), _p64(end), _p64(committed),
                                             reset_mode, _p64(out)))
        return out

    def assign_batch(self, part_off, partition_id, begin, end, committed, reset_mode: int,
                     cons_off, cons_rank, want_totals: bool = True, out=None, keep_on_device: bool = False
                     ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        """`out` = (out_partition int32[N], out_member_rank int32[N], out_total_lag int64[K] or None): caller-owned
        result buffers to reuse across calls (fresh numpy arrays are page-faulted in by the D2H copy, which
        doubles the time of a 25.6 M-partition call)."""
        part_off, cons_off = _a64(part_off), _a64(cons_off)
        partition_id, cons_rank = _a32(partition_id), _a32(cons_rank)
        end, committed = _a64(end), _a64(committed)
        begin = None if begin is None else _a64(begin)
        if keep_on_device:
            # results stay on the device for group_last_by_member(); only the totals come back
            out_p = out_m = None
            out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        elif out is not None:
            out_p, out_m, out_t = _check_out3(out, partition_id.size, cons_rank.size)
        else:
            out_p = np.empty(partition_id.size, dtype=np.int32)
            out_m = np.empty(partition_id.size, dtype=np.int32)
            out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        self._check(self._lib.la_assign_batch(self._h, part_off.size - 1, _p64(part_off), _p32(partition_id),
                                              _p64(begin), _p64(end), _p64(committed), reset_mode,
                                              _p64(cons_off), _p32(cons_rank), _p32(out_p), _p32(out_m),
                                              _p64(out_t)))
        return out_p, out_m, out_t

    def assign_batch_sparse(self, part_off, partition_id, end, committed, reset_mode: int, none_index, none_begin,
                            cons_off, cons_rank, want_totals: bool = True, out=None, keep_on_device: bool = False):
        """la_assign_batch_sparse: `begin` only for the partitions listed in none_index (ascending positions)."""
        part_off, cons_off = _a64(part_off), _a64(cons_off)
        partition_id, cons_rank = _a32(partition_id), _a32(cons_rank)
        end, committed = _a64(end), _a64(committed)
        none_index = None if none_index is None else _a64(none_index)
        none_begin = None if none_begin is None else _a64(none_begin)
        n_none = 0 if none_index is None else none_index.size
        if keep_on_device:
            out_p = out_m = None
            out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        elif out is not None:
            out_p, out_m, out_t = _check_out3(out, partition_id.size, cons_rank.size)
        else:
            out_p = np.empty(partition_id.size, dtype=np.int32)
            out_m = np.empty(partition_id.size, dtype=np.int32)
            out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        self._check(self._lib.la_assign_batch_sparse(self._h, part_off.size - 1, _p64(part_off), _p32(partition_id),
                                                     _p64(end), _p64(committed), reset_mode, n_none, _p64(none_index),
                                                     _p64(none_begin), _p64(cons_off), _p32(cons_rank), _p32(out_p),
                                                     _p32(out_m), _p64(out_t)))
        return out_p, out_m, out_t

    def assign_batch_grouped_sparse(self, part_off, partition_id, end, committed, reset_mode: int, none_index, none_begin,
                                    cons_off, cons_rank, n_members: int, want_totals: bool = True, want_topic: bool = True):
        """la_assign_batch_grouped_sparse -> (member_off, grouped_topic or None, grouped_partition, totals or None)."""
        part_off, cons_off = _a64(part_off), _a64(cons_off)
        partition_id, cons_rank = _a32(partition_id), _a32(cons_rank)
        end, committed = _a64(end), _a64(committed)
        none_index = None if none_index is None else _a64(none_index)
        none_begin = None if none_begin is None else _a64(none_begin)
        n_none = 0 if none_index is None else none_index.size
        off = np.zeros(n_members + 1, dtype=np.int64)
        g_t = np.empty(partition_id.size, dtype=np.int32) if want_topic else None
        g_p = np.empty(partition_id.size, dtype=np.int32)
        out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        self._check(self._lib.la_assign_batch_grouped_sparse(self._h, part_off.size - 1, _p64(part_off), _p32(partition_id),
                                                             _p64(end), _p64(committed), reset_mode, n_none,
                                                             _p64(none_index), _p64(none_begin), _p64(cons_off),
                                                             _p32(cons_rank), n_members, _p64(off), _p32(g_t), _p32(g_p),
                                                             _p64(out_t)))
        return off, g_t, g_p, out_t

    def assign_batch_lags(self, part_off, partition_id, lag, cons_off, cons_rank, want_totals: bool = True,
                          keep_on_device: bool = False, out=None
                          ) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], Optional[np.ndarray]]:
        """keep_on_device: the results stay on the device for group_last_by_member() (None, None, totals come back).
        out: caller-owned (int32[N], int32[N], int64[K] or None) result buffers, as for assign_batch."""
        part_off, cons_off = _a64(part_off), _a64(cons_off)
        partition_id, cons_rank, lag = _a32(partition_id), _a32(cons_rank), _a64(lag)
        if out is not None and not keep_on_device:
            out_p, out_m, out_t = _check_out3(out, partition_id.size, cons_rank.size)
        else:
            out_p = None if keep_on_device else np.empty(partition_id.size, dtype=np.int32)
            out_m = None if keep_on_device else np.empty(partition_id.size, dtype=np.int32)
            out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        self._check(self._lib.la_assign_batch_lags(self._h, part_off.size - 1, _p64(part_off),
                                                   _p32(partition_id), _p64(lag), _p64(cons_off),
                                                   _p32(cons_rank), _p32(out_p), _p32(out_m), _p64(out_t)))
        return out_p, out_m, out_t

    def assign_batch_grouped(self, part_off, partition_id, begin, end, committed, reset_mode: int, cons_off, cons_rank,
                             n_members: int, want_totals: bool = True, want_topic: bool = True
                             ) -> Tuple[np.ndarray, Optional[np.ndarray], np.ndarray, Optional[np.ndarray]]:
        """la_assign_batch_grouped: assign and every member's list in ONE call (for a small batch: one upload, one
        download).  Returns (member_off [M+1], grouped_topic [N] or None, grouped_partition [N], totals [K] or None)."""
        part_off, cons_off = _a64(part_off), _a64(cons_off)
        partition_id, cons_rank = _a32(partition_id), _a32(cons_rank)
        end, committed = _a64(end), _a64(committed)
        begin = None if begin is None else _a64(begin)
        off = np.zeros(n_members + 1, dtype=np.int64)
        g_t = np.empty(partition_id.size, dtype=np.int32) if want_topic else None
        g_p = np.empty(partition_id.size, dtype=np.int32)
        out_t = np.zeros(cons_rank.size, dtype=np.int64) if want_totals else None
        self._check(self._lib.la_assign_batch_grouped(self._h, part_off.size - 1, _p64(part_off), _p32(partition_id),
                                                      _p64(begin), _p64(end), _p64(committed), reset_mode,
                                                      _p64(cons_off), _p32(cons_rank), n_members, _p64(off), _p32(g_t),
                                                      _p32(g_p), _p64(out_t)))
        return off, g_t, g_p, out_t

    def group_by_member(self, part_off, out_partition, out_member_rank, n_members: int
                        ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(member_off [M+1], grouped_topic [N], grouped_partition [N]): every member's list in the
        reference's order; entries before member_off[0] belong to topics without consumers."""
        part_off = _a64(part_off)
        out_partition, out_member_rank = _a32(out_partition), _a32(out_member_rank)
        off = np.zeros(n_members + 1, dtype=np.int64)
        g_t = np.empty(out_partition.size, dtype=np.int32)
        g_p = np.empty(out_partition.size, dtype=np.int32)
        self._check(self._lib.la_group_by_member(self._h, part_off.size - 1, _p64(part_off), _p32(out_partition),
                                                 _p32(out_member_rank), n_members, _p64(off), _p32(g_t), _p32(g_p)))
        return off, g_t, g_p

    def group_last_by_member(self, n_partitions: int, n_members: int, out=None
                             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """group_by_member on the results the last assign_batch / assign_batch_lags call left on the device
        (give that call keep_on_device=True to skip the download of the ungrouped arrays).
        `out` = (member_off int64[M+1], grouped_topic int32[N], grouped_partition int32[N]): caller-owned arrays to
        reuse across calls (pinned ones from host_alloc, for instance), as for assign_batch."""
        if out is not None:
            off, g_t, g_p = out
            if (off.dtype != np.int64 or off.size != n_members + 1 or g_t.dtype != np.int32 or g_p.dtype != np.int32 or
                    g_t.size != n_partitions or g_p.size != n_partitions or
                    not (off.flags.c_contiguous and g_t.flags.c_contiguous and g_p.flags.c_contiguous)):
                raise ValueError("out buffers must be contiguous int64[M+1], int32[N], int32[N]")
        else:
            off = np.zeros(n_members + 1, dtype=np.int64)
            g_t = np.empty(n_partitions, dtype=np.int32)
            g_p = np.empty(n_partitions, dtype=np.int32)
        self._check(self._lib.la_group_last_by_member(self._h, n_members, _p64(off), _p32(g_t), _p32(g_p)))
        return off, g_t, g_p

    def group_by_member_device(self, n_topics: int, n_partitions: int, d_part_off: int, d_out_partition: int,
                               d_out_member_rank: int, n_members: int, d_member_off: int, d_grouped_topic: int,
                               d_grouped_partition: int, stream: int = 0, shard: int = 0) -> None:
        self._check(self._lib.la_group_by_member_device_on(self._h, shard, n_topics, n_partitions, d_part_off,
                                                           d_out_partition, d_out_member_rank, n_members,
                                                           d_member_off, d_grouped_topic, d_grouped_partition,
                                                           ctypes.c_void_p(stream)))

    # -- device-resident entry point ------------------------------------------------
    def assign_batch_device(self, batch: DeviceBatch, stream: int = 0, shard: int = 0) -> None:
        """Enqueue on `stream` (a hipStream_t handle as an int; 0 = HIP's default stream) of shard `shard`'s device."""
        self._check(self._lib.la_assign_batch_device_on(self._h, shard, ctypes.byref(batch), ctypes.c_void_p(stream)))

    def sync(self, stream: int = 0, shard: int = 0) -> None:
        self._check(self._lib.la_sync_on(self._h, shard, ctypes.c_void_p(stream)))

    def allgather_results(self, count: int, d_send, d_recv) -> None:
        """la_allgather_results: d_send / d_recv are lists of device pointers (ints), one per shard."""
        n = self.shard_count
        send = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in d_send])
        recv = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in d_recv])
        self._check(self._lib.la_allgather_results(self._h, count, send, recv))

    def pack_results(self, n: int, d_out_partition: int, d_out_member_rank: int, fmt: WireFormat, d_packed: int,
                     stream: int = 0, shard: int = 0) -> None:
        """la_pack_results_on: n (partition id, member rank) pairs -> n wire elements (device pointers as ints)."""
        self._check(self._lib.la_pack_results_on(self._h, shard, n, ctypes.c_void_p(d_out_partition),
                                                 ctypes.c_void_p(d_out_member_rank), ctypes.byref(fmt),
                                                 ctypes.c_void_p(d_packed), ctypes.c_void_p(stream)))

    def unpack_results(self, n: int, d_packed: int, fmt: WireFormat, d_out_partition: int, d_out_member_rank: int,
                       stream: int = 0, shard: int = 0) -> None:
        """la_unpack_results_on: the reverse."""
        self._check(self._lib.la_unpack_results_on(self._h, shard, n, ctypes.c_void_p(d_packed), ctypes.byref(fmt),
                                                   ctypes.c_void_p(d_out_partition), ctypes.c_void_p(d_out_member_rank),
                                                   ctypes.c_void_p(stream)))

    def allgather_packed(self, count: int, elem_bytes: int, d_send, d_recv) -> None:
        """la_allgather_packed: d_send / d_recv are lists of device pointers (ints), one per shard."""
        n = self.shard_count
        send = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in d_send])
        recv = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in d_recv])
        self._check(self._lib.la_allgather_packed(self._h, count, elem_bytes, send, recv))

    def last_pipeline(self) -> int:
        """LA_PIPELINE_* of the last host-buffer assign call."""
        return int(self._lib.la_last_pipeline(self._h))

    def shard_stream(self, shard: int) -> int:
        return int(self._lib.la_shard_stream(self._h, shard) or 0)

    def last_phase_times(self) -> PhaseTimes:
        """Phase times of the large-path topic of the last assign_batch_device call with LA_FLAG_PROFILE."""
        t = PhaseTimes()
        self._check(self._lib.la_last_phase_times(self._h, ctypes.byref(t)))
        return t

    @property
    def stream(self) -> int:
        return int(self._lib.la_stream(self._h) or 0)
