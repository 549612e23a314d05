"""Row-range sharding of the select / where / by path over the GPUs of one node (one process per GPU).

GPU ``g`` owns rows ``[row0_g, row0_g + n_g)`` of every column.  Nothing is exchanged on the data path; ONE small
exchange merges the per-GPU partial states:

* scalar aggregates -> ``all_gather`` of ``(nagg + 1) x 64`` bytes of ``rfx_partial_t`` and a host-side fold in rank order
  (``rfx_partial_merge``; integer results exact, f64 sums added in a fixed rank order);
* dense group-by     -> ``all_reduce`` of the tables: ``first`` with MIN (global row ids), accumulators with SUM / MIN / MAX
  by aggregate kind, the count arrays with SUM.  The scope ``[kmin, kmax]`` is agreed first with an all_reduce of 3 numbers.
  Ranking by first row and emit then run identically on every rank (replicated result);
* hashed group-by    -> ``all_gather`` of the whole table set and re-insertion of the other ranks' occupied slots
  (``rfx_hip_hash_tables_merge``);
* ``where`` ids      -> per-rank ascending global ids are already globally ordered by rank: all_gather of counts + padded ids.

``torch.distributed`` with backend ``nccl`` IS RCCL on ROCm (xGMI inside a node); the same code runs on ``gloo`` with CPU
tensors, which is how the merge logic is tested without GPUs (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib as L


def _gloo(group=None) -> bool:
    return dist.is_initialized() and dist.get_backend(group) == "gloo"


def _all_reduce(t: torch.Tensor, op, group=None) -> None:
    """all_reduce in place; under gloo a device tensor takes a host round trip (test mode: ranks sharing one GPU)."""
    if _gloo(group) and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather(t: torch.Tensor, group=None):
    world = dist.get_world_size(group)
    if _gloo(group) and t.is_cuda:
        h = t.cpu()
        bufs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(bufs, h, group=group)
        return [b.to(t.device) for b in bufs]
    if not _gloo(group):
        # one contiguous receive buffer: a single collective and, for small payloads read back by the host, a single copy
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return list(out.unbind(0))
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t, group=group)
    return bufs


class RowShard:
    """Where this rank's rows sit in the global table."""

    def __init__(self, local_rows: int, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_rows = int(local_rows)
        if self.world > 1:
            counts = [None] * self.world
            dist.all_gather_object(counts, self.local_rows, group=group)
        else:
            counts = [self.local_rows]
        self.counts = [int(c) for c in counts]
        self.row0 = sum(self.counts[: self.rank])
        self.total_rows = sum(self.counts)


# ---------------------------------------------------------------------------------------------- scalar aggregates
def merge_scalar_partials(partials: torch.Tensor, kinds: Sequence[int], col_types: Sequence[int], group=None):
    """partials: uint8 tensor of (nagg + 1) * 64 bytes (device or host).  Returns ([rfx_value ...] as python values, selected)."""
    lib = L.load_library()
    nagg = len(kinds)
    nbytes = (nagg + 1) * 64
    assert partials.dtype == torch.uint8 and partials.numel() == nbytes
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        bufs = _all_gather(partials, group)
        base = bufs[0]._base if bufs[0]._base is not None else None
        if base is not None and base.numel() == world * nbytes:  # rows of one receive buffer: one device-to-host copy
            whole = base.cpu().numpy().tobytes()
            host = [whole[r * nbytes:(r + 1) * nbytes] for r in range(world)]
        else:
            host = [b.cpu().numpy().tobytes() for b in bufs]
    else:
        host = [partials.cpu().numpy().tobytes()]
    acc = (L.Partial * (nagg + 1)).from_buffer_copy(host[0])
    for other in host[1:]:
        o = (L.Partial * (nagg + 1)).from_buffer_copy(other)
        for a in range(nagg):
            lib.rfx_partial_merge(kinds[a], col_types[a], C.byref(acc[a]), C.byref(o[a]))
        lib.rfx_partial_merge(L.RFX_AGG_COUNT, L.RFX_I64, C.byref(acc[nagg]), C.byref(o[nagg]))
    vals = []
    for a in range(nagg):
        v = L.Value()
        L.check(lib.rfx_agg_finalize(kinds[a], col_types[a], C.byref(acc[a]), C.byref(v)), "agg_finalize")
        if v.type == L.RFX_F64:
            vals.append(float("nan") if v.is_null else float(v.f))
        else:
            vals.append(None if v.is_null else int(v.i))
    return vals, int(acc[nagg].cnt)


# ---------------------------------------------------------------------------------------------- dense group tables
def _reduce_op(kind: int, f64: bool, what: str):
    """(torch dtype to view the 8-byte cells as, ReduceOp) for one table array."""
    if what == "first":
        return torch.int64, dist.ReduceOp.MIN
    if what == "cnt":
        return torch.int64, dist.ReduceOp.SUM
    if kind == L.RFX_AGG_MIN:
        return torch.int64, dist.ReduceOp.MIN  # order-preserving i64 image for f64 columns too
    if kind == L.RFX_AGG_MAX:
        return torch.int64, dist.ReduceOp.MAX
    if kind == L.RFX_AGG_AVG or (kind == L.RFX_AGG_SUM and f64):
        return torch.float64, dist.ReduceOp.SUM
    return torch.int64, dist.ReduceOp.SUM  # SUM(i64) wraps, COUNT


def allreduce_tables(store: torch.Tensor, layout, kinds: Sequence[int], f64s: Sequence[bool], group=None) -> None:
    """In-place all-reduce of a dense table set ``store[n_arrays, range]`` (int64 view of 8-byte cells)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for row, (what, a) in enumerate(layout):
        kind = kinds[a] if a is not None else -1
        f64 = f64s[a] if a is not None else False
        dt, op = _reduce_op(kind, f64, what)
        _all_reduce(store[row].view(dt), op, group)


def allreduce_scope(kmin: int, kmax: int, seen: int, device, group=None) -> Tuple[int, int, int]:
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return kmin, kmax, seen
    # ranks that saw no row contribute neutral elements
    big = 2**63 - 1
    t = torch.tensor([kmin if seen else big, -(kmax if seen else -big), -seen], dtype=torch.int64, device=device)
    mm = t[:2].clone()
    _all_reduce(mm, dist.ReduceOp.MIN, group)
    t[:2] = mm
    s = t[2:].clone()
    _all_reduce(s, dist.ReduceOp.SUM, group)
    mn, negmx = int(t[0]), int(t[1])
    return mn, -negmx, -int(s[0])


# ---------------------------------------------------------------------------------------------- where ids
def gather_ids(local_ids: torch.Tensor, group=None, native=None) -> torch.Tensor:
    """Concatenate per-rank ascending GLOBAL ids (already offset by row0) into the global ascending id vector."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_ids
    if native:
        eng, lib = native
        world = dist.get_world_size(group)
        cnt = torch.tensor([local_ids.numel()], dtype=torch.int64, device=local_ids.device)
        cnts = torch.empty(world, dtype=torch.int64, device=local_ids.device)
        L.check(lib.rfx_dist_allgather(eng._ctx, cnt.data_ptr(), 8, cnts.data_ptr()), "dist_allgather")
        eng.sync()
        sizes = [int(x) for x in cnts.cpu()]
        m = max(sizes) if sizes else 0
        pad = torch.zeros(max(m, 1), dtype=torch.int64, device=local_ids.device)
        pad[: local_ids.numel()] = local_ids
        whole = torch.empty((world, max(m, 1)), dtype=torch.int64, device=local_ids.device)
        L.check(lib.rfx_dist_allgather(eng._ctx, pad.data_ptr(), max(m, 1) * 8, whole.data_ptr()), "dist_allgather")
        eng.sync()
        return torch.cat([whole[r, :s] for r, s in enumerate(sizes)])
    world = dist.get_world_size(group)
    cnt = torch.tensor([local_ids.numel()], dtype=torch.int64, device=local_ids.device)
    cnts = _all_gather(cnt, group)
    sizes = [int(c[0]) for c in cnts]
    m = max(sizes) if sizes else 0
    pad = torch.zeros(m, dtype=torch.int64, device=local_ids.device)
    pad[: local_ids.numel()] = local_ids
    bufs = _all_gather(pad, group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


# ---------------------------------------------------------------------------------------------- sharded front-end
class _TorchTransport:
    """rfx_transport_t over torch.distributed: what carries the planner's inter-process exchanges when the processes cannot share an RCCL
    communicator (the tests: two ranks on ONE GPU under gloo).  Device buffers take a host round trip -- test plumbing, not a data path."""

    def __init__(self, engine, group):
        self.eng, self.group = engine, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.calls = 0
        self.struct = L.Transport()
        self.struct.world_rank = L.TR_WORLD_RANK(self._world_rank)
        self.struct.allgather_host = L.TR_ALLGATHER_HOST(self._allgather_host)
        self.struct.allreduce = L.TR_ALLREDUCE(self._allreduce)
        self.struct.allgather_dev = L.TR_ALLGATHER_DEV(self._allgather_dev)

    def _guard(self, fn):
        try:
            fn()
            self.calls += 1
            return L.RFX_OK
        except Exception:  # noqa: BLE001 -- a Python exception must not unwind through the C planner
            import traceback
            traceback.print_exc()
            return -4

    def _world_rank(self, user, pw, pr):
        pw[0], pr[0] = self.world, self.rank
        return L.RFX_OK

    def _gather_bytes(self, raw: bytes):
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        bufs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(bufs, t, group=self.group)
        return b"".join(bytes(b.numpy().tobytes()) for b in bufs)

    def _allgather_host(self, user, pin, nbytes, pout):
        return self._guard(lambda: C.memmove(pout, self._gather_bytes(C.string_at(pin, nbytes)), nbytes * self.world))

    def _dev_bytes(self, d_ptr, nbytes) -> bytes:
        buf = C.create_string_buffer(nbytes)
        L.check(self.eng.lib.rfx_hip_d2h(self.eng._ctx, buf, C.c_void_p(d_ptr), nbytes), "d2h")
        return buf.raw

    def _allreduce(self, user, d_buf, n, typ, op):
        def run():
            h = torch.frombuffer(bytearray(self._dev_bytes(d_buf, n * 8)), dtype=torch.float64 if typ == 1 else torch.int64)
            dist.all_reduce(h, op=(dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX)[op], group=self.group)
            L.check(self.eng.lib.rfx_hip_h2d(self.eng._ctx, C.c_void_p(d_buf), h.numpy().ctypes.data, n * 8), "h2d")
        return self._guard(run)

    def _allgather_dev(self, user, d_in, nbytes, d_out):
        def run():
            whole = self._gather_bytes(self._dev_bytes(d_in, nbytes))
            L.check(self.eng.lib.rfx_hip_h2d(self.eng._ctx, C.c_void_p(d_out), whole, len(whole)), "h2d")
        return self._guard(run)


class ShardedEngine:
    """One process per GPU, every rank holding its row range of every column: the same calls as Engine, answered by the library's planner
    with its inter-process exchange switched on -- under NCCL the lead context's own RCCL communicator (rfx_dist_init; torch.distributed
    only carries the 128-byte id), under gloo a transport over torch.distributed.  Results are replicated on all ranks."""

    def __init__(self, engine, local_rows: int, group=None):
        self.eng = engine
        self.shard = RowShard(local_rows, group)
        self.native = None
        self.transport = None
        lib = engine.lib
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            ident = [None]
            if self.shard.rank == 0:
                buf = C.create_string_buffer(128)
                ident = [buf.raw if lib.rfx_dist_unique_id(buf) == L.RFX_OK else None]
            if self.shard.world > 1:
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ok = ident[0] is not None and lib.rfx_dist_init(engine._ctx, self.shard.world, self.shard.rank, C.c_char_p(ident[0])) == L.RFX_OK
            if self.shard.world > 1:  # every rank takes the same door
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = bool(int(flag[0]))
            if not ok:
                raise L.RfxError("the library's RCCL communicator did not come up on every rank (rfx_dist_init)")
            self.native = (engine, lib)
        elif dist.is_initialized():
            self.transport = _TorchTransport(engine, group)
            L.check(lib.rfx_exec_set_transport(engine._x, C.byref(self.transport.struct)), "exec_set_transport")

    def close(self):
        if self.native:
            L.check(self.native[1].rfx_dist_finalize(self.eng._ctx), "dist_finalize")
            self.native = None
        if self.transport:
            self.eng.lib.rfx_exec_set_transport(self.eng._x, None)
            self.transport = None

    def filter_aggr(self, aggs, where=None, table=None):
        return self.eng.filter_aggr(aggs, where, table, nrows=self.shard.local_rows)

    def where(self, where, table=None) -> torch.Tensor:
        ids = self.eng.where(where, table, row0=self.shard.row0)
        return gather_ids(ids, self.shard.group, self.native)

    def group_by(self, key, aggs, where=None, table=None):
        return self.eng.group_by(key, aggs, where, table)
