"""world_size-2 `gloo` test of the row-range sharded merge (rayforce_amd/dist.py) on CPU.

The per-rank LOCAL step (the HIP kernels) is replaced here by the CPU oracle producing the same partial states -- the
thing under test is everything after it: the all_gather / all_reduce choreography, the host-side partial algebra
(rfx_partial_merge / rfx_agg_finalize from librfx.so) and the table merge rules, compared with the unsharded oracle."""
import ctypes as C
import os
import socket
import struct

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NULL = -(2**63)
INF = 2**63 - 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _f64_bits(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _ord(x):
    b = _f64_bits(x)
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFF)


def _local_partials(rfo, L, t, where, aggs):
    """What rfx_hip_filter_aggr leaves on one rank, computed by the oracle."""
    n = len(next(iter(t.values())))
    ids = rfo.where(rfo.mask_of(where, t)) if where else np.arange(n)
    parts = (L.Partial * (len(aggs) + 1))()
    for i, (fn, col) in enumerate(aggs):
        p = parts[i]
        p.pos = INF
        x = t[col][ids]
        f64 = x.dtype == np.float64
        ok = ~np.isnan(x) if f64 else (x != NULL)
        p.cnt = int(ok.sum())
        if fn in ("sum", "avg"):
            if f64:
                p.fsum = float(x[ok].sum())
            else:
                p.isum = int(x[ok].astype(np.uint64).sum().astype(np.int64))
        elif fn in ("min", "max") and p.cnt:
            v = x[ok].min() if fn == "min" else x[ok].max()
            p.ext = _f64_bits(v) if f64 else int(v)
        elif fn == "count":
            p.cnt = len(ids)
    parts[len(aggs)].cnt = len(ids)
    parts[len(aggs)].pos = INF
    return torch.frombuffer(bytearray(bytes(parts)), dtype=torch.uint8).clone()


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rfo
        from rayforce_amd import _lib as L
        from rayforce_amd import dist as D
        rfo.set_threads(2)
        n = 60_000
        full = {"k": rfo.gen_i64(n, 4, 700), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5) - 0.5}
        full["a"][::97] = NULL
        full["v"][::89] = np.nan
        cut = [0, 23_456, n]  # uneven shards
        mine = {c: x[cut[rank]:cut[rank + 1]] for c, x in full.items()}
        shard = D.RowShard(len(mine["k"]))
        assert shard.row0 == cut[rank] and shard.total_rows == n

        # ---- scalar aggregates: all_gather of partials + host fold ----
        where = ("and", ("<", "a", 600_000), (">", "v", -0.4))
        aggs = [("sum", "a"), ("sum", "v"), ("min", "v"), ("max", "a"), ("avg", "v"), ("avg", "a"), ("count", "a")]
        kinds = [L.AGGS[f] for f, _ in aggs]
        ctypes_ = [L.RFX_F64 if full[c].dtype == np.float64 else L.RFX_I64 for _, c in aggs]
        vals, sel = D.merge_scalar_partials(_local_partials(rfo, L, mine, where, aggs), kinds, ctypes_)
        want = rfo.select({"from": full, "where": where, **{f"o{i}": a for i, a in enumerate(aggs)}})
        for i, v in enumerate(vals):
            w = want[f"o{i}"][0]
            if isinstance(v, float):
                assert abs(v - w) <= 1e-9 * abs(w), (aggs[i], v, w)
            else:
                assert v == int(w), (aggs[i], v, w)
        assert sel == int(rfo.mask_of(where, full).sum())

        # ---- dense group tables: scope agreement + all_reduce with the per-kind ops ----
        kmin, kmax, seen = D.allreduce_scope(int(mine["k"].min()), int(mine["k"].max()), len(mine["k"]), torch.device("cpu"))
        assert (kmin, kmax, seen) == (int(full["k"].min()), int(full["k"].max()), n)
        rng = kmax - kmin + 1
        gaggs = [("sum", "v"), ("sum", "a"), ("min", "v"), ("max", "a"), ("avg", "a"), ("count", "v")]
        gk = [L.AGGS[f] for f, _ in gaggs]
        gf = [full[c].dtype == np.float64 for _, c in gaggs]
        layout = [("first", None)]
        rows = [np.full(rng, INF, np.int64)]
        slot = mine["k"] - kmin
        np.minimum.at(rows[0], slot, np.arange(len(slot)) + shard.row0)
        for a, (fn, col) in enumerate(gaggs):
            x = mine[col]
            f64 = x.dtype == np.float64
            ok = ~np.isnan(x) if f64 else (x != NULL)
            if fn == "sum" and f64:
                acc = np.zeros(rng)
                np.add.at(acc, slot, x)  # NaN sticky by IEEE
                rows.append(acc.view(np.int64)); layout.append(("acc", a))
            elif fn == "sum":
                acc = np.zeros(rng, np.int64)
                np.add.at(acc, slot[ok], x[ok])
                cnt = np.zeros(rng, np.int64)
                np.add.at(cnt, slot[~ok], 1)
                rows += [acc, cnt]; layout += [("acc", a), ("cnt", a)]
            elif fn in ("min", "max"):
                img = np.array([_ord(y) for y in x[ok]], np.int64) if f64 else x[ok]
                acc = np.full(rng, INF if fn == "min" else NULL, np.int64)
                (np.minimum if fn == "min" else np.maximum).at(acc, slot[ok], img)
                rows.append(acc); layout.append(("acc", a))
            elif fn == "avg":
                acc = np.zeros(rng)
                np.add.at(acc, slot[ok], x[ok].astype(np.float64))
                cnt = np.zeros(rng, np.int64)
                np.add.at(cnt, slot[ok], 1)
                rows += [acc.view(np.int64), cnt]; layout += [("acc", a), ("cnt", a)]
            else:
                acc = np.zeros(rng, np.int64)
                np.add.at(acc, slot, 1)
                rows.append(acc); layout.append(("acc", a))
        store = torch.from_numpy(np.stack(rows))
        D.allreduce_tables(store, layout, gk, gf)
        merged = store.numpy()
        # finalise like group_final / k_group_emit and compare with the unsharded oracle (first-occurrence order)
        first = merged[0]
        occ = np.nonzero(first != INF)[0]
        order = occ[np.argsort(first[occ], kind="stable")]
        want = rfo.select({"from": full, "by": "k", **{f"o{i}": a for i, a in enumerate(gaggs)}})
        assert np.array_equal(order + kmin, want["k"])
        r = 1
        for a, (fn, col) in enumerate(gaggs):
            cell = merged[r][order]
            w = want[f"o{a}"]
            if fn == "sum" and gf[a]:
                got = cell.view(np.float64)
                assert np.array_equal(np.isnan(got), np.isnan(w)) and np.allclose(got[~np.isnan(w)], w[~np.isnan(w)], rtol=1e-9, atol=0)
                r += 1
            elif fn == "sum":
                nulls = merged[r + 1][order]
                assert np.array_equal(np.where(nulls > 0, NULL, cell), w)
                r += 2
            elif fn == "min":
                img = np.array([_ord(y) for y in w], np.int64)
                assert np.array_equal(cell, img)
                r += 1
            elif fn == "max":
                assert np.array_equal(cell, w)
                r += 1
            elif fn == "avg":
                cnt = merged[r + 1][order]
                got = cell.view(np.float64) / np.where(cnt == 0, 1, cnt)
                assert np.allclose(got[cnt > 0], w[cnt > 0], rtol=1e-9, atol=0)
                r += 2
            else:
                assert np.array_equal(cell, w)
                r += 1

        # ---- where ids: per-rank ascending global ids concatenate in rank order ----
        ids = torch.from_numpy(rfo.where(rfo.mask_of(("<", "a", 100_000), mine)) + shard.row0)
        allids = D.gather_ids(ids)
        assert np.array_equal(allids.numpy(), rfo.where(rfo.mask_of(("<", "a", 100_000), full)))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_row_sharded_merge_world2_gloo(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
