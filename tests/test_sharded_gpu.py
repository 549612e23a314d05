"""The planner's SHARDED mode (include/rfx_exec.h) on one GPU: a table split row-range over k shards -- each with its own context, stream and
host thread -- must answer what the unsharded oracle answers, whatever the path: scalar partials folded in shard order, dense tables merged
by the device kernel (the merge ray_select makes of its pool workers' partials, core/aggr.c:163-181), hashed tables re-inserted,
FIRST values read where the rows live, `where` ids concatenated in shard order.  The same through the C operator door: RFX_SHARDS=k
splits every pinned / uploaded column and rfx_select answers from all shards; RFX_EXEC_FORCE_RCCL=1 adds a one-rank RCCL world so that the
fused exchange of the multi-device case (rfx_dist_group_tables_allreduce_all) really runs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import rfo
from test_gpu_parity import check_select, same_f64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NULL = -(2**63)
CHUNK_SMALL = 32768


def table(n, keys=3000, nulls=True):
    t = {"k": rfo.gen_i64(n, 4, keys) - 7, "j": rfo.gen_i64(n, 14, 13), "a": rfo.gen_i64(n, 2, 1_000_000), "ts": rfo.gen_i64(n, 3, 1_000_000),
         "v": rfo.gen_f64(n, 5) - 0.5, "w": rfo.gen_f64(n, 6)}
    if nulls:
        t["a"][::97] = NULL
        t["v"][::89] = np.nan
    return t


QUERIES = [
    {"s": ("sum", "a"), "f": ("sum", "v"), "mn": ("min", "v"), "mx": ("max", "a"), "av": ("avg", "a"), "c": ("count", "v"), "fi": ("first", "w"), "where": ("and", ("<", "a", 600_000), (">", "v", -0.4))},
    {"s": ("sum", ("*", "w", ("-", 1, "v"))), "where": ("or", ("<", "a", 1000), ("and", (">", "w", 0.5), ("<", "v", 0.0)))},
    {"by": "k", "s": ("sum", "v"), "s2": ("sum", "a"), "mn": ("min", "v"), "mx": ("max", "a"), "av": ("avg", "a"), "c": ("count", "v"), "fa": ("first", "a"), "fv": ("first", "w")},
    {"by": "k", "s": ("sum", "w"), "fa": ("first", "a"), "where": (">", "v", -0.25)},
    {"by": {"g1": "j", "g2": "k"}, "s": ("sum", "w"), "c": ("count", "a"), "mx": ("max", "a")},
    {"by": {"b": ("xbar", "ts", 50_000)}, "s": ("sum", "w"), "f": ("first", "a")},
    {"by": "k", **{f"o{i}": (fn, c) for i, (fn, c) in enumerate([("max", "a"), ("sum", "v"), ("min", "a"), ("avg", "v"), ("count", "a"), ("sum", "a"), ("min", "v"), ("max", "v"), ("avg", "a"), ("first", "w"), ("sum", "w")])}},
]


@pytest.mark.parametrize("shards", [2, 3, 5])
def test_sharded_engine_matches_the_oracle(built, shards):
    from rayforce_amd import _lib as L
    from rayforce_amd.engine import Engine
    e = Engine(0, shards=shards)
    try:
        for n in (1_000_003, 2_047, 1):  # uneven shards, shards without rows
            host = table(n)
            dev = {c: e.column(x) for c, x in host.items()}
            for q in QUERIES:
                check_select(e, host, q, dev)
            ids = e.where(("<", "a", 100_000), dev)
            assert np.array_equal(ids.cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 100_000), host)))
        # sparse keys: every shard's hashed table re-inserted into the lead's
        host = table(600_011, keys=40_000)
        host["k"] = host["k"] * 1_000_003 - 5
        dev = {c: e.column(x) for c, x in host.items()}
        check_select(e, host, {"by": "k", "s": ("sum", "w"), "c": ("count", "a"), "f": ("first", "v")}, dev)
        assert e.xstat(L.RFX_XSTAT_MERGES_KERNEL) > 0
        # wide key range: every shard partitions its rows into planes, aggregates them into its own tables under the AGREED scope
        e.tune(flags=CHUNK_SMALL)
        host = table(1_500_007, keys=300_000, nulls=False)
        dev = {c: e.column(x) for c, x in host.items()}
        before = e.stat(0)
        check_select(e, host, {"by": "k", "s": ("sum", "w")}, dev)
        check_select(e, host, {"by": "k", "s": ("sum", "w"), "where": ("<", "a", 300_000)}, dev)
        assert e.stat(0) - before >= shards  # RFX_STAT_PLANE_SCATTER: one scatter per shard and query at least
        e.tune(flags=0)
    finally:
        e.close()


def test_c4_row_range_shards_merge_to_the_unsharded_answer(built):
    """configs[3]: the C3 group-by (1e6 keys, sum of f64) over 4 row-range shards equals the unsharded answer group for group, first-occurrence
    order included -- through the planner, whose merge is what the ranks' exchange computes (MIN of first rows, SUM of sums)."""
    from rayforce_amd.engine import Engine
    n = 6_000_007
    host = {"k": rfo.gen_i64(n, 4, 1_000_000), "v": rfo.gen_f64(n, 5)}
    want = rfo.select({"from": host, "by": "k", "s": ("sum", "v")})
    e = Engine(0, shards=4)
    try:
        e.tune(flags=CHUNK_SMALL)
        dev = {k: e.column(v) for k, v in host.items()}
        r = e.group_by("k", [("sum", "v")], None, dev)
        assert np.array_equal(r["keys"].cpu().numpy(), want["k"])
        first = r["first"].cpu().numpy()
        assert np.all(first[1:] > first[:-1]) and np.array_equal(host["k"][first], want["k"])
        same_f64(r["results"][0].cpu().numpy(), want["s"])
        assert e.stat(0) >= 4 and e.stat(2) >= 4  # plane scatter + plane aggregate on every shard
    finally:
        e.close()


_DOOR = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rfo
from rayforce_amd import hostobj as H, _lib as L
from test_gpu_parity import same_f64
import ctypes as C
ops = H.lib()
ops.rfx_host_bind()
NULL = -(2**63)
n = 2_000_003
host = {"k": rfo.gen_i64(n, 4, 100_000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5), "b": rfo.gen_f64(n, 3), "c": rfo.gen_f64(n, 8), "d": rfo.gen_f64(n, 9)}
host["a"][::101] = NULL
host["k2"] = rfo.gen_i64(n, 14, 13)
host["nn"] = np.full(n, NULL, dtype=np.int64)
host["kk"] = (host["k"] % 2) * 1_000_000   # two groups over a range beyond the small-table form: fewer groups than slices (slice 0 must not be the empty one)
for i, m in enumerate((1000, 1000, 1_000_000, 1000, 1000, 1_000_000)):  # six key columns whose ranges multiply to 1e24 > 2^63: the H2O Q7 shape
    host[f"id{i + 1}"] = rfo.gen_i64(n, 20 + i, m)
tab = H.table(host)
_WIDE = ("or", ("and", ("<", "a", 100_000), (">", "v", 0.1)), ("and", (">", "a", 900_000), ("<", "v", 0.9)), ("and", ("==", "k", 5), ("!=", "b", 0.5)),
         ("and", ("<", "c", 0.05), (">", "d", 0.95)), ("and", (">=", "a", 450_000), ("<=", "a", 460_000)))
queries = [
    {"s": ("sum", "a"), "where": ("<", "a", 100_000)},
    {"s": ("sum", "b"), "c": ("count", "a"), "f": ("first", "v"), "where": ("<", "a", 100_000)},
    {"s": ("sum", "v"), "by": "k"},
    {"s": ("sum", "v"), "f": ("first", "a"), "m": ("max", "a"), "by": "k", "where": ("<", "a", 100_000)},
    {"x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d"), "where": ("and", ("<", "v", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25))},
    {"s": ("sum", ("*", "v", "b")), "where": ("and", ("or", ("<", "a", 5000), ("and", (">", "v", 0.5), ("<", "b", 0.5))), (">", "c", 0.1))},
    {"where": ("<", "a", 3000)},
    {"s": ("sum", "v"), "c": ("count", "a"), "by": "kk"},
    {"s": ("sum", "v"), "by": "kk", "where": ("<", "kk", 5)},   # ONE group
    {"s": ("sum", "v"), "c": ("count", "a"), "m": ("min", "b"), "by": {"g1": "k2", "g2": "k"}},
    {"by": "k", **{f"o{i}": (fn, c) for i, (fn, c) in enumerate([("max", "a"), ("sum", "v"), ("min", "a"), ("avg", "v"), ("sum", "b"), ("avg", "c"), ("min", "d"), ("count", "a")])}},  # five argument columns: two passes, the same slices
    # where: trees beyond the fused form (ten comparisons): every shard evaluates the tree over ITS rows into a 0 / 1 column that the query reads as one comparison
    {"s": ("sum", "v"), "c": ("count", "a"), "f": ("first", "b"), "by": "k", "where": _WIDE},
    {"s": ("sum", "b"), "m": ("min", "a"), "where": _WIDE},
    {"where": ("and", _WIDE, ("<", "a", 200_000))},
    # key tuples beyond a 64-bit composite key: every shard groups ITS rows on the reference's row hash, the hashed tables are re-inserted, and the
    # tuples are PROVEN by a (min, max) pair per key column riding through the same merge (index_group_list, core/index.c:2731-2790)
    {"s": ("sum", "v"), "c": ("count", "a"), "m": ("max", "b"), "by": {f"id{i + 1}": f"id{i + 1}" for i in range(6)}},
    # ... nulls among the key tuples too (MIN / MAX skip nulls: a null key rides through the proof as max + 1 and comes back as the null); a key column of nulls only
    {"s": ("sum", "v"), "c": ("count", "k"), "by": {"a1": "a", "g": "k2"}},
    {"s": ("sum", "v"), "m": ("min", "b"), "by": {"x": "nn", "g": "k2", "y": "a"}},
    # ... FIRST values beside them: the result stays whole on the lead, and so do the proof passes' key columns (how a result is left is
    # decided for the QUERY: every pass of it -- the proof's, a second chunk of aggregates -- follows)
    {"f": ("first", "v"), "s": ("sum", "v"), "by": {f"id{i + 1}": f"id{i + 1}" for i in range(6)}},
    {"f": ("first", "b"), "c": ("count", "k"), "by": {"a1": "a", "g": "k2"}},
    {"by": "k", **{f"o{i}": (fn, c) for i, (fn, c) in enumerate([("max", "a"), ("sum", "v"), ("min", "a"), ("avg", "v"), ("sum", "b"), ("avg", "c"), ("min", "d"), ("first", "d")])}},
    # comparison operands that are element-wise expressions: every shard evaluates ITS rows into a scratch column of its own
    {"s": ("sum", "v"), "c": ("count", "a"), "where": ("or", ("==", ("div", "a", 1000), 7), ("and", (">", ("*", "v", 2.0), 1.5), ("!=", "k", 3)))},
    {"s": ("sum", "b"), "by": "k", "where": ("and", ("<", ("+", "v", "b"), 0.7), (">", ("-", "a", "k"), 1000))},
]
def ask(q, t):
    d = H.select_dict(q, t)
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    assert ops.rfx_last_select_on_gpu() == 1, (q, ops.rfx_ops_last_error())
    out = H.table_to_numpy(r)
    ops.rfx_host_drop(r); ops.rfx_host_drop(d)
    return out
def same(got, want, q):
    assert list(got) == list(want), (list(got), list(want))
    for name in want:
        g, w = got[name], want[name]
        assert g.dtype == w.dtype and g.shape == w.shape, name
        if w.dtype == np.float64 and name in q and q[name][0] in ("sum", "avg"):
            same_f64(g, w)
        else:
            assert np.array_equal(g, w, equal_nan=w.dtype == np.float64), name
for rep in range(2):  # first touch uploads every column shard by shard; the second round finds them resident
    for q in queries:
        same(ask(q, tab), rfo.select({"from": host, **q}), q)
assert ops.rfx_ops_shards() == SHARDS, ops.rfx_ops_shards()
# the same table as DEVICE columns (one allocation each: the shards take their row ranges of it)
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
dev = {c: eng.column(x) for c, x in host.items()}
dtab = H.device_table(dev)
torch.cuda.synchronize()
for q in queries:
    same(ask(q, dtab), rfo.select({"from": host, **q}), q)
x = C.c_void_p(ops.rfx_ops_exec())
assert ops.rfx_exec_shards(x) == SHARDS
assert ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_KERNEL) > 0
if os.environ.get("RFX_EXEC_FORCE_RCCL"):
    assert ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_RCCL) > 0 and ops.rfx_dist_calls(C.c_void_p(ops.rfx_exec_ctx(x, 0))) > 0
# grouped results without FIRST values came back as one slice per shard when every shard owns one (the FIRST ones stay whole on the lead)
assert (ops.rfx_exec_stat(x, L.RFX_XSTAT_SLICED) > 0) == bool(os.environ.get("RFX_EXEC_SLICE_SHARDS")), ops.rfx_exec_stat(x, L.RFX_XSTAT_SLICED)
# ---- the operators beside rfx_select, over the shards (the reference runs every FN_AGGR built-in over its pool: aggr_map core/aggr.c:375,
# unop_fold core/math.c:2176-2231): folds of a vector, of a lazy MAPFILTER (values, ids) pair, of a MAPGROUP (values, index) pair; where
def atom_of(r):
    assert not H.is_error(r), H.error_text(r)
    v = C.c_double.from_address(r + 8).value if H.header(r).type == -H.T_F64 else C.c_int64.from_address(r + 8).value
    ops.rfx_host_drop(r)
    return v
def close(got, want, what):
    if want is None:  # null
        assert got == NULL or got != got, (what, got)
    elif isinstance(want, float):
        assert abs(got - want) <= 1e-9 * max(abs(want), 1e-300), (what, got, want)
    else:
        assert got == want, (what, got, want)
before = ops.rfx_exec_stat(x, L.RFX_XSTAT_QUERIES)
sel = rfo.where(rfo.cmp("<", host["a"], 400_000))
for cname, fns in (("a", ("sum", "min", "max", "avg", "count", "first")), ("v", ("sum", "min", "max", "avg", "first"))):
    vec, idv = H.vector(host[cname]), H.vector(sel)
    pair = H.list_of([vec, idv])
    H.header(pair).type = 71  # TYPE_MAPFILTER
    for fn in fns:
        close(atom_of(getattr(ops, f"rfx_{fn}")(vec)), len(host[cname]) if fn == "count" else rfo.fold(fn, host[cname]), (cname, fn, "vector"))
        close(atom_of(getattr(ops, f"rfx_{fn}")(pair)), len(sel) if fn == "count" else rfo.fold(fn, host[cname][sel]), (cname, fn, "mapfilter"))
    ops.rfx_host_drop(pair)
# ids that do not ascend through the shards are not a filter's: said so (no host here to take them), never answered from the wrong rows
pair = H.list_of([H.vector(host["v"]), H.vector(sel[::-1].copy())])
H.header(pair).type = 71
r = ops.rfx_sum(pair)
assert H.is_error(r), "descending ids must not be answered shard by shard"
ops.rfx_host_drop(r); ops.rfx_host_drop(pair)
# MAPGROUP: the reference's group index over k (IDS form: a group id per row; SHIFT form: the key table + the source column), values folded per group
gids, firsts, groups, dense = rfo.group_index(host["k"], None)
assert dense
def host_index(itype, shift, group_ids, source):
    ix = ops.rfx_host_list(7)
    arr = (C.c_void_p * 7).from_address(H.payload(ix))
    arr[0], arr[1] = H.atom(itype), H.atom(groups)
    arr[2] = H.vector(group_ids)
    arr[3] = H.atom(shift if itype == 1 else NULL)
    if itype == 1:
        arr[4] = H.vector(source)
    arr[6] = H.vector(firsts)
    return ix
kmin = int(host["k"].min())
table = np.full(int(host["k"].max()) - kmin + 1, NULL, np.int64)
table[host["k"] - kmin] = gids
want_by = rfo.select({"from": host, "by": "k", "s": ("sum", "v"), "m": ("max", "a"), "c": ("count", "a"), "av": ("avg", "v"), "f": ("first", "a")})
for ix in (host_index(0, 0, gids, None), host_index(1, kmin, table, host["k"])):
    for fn, cname, out in (("sum", "v", "s"), ("max", "a", "m"), ("count", "a", "c"), ("avg", "v", "av"), ("first", "a", "f")):
        pair = H.list_of([H.vector(host[cname]), ix])
        H.header(pair).type = 72  # TYPE_MAPGROUP
        r = getattr(ops, f"rfx_{fn}")(pair)
        assert not H.is_error(r), H.error_text(r)
        got = H.to_numpy(r)
        if got.dtype == np.float64:
            same_f64(got, want_by[out])
        else:
            assert np.array_equal(got, want_by[out]), (fn, cname)
        ops.rfx_host_drop(r)
# where: a B8 mask -> ascending global row ids, every shard its rows
m = (host["a"] < 300_000) & (host["v"] > 0.25)
r = ops.rfx_where(H.vector(m))  # (a bool array is a B8 vector)
assert not H.is_error(r), H.error_text(r)
assert np.array_equal(H.to_numpy(r), np.nonzero(m)[0])
ops.rfx_host_drop(r)
assert ops.rfx_exec_stat(x, L.RFX_XSTAT_QUERIES) - before >= 30   # all of it went through the planner's shards
# element-wise operators: every shard its rows, the pieces of the result at their offsets (cmp_map core/cmp.c:35-68, binop_map core/math.c:2280-2345)
va, vv, vb = H.vector(host["a"]), H.vector(host["v"]), H.vector(host["b"])
def vec_of(r):
    assert not H.is_error(r), H.error_text(r)
    out = H.to_numpy(r).copy()
    ops.rfx_host_drop(r)
    return out
assert np.array_equal(vec_of(ops.rfx_lt(va, H.atom(400_000))).astype(bool), rfo.cmp("<", host["a"], 400_000).astype(bool))
assert np.array_equal(vec_of(ops.rfx_ge(vv, vb)).astype(bool), rfo.cmp(">=", host["v"], host["b"]).astype(bool))
assert np.array_equal(vec_of(ops.rfx_ne(va, H.atom(NULL))).astype(bool), rfo.cmp("!=", host["a"], NULL).astype(bool))
got = vec_of(ops.rfx_mul(vv, vb)); assert got.dtype == np.float64 and np.array_equal(got, host["v"] * host["b"])
got = vec_of(ops.rfx_add(va, H.atom(7))); want = host["a"] + 7; want[host["a"] == NULL] = NULL
assert got.dtype == np.int64 and np.array_equal(got, want)
m1, m2, m3 = host["a"] < 300_000, host["v"] > 0.25, host["b"] < 0.9
masks = (C.c_void_p * 3)(H.vector(m1), H.vector(m2), H.vector(m3))
assert np.array_equal(vec_of(ops.rfx_and(masks, 3)).astype(bool), m1 & m2 & m3)
assert np.array_equal(vec_of(ops.rfx_or(masks, 2)).astype(bool), m1 | m2)
# at: a filter's ids (ascending) are cut at the shards' rows; descending ids need the column whole (no host here: said so)
assert np.array_equal(vec_of(ops.rfx_at(vv, H.vector(sel))), host["v"][sel])
r = ops.rfx_at(vv, H.vector(sel[::-1].copy()))
assert H.is_error(r) and "whole on one device" in H.error_text(r)
ops.rfx_host_drop(r)
# group: the key-table form of the index (INDEX_TYPE_SHIFT) from the planner's sharded group-by
r = ops.rfx_group(H.vector(host["k"]))
assert not H.is_error(r), H.error_text(r)
items = H.list_items(r)
assert C.c_int64.from_address(items[0] + 8).value == 1 and C.c_int64.from_address(items[1] + 8).value == groups and C.c_int64.from_address(items[3] + 8).value == kmin
assert np.array_equal(H.to_numpy(items[2]), table) and np.array_equal(H.to_numpy(items[6]), firsts)
ops.rfx_host_drop(r)
# ... and the per-row form (INDEX_TYPE_IDS: a key range beyond INDEX_SCOPE_LIMIT = 524 288): every row mapped through the key table where it lives
kbig = rfo.gen_i64(n, 31, 900_000)
gids_b, firsts_b, groups_b, dense_b = rfo.group_index(kbig, None)
assert dense_b
r = ops.rfx_group(H.vector(kbig))
assert not H.is_error(r), H.error_text(r)
items = H.list_items(r)
assert C.c_int64.from_address(items[0] + 8).value == 0 and C.c_int64.from_address(items[1] + 8).value == groups_b
assert np.array_equal(H.to_numpy(items[2]), gids_b) and np.array_equal(H.to_numpy(items[6]), firsts_b)
ops.rfx_host_drop(r)
# sparse keys stay the host's (said so without one)
r = ops.rfx_group(H.vector(kbig * 1_000_003))
assert H.is_error(r) and "whole on one device" in H.error_text(r), H.error_text(r)
ops.rfx_host_drop(r)
# round 6: the equi-joins run over the shards -- a broadcast join: the left table's rows stay on their shards, the right table is kept whole on every shard,
# every shard probes and gathers its own rows -- bit for bit the oracle's (= the reference's golden joins): one key (dense / hashed table), two keys
# (composite), wide tuples with nulls (row hash + the tuple check), an empty match set
import golden_cases as G
for case in G.join_cases():
    name, keys, left, right, want_lj, want_ij = case
    lt_, rt_, ks_ = H.table(left), H.table(right), H.symbols(keys)
    args = (C.c_void_p * 3)(ks_, lt_, rt_)
    for fn, ora in (("rfx_left_join", rfo.left_join), ("rfx_inner_join", rfo.inner_join)):
        out = getattr(ops, fn)(args, 3)
        assert not H.is_error(out), (name, fn, H.error_text(out))
        got, o = H.table_to_numpy(out), ora(keys, left, right)
        assert list(got) == list(o), (name, fn)
        for c in o:
            assert got[c].dtype == o[c].dtype and np.array_equal(got[c].view(np.int64), o[c].view(np.int64)), (name, fn, c)
        ops.rfx_host_drop(out)
    for o in (lt_, rt_, ks_):
        ops.rfx_host_drop(o)
# ... at a size where every shard holds many left rows: 2e6 left rows against 30 000 right rows (a third of the left keys find no partner), asked twice
# (the second time the right table's whole-column copies and the left pieces are cache hits)
right = {"k": rfo.gen_i64(30_000, 41, 70_000), "k2": rfo.gen_i64(30_000, 42, 13), "w": rfo.gen_f64(30_000, 43), "v": rfo.gen_f64(30_000, 44) + 5.0}
left = {"k": host["k"], "k2": host["k2"], "a": host["a"], "v": host["v"]}
lt_, rt_ = H.table(left), H.table(right)
st0 = H.to_numpy(ops.rfx_stats(0))
for rep in range(2):
    for keys in (["k"], ["k", "k2"]):
        ks_ = H.symbols(keys)
        args = (C.c_void_p * 3)(ks_, lt_, rt_)
        for fn, ora in (("rfx_left_join", rfo.left_join), ("rfx_inner_join", rfo.inner_join)):
            out = getattr(ops, fn)(args, 3)
            assert not H.is_error(out), (keys, fn, H.error_text(out))
            got, o = H.table_to_numpy(out), ora(keys, left, right)
            assert list(got) == list(o), (keys, fn)
            for c in o:
                assert got[c].dtype == o[c].dtype and np.array_equal(got[c].view(np.int64), o[c].view(np.int64)), (keys, fn, c)
            ops.rfx_host_drop(out)
        ops.rfx_host_drop(ks_)
# a table joined WITH ITSELF: the same column objects are needed as row ranges (left) and whole on every shard (right) -- two cache entries per vector, both alive
selfc = {"k": rfo.gen_i64(50_003, 45, 20_000), "v": rfo.gen_f64(50_003, 46)}
st_ = H.table(selfc)
for fn, ora in (("rfx_left_join", rfo.left_join), ("rfx_inner_join", rfo.inner_join)):
    ks_ = H.symbols(["k"])
    out = getattr(ops, fn)((C.c_void_p * 3)(ks_, st_, st_), 3)
    assert not H.is_error(out), (fn, H.error_text(out))
    got, o = H.table_to_numpy(out), ora(["k"], selfc, selfc)
    assert list(got) == list(o) and all(np.array_equal(got[c].view(np.int64), o[c].view(np.int64)) for c in o), fn
    ops.rfx_host_drop(out)
    ops.rfx_host_drop(ks_)
kcol = H.list_items(H.list_items(st_)[1])[0]
assert H.header(kcol).rc == 3  # the table's reference + one per cached copy (row ranges, whole)
ops.rfx_host_drop(st_)
st1 = H.to_numpy(ops.rfx_stats(0))
assert st1[2] - st0[2] == 10 and st1[3] == st0[3], (st0, st1)  # ten joins on the device, none handed back
assert st1[4] - st0[4] <= 4 + 4 + 4, (st0, st1)               # uploads: four left columns (row ranges) + four right columns (whole), once; the self-join's two columns twice
# ... and `update ... where / by` (round 6): every shard writes its rows of the new column (selection as a 0 / 1 column per shard, values element-wise per shard or --
# under by: -- every row's group aggregate looked up in the value table of the merged groups); the families of tests/test_ops_gpu.py::UPDATES against the oracle
from test_ops_gpu import UPDATES, check
uh = {"k": rfo.gen_i64(n, 4, 300), "a": host["a"], "v": host["v"]}
ut = H.table(uh)
for q in UPDATES + [{"f": ("first", "a"), "mn": ("min", "v"), "where": (">", "v", 0.5), "by": "k"}, {"s": ("sum", "a"), "where": ("<", "a", -5), "by": "k"}]:
    d = H.select_dict(q, ut)
    r = ops.rfx_update(d)
    assert r and not H.is_error(r), (q, H.error_text(r))
    assert ops.rfx_last_select_on_gpu() == 1, q
    check(H.table_to_numpy(r), rfo.update({"from": uh, **q}))
    ops.rfx_host_drop(r)
    ops.rfx_host_drop(d)
# the shapes that stay the host's say so
for q, why in (({"a": 1.5, "where": ("<", "a", 10)}, "value type differs"), ({"t": ("sum", "v"), "by": "kw"}, "sparse or null keys")):
    t2 = H.table({**uh, "kw": uh["k"] * 1_000_003})
    d = H.select_dict(q, t2)
    r = ops.rfx_update(d)
    assert H.is_error(r) and why in H.error_text(r), (q, H.error_text(r))
    for o in (r, d, t2):
        ops.rfx_host_drop(o)
ops.rfx_host_drop(ut)
for o in (lt_, rt_):
    ops.rfx_host_drop(o)
print("DOOR-OK")
'''


@pytest.mark.parametrize("shards,rccl,sliced", [(4, False, False), (3, True, False), (4, False, True), (3, True, True)])
def test_sharded_operator_door(built, shards, rccl, sliced):
    """rfx_select with RFX_SHARDS=k in a process of its own (the operator layer's shards are fixed at its first call).  sliced: every shard
    owns a slice of every group-by result (RFX_EXEC_SLICE_SHARDS=1: the sharded tail of the multi-device case -- every owner ranks the merged
    tables, emits its range of the groups and copies it into the host table from its own thread -- on one GPU)."""
    env = dict(os.environ, RFX_SHARDS=str(shards), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RFX_EXEC_FORCE_RCCL", None)
    env.pop("RFX_EXEC_SLICE_SHARDS", None)
    if rccl:
        env["RFX_EXEC_FORCE_RCCL"] = "1"
    if sliced:
        env["RFX_EXEC_SLICE_SHARDS"] = "1"
    code = f"ROOT = {ROOT!r}\nSHARDS = {shards}\n" + _DOOR
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "DOOR-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


_PLANE_DOOR = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rfo
from rayforce_amd import hostobj as H, _lib as L
from test_gpu_parity import same_f64
import ctypes as C
ops = H.lib()
ops.rfx_host_bind()
n = 17_000_003   # 4 shards x 4.25e6 rows: every shard above the 2^22-row threshold of the plane kernels, DEFAULT thresholds
host = {"k": rfo.gen_i64(n, 4, 1_000_000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5), "b": rfo.gen_f64(n, 3)}
host["ks"] = host["k"] * 1_000_003 - 77   # sparse: the hash-partitioned planes, hashed tables re-inserted on the lead
tab = H.table(host)
queries = [
    {"s": ("sum", "v"), "by": "k"},                                     # configs[2] / [3]
    {"s": ("sum", "v"), "by": "k", "where": ("<", "a", 100_000)},       # the metric's shape
    {"s": ("sum", "v"), "x": ("avg", "b"), "c": ("count", "a"), "m": ("max", "a"), "by": "k", "where": ("<", "a", 500_000)},
    {"s": ("sum", "v"), "f": ("first", "a"), "by": "k"},                # FIRST values: the whole result on the lead
    {"s": ("sum", "v"), "by": "ks"},
]
def ask(q):
    d = H.select_dict(q, tab)
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    assert ops.rfx_last_select_on_gpu() == 1, (q, ops.rfx_ops_last_error())
    out = H.table_to_numpy(r)
    ops.rfx_host_drop(r); ops.rfx_host_drop(d)
    return out
x = None
for rep in range(2):
    for q in queries:
        got, want = ask(q), rfo.select({"from": host, **q})
        assert list(got) == list(want)
        for name in want:
            g, w = got[name], want[name]
            assert g.dtype == w.dtype and g.shape == w.shape, name
            if w.dtype == np.float64 and name in q and q[name][0] in ("sum", "avg"):
                same_f64(g, w)
            else:
                assert np.array_equal(g, w), name
x = C.c_void_p(ops.rfx_ops_exec())
assert ops.rfx_exec_shards(x) == SHARDS
scatter = sum(int(ops.rfx_hip_ctx_stat(C.c_void_p(ops.rfx_exec_ctx(x, s)), 0)) for s in range(SHARDS))
aggregate = sum(int(ops.rfx_hip_ctx_stat(C.c_void_p(ops.rfx_exec_ctx(x, s)), 2)) for s in range(SHARDS))
assert scatter >= 2 * 5 * SHARDS and aggregate >= 2 * 5 * SHARDS, (scatter, aggregate)   # the plane kernels ran on EVERY shard for every query
assert (ops.rfx_exec_stat(x, L.RFX_XSTAT_SLICED) > 0) == bool(os.environ.get("RFX_EXEC_SLICE_SHARDS"))
print("PLANE-DOOR-OK")
'''


@pytest.mark.parametrize("sliced", [False, True])
def test_sharded_door_at_plane_path_sizes(built, sliced):
    """The C door over 4 shards at 1.7e7 rows with the DEFAULT thresholds: every shard runs k_plane_scatter / k_plane_aggregate (and the
    hash-partitioned planes for sparse keys), the tables merge, and -- sliced -- every shard emits and reads back its range of the 1e6 groups."""
    env = dict(os.environ, RFX_SHARDS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RFX_EXEC_FORCE_RCCL", None)
    env.pop("RFX_EXEC_SLICE_SHARDS", None)
    if sliced:
        env["RFX_EXEC_SLICE_SHARDS"] = "1"
    code = f"ROOT = {ROOT!r}\nSHARDS = 4\n" + _PLANE_DOOR
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "PLANE-DOOR-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("seeds", [(0, 6), (100, 106)])
@pytest.mark.parametrize("shards", [1, 3])
def test_fuzz_large_collected(built, seeds, shards):
    """tools/fuzz_large.py's loop as collected seeds: random group-bys at plane-path sizes (2^22 .. 2^24 rows) against the oracle, over one
    and over three shards of the device."""
    env = dict(os.environ, FUZZ_SHARDS=str(shards), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_large.py"), str(seeds[0]), str(seeds[1])], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and f"done seeds {seeds[0]}..{seeds[1]}: 0 failures" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
