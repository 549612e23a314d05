#!/usr/bin/env python
"""bench.py -- the hot-path benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3w|c2|c2b|c3|c5|...] [--scaling strong|weak] [--rows R] [--no-cpu-baseline] [--no-also]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Two ways to N GPUs, both through the product's door (the C operator rfx_select) and both planned by the library's one planner (rfx_exec.c):
  * under a launcher (WORLD_SIZE set: the driver's `python -m torch.distributed.run ... bench.py --gpus N`): ONE PROCESS PER GPU; every rank hands
    rfx_select its row range as device columns, the operator layer's context joins an RCCL communicator (rfx_ops_dist_init) and every rank gets
    the whole answer -- one exchange of scopes + one fused all-reduce of the group tables per query;
  * `python bench.py --gpus N` without a launcher: ONE PROCESS, N DEVICES (rfx_ops_set_shards) -- the evaluator process that owns the node:
    every shard's pass on its own host thread, the partial tables merged by one fused RCCL exchange issued from that process.
A rank count that differs from --gpus is an error, never a silent 1-GPU run.

A "step" = one execution of the query over the HBM-resident synthetic columns of this rank (kernels + the one merge
collective + result read-back).  The table is rows [0, TOTAL) generated on the device with the counter-based splitmix64 of
SURVEY 8d; rank r holds the row range [r * TOTAL / N, (r + 1) * TOTAL / N).  --scaling strong (default; SURVEY 8d C4/C5: the
SAME 1e9 / 2e9 rows split N ways): TOTAL = the workload's BASELINE size; --scaling weak: TOTAL = N x that size.
`value` = TOTAL / max-over-ranks step time.  At N = 1 through rfx_select (round 6): `value` and `roofline.frac` are taken on the MEDIAN step (SURVEY 8d defines the
metric so); `ms_per_step` stays the mean of the timed loop, `value_mean` / `roofline.frac_mean` ride beside, `steps_ms_in_order` lists every timed step, and
`roofline.dominant_kernel` is the scatter kernel alone by HIP events on its stream.

Workloads (BASELINE.json configs / SURVEY 8d):
  c2   configs[1]  select sum(a) where a < 100000      a: i64[1e9] in [0,1e6)              8 B/row
  c2b  north-star  select sum(b) where a < 100000      + b: f64[1e9]                       16 B/row
  c3   configs[2]  select sum(v) by k                  k: i64[1e9] in [0,1e6), v: f64      16 B/row
  c1   configs[0]  (sum v)                             v: f64[1e7]                         8 B/row   (plumbing case)
  c3w  metric      select sum(v) by k where a < 100000 k, v as c3 + a as c2                24 B/row  <- default, `value` (filter->group-by->sum)
  q2   8f-1        select sum(v) by {id1, id2}         id1, id2: i64[1e9] in [0,100), v    24 B/row
  q1   8f-3        TPC-H Q1 shape: 8 aggregates, two of them nested expressions, by {rf, ls} where sd <= 2400  56 B/row
  k9   a10         select sum(v) by k, sparse keys (range > rows: open-addressing path)          16 B/row
  x6   8f-3        select sum(p*d) where q<24 and .05<=d<=.07   p, d: f64[1e9], q: i64[1e9]   24 B/row (TPC-H Q6 shape)
  c5   configs[4]  avg,min,max(d) where a<.316228 & b>.683772 & c!=.25   4 x f64[2e9]       32 B/row
One JSON line on stdout (rank 0); everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3-7.0 TB/s achievable)
METRIC = "rows/s on filter→group-by→sum, 1e9-row i64/f64; % HBM roofline at 1/2/4/8 GPU"


_T0 = time.perf_counter()


def log(*a):
    print(f"[{time.perf_counter() - _T0:7.1f}s]", *a, file=sys.stderr, flush=True)


WORKLOADS = {
    "c1": dict(desc="configs[0]: (sum v), v f64[1e7] uniform [0,1), seed 1 -- the reference's `make bench` plumbing case, here on the GPU", rows=10_000_000,
               bytes_per_row=8, dtype="f64", kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<1, 1, 4, 1, 0, false> without hiprtc)"),
    "c2": dict(desc="configs[1]: select sum(a) where a < 100000, a i64 uniform [0,1e6), seed 2", rows=1_000_000_000, bytes_per_row=8, dtype="int64",
               kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<1, 1, 4, 1, 0, false> without hiprtc)"),
    "c2_1pct": dict(desc="C2 at 1 % selectivity (a < 10000) -- SURVEY 8d asks for 1 / 10 / 50 %", rows=1_000_000_000, bytes_per_row=8, dtype="int64",
                    kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<1, 1, 4, 1, 0, false> without hiprtc)"),
    "c2_50pct": dict(desc="C2 at 50 % selectivity (a < 500000)", rows=1_000_000_000, bytes_per_row=8, dtype="int64", kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<1, 1, 4, 1, 0, false> without hiprtc)"),
    "c2b": dict(desc="north-star: select sum(b) where a < 100000, a i64 seed 2, b f64 seed 3", rows=1_000_000_000, bytes_per_row=16, dtype="f64",
                kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<2, 1, 4, 1, 0, false> without hiprtc)"),
    "c3": dict(desc="configs[2]: select sum(v) by k, k i64 uniform [0,1e6) seed 4, v f64 seed 5", rows=1_000_000_000, bytes_per_row=16, dtype="f64",
               kernel="k_plane_scatter<2, 0, 1, 7, 5> (scope + partition into 8 + 4-byte planes in one pass; the 16-byte chunk kernels for skewed keys) + k_plane_aggregate<1024, 1, true>"),
    "c3w": dict(desc="metric shape filter->group-by->sum: select sum(v) by k where a < 100000 (10 %), k/v as C3, a as C2", rows=1_000_000_000,
                bytes_per_row=24, dtype="f64", kernel="k_plane_scatter<3, 1, 1, 7, 5> (filter + scope + partition into 8 + 4-byte planes in one pass) + k_plane_aggregate<1024, 1, true>"),
    "q2": dict(desc="several by: columns (H2O Q2 shape): select sum(v) by {id1, id2}, id1/id2 i64 uniform [0,100) seeds 10/11, v f64 seed 5", rows=1_000_000_000,
               bytes_per_row=24, dtype="f64", kernel="k_part_scope_hist<1, 0> x2 + k_group_dense<3, true, 1024>"),
    "q7": dict(desc="key tuples beyond the composite key (H2O Q7 shape, row-hash path): select sum(v), count by {id1..id6}; id1,id2,id4,id5 i64 uniform [0,100), "
                    "id3,id6 i64 uniform [0,1e6) seeds 20-25, v f64 seed 5 (ranges multiply to 1e20 > 2^63; ~1e8 groups)", rows=100_000_000,
               bytes_per_row=56, dtype="f64", kernel="k_row_hash<6> + k_group_hash<2> (device-wide table) + k_join_probe_hash + 6 x (k_gather_or, compare): tuple proof at the groups' first rows + k_group_emit_by_group"),
    "x6": dict(desc="expression aggregate (TPC-H Q6 shape): select sum(p * d) where q < 24 and d >= 0.05 and d <= 0.07; p f64 seed 12, d f64 seed 13 "
                    "scaled to [0,0.1), q i64 uniform [0,50) seed 14", rows=1_000_000_000, bytes_per_row=24, dtype="f64", kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<3, 1, 4, 4, 1, false> without hiprtc)"),
    "q1": dict(desc="nested expressions (TPC-H Q1 shape): sum(q), sum(p), sum(p*(1-d)), sum(p*(1-d)*(1+t)), avg(q), avg(p), avg(d), count by {rf, ls} "
                    "where sd <= 2400; rf in [0,3), ls in [0,2), q i64 [1,50], p f64, d f64 [0,.1), t f64 [0,.08), sd i64 [0,2500)", rows=1_000_000_000,
               bytes_per_row=56, dtype="f64", kernel="k_filter_aggr<3, 4, 4, 1, 0, false> (both key scopes) + k_group_few (compiled at run time for the plan: register accumulators per (aggregate, group); prebuilt k_group_dense<7, true, 256, true, 2> without hiprtc)"),
    "k9": dict(desc="sparse keys (range > rows -> the reference's open-addressing path): select sum(v) by k, k = 1000003 * (i64 uniform [0,1e6) seed 4) - 77, "
                    "v f64 seed 5", rows=1_000_000_000, bytes_per_row=16, dtype="f64", kernel="k_plane_scatter<3, 0, 2, 7, 4, true> (hash-partitioned planes) + k_plane_hash_aggregate<true>"),
    "w2": dict(desc="where ids: (where (< a 100000)) on the C2 column -> 1e8 ascending i64 row ids (8 B/row in + 8 B/selected row out)", rows=1_000_000_000,
               bytes_per_row=8.8, dtype="int64", kernel="k_where_once_plan (one pass: ballots -> decoupled look-back -> ids, compiled at run time for the plan; prebuilt k_where_once<1, 1> without hiprtc)"),
    "m2": dict(desc="B8 mask: (< a 100000) materialised as the reference's byte mask (8 B/row in + 1 B/row out)", rows=1_000_000_000, bytes_per_row=9,
               dtype="int64", kernel="k_cmp_mask<1>"),
    # (round 5: the bytes a monotone gather at 10 % MUST move are line-granular -- 1 - 0.9^16 = 81 % of b's 128-byte lines hold a selected row: 6.5 GB of
    #  the 8 GB column + 0.8 GB of ids + 0.8 GB written = 8.1 B per table row; rounds 1-4 priced it at 24 B per SELECTED row = 2.4, which no gather can reach)
    "g2": dict(desc="gather: (at b ids) for the 1e8 ascending ids of w2: 8 B id + 8 B write per id + the 81 % of b's 128-byte lines that hold a selected row", rows=1_000_000_000, bytes_per_row=8.1,
               dtype="f64", kernel="k_gather8"),
    "c5": dict(desc="configs[4]: avg,min,max(d) where a<0.316228 and b>0.683772 and c!=0.25, 4 x f64[2e9] seeds 6-9 (64 GB: 2.5e8 rows per GPU at 8)", rows=2_000_000_000,
               bytes_per_row=32, dtype="f64", kernel="k_filter_aggr_plan (K1 compiled at run time for the plan; prebuilt k_filter_aggr<4, 4, 4, 4, 0, false> without hiprtc)"),
}


class Job:
    """One workload on this rank: device columns + a `step()` that runs the whole query and returns its result."""

    def __init__(self, name, eng, sharded, rows, row0):
        from rayforce_amd import _lib as L
        self.name, self.eng, self.sh, self.rows = name, eng, sharded, rows
        g = eng
        self.key = "k"
        if name in ("c2", "c2_1pct", "c2_50pct"):
            self.t = {"a": g.gen_i64(rows, 2, 1_000_000, row0)}
            self.aggs, self.where = [("sum", "a")], ("<", "a", {"c2": 100_000, "c2_1pct": 10_000, "c2_50pct": 500_000}[name])
        elif name == "c1":
            self.t = {"v": g.gen_f64(rows, 1, row0)}
            self.aggs, self.where = [("sum", "v")], None
        elif name == "c3w":
            self.t = {"k": g.gen_i64(rows, 4, 1_000_000, row0), "v": g.gen_f64(rows, 5, row0), "a": g.gen_i64(rows, 2, 1_000_000, row0)}
            self.aggs, self.where = [("sum", "v")], ("<", "a", 100_000)
        elif name == "x6":
            d = g.gen_f64(rows, 13, row0)
            d.mul_(0.1)  # plumbing: scale the synthetic discount column once, outside every timed region
            self.t = {"p": g.gen_f64(rows, 12, row0), "d": d, "q": g.gen_i64(rows, 14, 50, row0)}
            self.aggs = [("sum", ("*", "p", "d"))]
            self.where = ("and", ("<", "q", 24), (">=", "d", 0.05), ("<=", "d", 0.07))
        elif name == "q1":
            q = g.gen_i64(rows, 73, 50, row0)
            q.add_(1)
            d, t = g.gen_f64(rows, 75, row0), g.gen_f64(rows, 76, row0)
            d.mul_(0.1)
            t.mul_(0.08)  # plumbing: scale the synthetic columns once, outside every timed region
            self.t = {"rf": g.gen_i64(rows, 71, 3, row0), "ls": g.gen_i64(rows, 72, 2, row0), "q": q, "p": g.gen_f64(rows, 74, row0), "d": d, "t": t,
                      "sd": g.gen_i64(rows, 77, 2500, row0)}
            self.aggs = [("sum", "q"), ("sum", "p"), ("sum", ("*", "p", ("-", 1, "d"))), ("sum", ("*", ("*", "p", ("-", 1, "d")), ("+", 1, "t"))),
                         ("avg", "q"), ("avg", "p"), ("avg", "d"), ("count", "q")]
            self.where, self.key = ("<=", "sd", 2400), ["rf", "ls"]
        elif name == "k9":
            k = g.gen_i64(rows, 4, 1_000_000, row0)
            k.mul_(1_000_003).sub_(77)  # plumbing: spread the keys once, outside every timed region
            self.t = {"k": k, "v": g.gen_f64(rows, 5, row0)}
            self.aggs, self.where = [("sum", "v")], None
        elif name == "q2":
            self.t = {"id1": g.gen_i64(rows, 10, 100, row0), "id2": g.gen_i64(rows, 11, 100, row0), "v": g.gen_f64(rows, 5, row0)}
            self.aggs, self.where, self.key = [("sum", "v")], None, ["id1", "id2"]
        elif name == "q7":
            mods = (100, 100, 1_000_000, 100, 100, 1_000_000)
            self.t = {f"id{i + 1}": g.gen_i64(rows, 20 + i, m, row0) for i, m in enumerate(mods)}
            self.t["v"] = g.gen_f64(rows, 5, row0)
            self.aggs, self.where, self.key = [("sum", "v"), ("count", "v")], None, [f"id{i + 1}" for i in range(6)]
        elif name in ("w2", "m2"):
            self.t = {"a": g.gen_i64(rows, 2, 1_000_000, row0)}
            self.aggs, self.where = [], ("<", "a", 100_000)
        elif name == "g2":
            self.t = {"a": g.gen_i64(rows, 2, 1_000_000, row0), "b": g.gen_f64(rows, 3, row0)}
            self.aggs, self.where = [], ("<", "a", 100_000)
            self.ids = g.where(self.where, self.t)
        elif name == "c2b":
            self.t = {"a": g.gen_i64(rows, 2, 1_000_000, row0), "b": g.gen_f64(rows, 3, row0)}
            self.aggs, self.where = [("sum", "b")], ("<", "a", 100_000)
        elif name == "c3":
            self.t = {"k": g.gen_i64(rows, 4, 1_000_000, row0), "v": g.gen_f64(rows, 5, row0)}
            self.aggs, self.where = [("sum", "v")], None
        elif name == "c5":
            self.t = {c: g.gen_f64(rows, s, row0) for c, s in zip("abcd", (6, 7, 8, 9))}
            self.aggs = [("avg", "d"), ("min", "d"), ("max", "d")]
            self.where = ("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25))
        else:
            raise SystemExit(f"unknown workload {name}")
        g.sync()
        self.L = L

    def step(self):
        if self.name == "m2":
            self.eng.timer_start()
            m = self.eng.cmp("<", self.t["a"], 100_000)
            self.kms = self.eng.timer_stop()
            self._mask_keep = m
            return ([int(m.numel())], int(m.numel()))
        if self.name == "g2":
            self.eng.timer_start()
            out = self.eng.at_ids(self.t["b"], self.ids)
            self.kms = self.eng.timer_stop()
            self._out_keep = out
            return ([int(out.numel())], int(out.numel()))
        if self.name == "w2":
            ids = self.sh.where(self.where, self.t) if self.sh is not None else self.eng.where(self.where, self.t)
            self.eng.sync()
            self._ids_keep = ids
            return ([int(ids.numel())], int(ids.numel()))
        if self.name in ("c3", "c3w", "q2", "k9", "q1", "q7"):
            if self.sh is not None:
                return self.sh.group_by(self.key, self.aggs, self.where, self.t)
            return self.eng.group_by(self.key, self.aggs, self.where, self.t)
        if self.sh is not None:
            return self.sh.filter_aggr(self.aggs, self.where, self.t)
        return self.eng.filter_aggr(self.aggs, self.where, self.t, nrows=self.rows)

    def verify(self, res):
        """One UNTIMED check of the timed path's answer against an independent computation (plain torch ops on the same device columns):
        ties the benchmark to the parity tests.  Returns a short description; raises on a mismatch."""
        import torch as T
        t, name = self.t, self.name

        def close(a, b, what):
            a, b = float(a), float(b)
            if not abs(a - b) <= 1e-9 * max(abs(a), abs(b), 1e-300):
                raise SystemExit(f"bench.py: {name}: {what} differs: {a!r} vs torch {b!r}")

        def mask(w):
            if w is None:
                return None
            op = w[0]
            if op in ("and", "or"):
                ms = [mask(x) for x in w[1:]]
                m = ms[0]
                for x in ms[1:]:
                    m = (m & x) if op == "and" else (m | x)
                return m
            col, c = t[w[1]], w[2]
            return {"<": col < c, ">": col > c, "<=": col <= c, ">=": col >= c, "==": col == c, "!=": col != c}[op]

        if name in ("c2", "c2_1pct", "c2_50pct", "c2b", "c1", "c5", "x6"):
            vals, sel = res
            m = mask(self.where)
            nsel = int(m.sum()) if m is not None else self.rows
            if int(sel) != nsel:
                raise SystemExit(f"bench.py: {name}: selected {sel} vs torch {nsel}")
            for (fn, col), got in zip(self.aggs, vals):
                x = t[col] if isinstance(col, str) else (t[col[1]] * t[col[2]])
                x = x if m is None else x[m]
                want = {"sum": x.sum, "avg": lambda: x.mean(), "min": x.min, "max": x.max}[fn]()
                if x.dtype == T.int64 or fn in ("min", "max"):
                    if float(got) != float(want):
                        raise SystemExit(f"bench.py: {name}: {fn} {got!r} vs torch {want!r}")
                else:
                    close(got, want, fn)
            return f"selected rows and {len(vals)} aggregate(s) equal torch's on the same columns"
        if name in ("c3", "c3w", "k9", "q2"):
            m = mask(self.where)
            if name == "q2":
                dense = t["id1"] * 100 + t["id2"]
                size = 10_000
                gk = res["key_columns"][0] * 100 + res["key_columns"][1]
            elif name == "k9":
                dense = (t["k"] + 77) // 1_000_003
                size = 1_000_000
                gk = (res["keys"] + 77) // 1_000_003
            else:
                dense, size, gk = t["k"], 1_000_000, res["keys"]
            v = t["v"]
            if m is not None:
                dense, v = dense[m], v[m]
            want = T.zeros(size, dtype=T.float64, device=v.device).index_add_(0, dense, v)
            seen = T.zeros(size, dtype=T.bool, device=v.device)
            seen[dense] = True
            if int(res["groups"]) != int(seen.sum()):
                raise SystemExit(f"bench.py: {name}: {res['groups']} groups vs torch {int(seen.sum())}")
            got = T.zeros(size, dtype=T.float64, device=v.device)
            got[gk] = res["results"][0]
            scale = T.zeros(size, dtype=T.float64, device=v.device).index_add_(0, dense, v.abs())
            bad = ((got - want).abs() > 1e-9 * scale + 1e-300).sum()
            if int(bad):
                raise SystemExit(f"bench.py: {name}: {int(bad)} group sums differ from torch index_add_ by more than 1e-9 relative")
            first = res["first"]
            if int((first[1:] <= first[:-1]).sum()):
                raise SystemExit(f"bench.py: {name}: groups are not in first-occurrence order")
            return f"{int(res['groups'])} groups: every sum within 1e-9 of torch index_add_, first rows strictly ascending"
        if name == "m2":
            want = (t["a"] < 100_000).to(T.int8)
            if not bool(T.equal(self._mask_keep, want)):
                raise SystemExit("bench.py: m2: the B8 mask differs from torch's comparison")
            return f"{int(want.sum())} of {want.numel()} mask bytes set: equal to torch's (a < 100000)"
        if name == "g2":
            if not bool(T.equal(self._out_keep, t["b"][self.ids])):
                raise SystemExit("bench.py: g2: gathered cells differ from torch's b[ids]")
            return f"{self.ids.numel()} gathered cells equal torch's b[ids]"
        if name == "q1":
            # six groups, eight aggregates: every cell against torch over the group's selected rows (f64 within 1e-9 of sum |x|)
            sel = t["sd"] <= 2400
            kc = res["key_columns"]
            for g in range(int(res["groups"])):
                m = sel & (t["rf"] == int(kc[0][g])) & (t["ls"] == int(kc[1][g]))
                q_, p_, d_, tx = t["q"][m], t["p"][m], t["d"][m], t["t"][m]
                want = [q_.sum(), p_.sum(), (p_ * (1 - d_)).sum(), (p_ * (1 - d_) * (1 + tx)).sum(), q_.double().mean(), p_.mean(), d_.mean(), m.sum()]
                for a, w_ in enumerate(want):
                    got = res["results"][a][g]
                    if got.dtype == T.int64:
                        if int(got) != int(w_):
                            raise SystemExit(f"bench.py: q1: group {g} aggregate {a}: {int(got)} vs torch {int(w_)}")
                    else:
                        close(got, w_, f"group {g} aggregate {a}")
            first = res["first"]
            if int((first[1:] <= first[:-1]).sum()):
                raise SystemExit("bench.py: q1: groups are not in first-occurrence order")
            return f"{int(res['groups'])} groups x 8 aggregates equal torch's over each group's selected rows"
        if name == "q7":
            # ~1e8 groups of (almost) one row: every group's key tuple IS its first row's, counts add up to the rows, sums to the column's sum
            first, kc = res["first"], res["key_columns"]
            if int((first[1:] <= first[:-1]).sum()):
                raise SystemExit("bench.py: q7: groups are not in first-occurrence order")
            for i in range(6):
                if not bool(T.equal(t[f"id{i + 1}"][first], kc[i])):
                    raise SystemExit(f"bench.py: q7: key column {i} differs from the column at the groups' first rows")
            if int(res["results"][1].sum()) != self.rows:
                raise SystemExit("bench.py: q7: group counts do not add up to the rows")
            close(res["results"][0].sum(), t["v"].sum(), "sum over the groups' sums")
            return f"{int(res['groups'])} groups: key tuples equal the first rows', counts add up to the rows, sums to the column's sum"
        if name == "w2":
            ids, n = self._ids_keep, int(mask(self.where).sum())
            ok = ids.numel() == n and bool((ids[1:] > ids[:-1]).all()) and bool((t["a"][ids] < 100_000).all())
            if not ok:
                raise SystemExit("bench.py: w2: ids are not the ascending selected rows")
            return f"{n} ids: ascending, all selected, count equals torch's"
        return None


def timed(job: Job, steps: int, warmup: int, world: int):
    eng = job.eng
    import gc
    gc.collect()  # (as in door_run: no stop-the-world collection of the Python host inside the timed region, none between warm-up and timing either)
    gc_was = gc.isenabled()
    gc.disable()
    for _ in range(warmup):
        job.step()
    eng.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kms = []
    t0 = time.perf_counter()
    res = None
    for _ in range(steps):
        res = job.step()
        kms.append(job.kms if job.name in ("m2", "g2") else eng.last_kernel_ms())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return dt, sum(kms) / max(1, len(kms)), res


def run_workload(name, eng, sharded, rows, row0, steps, warmup, world, total_rows=None):
    """rows = this rank's rows; total_rows = rows of the whole job (all ranks)."""
    w = WORKLOADS[name]
    total_rows = rows * world if total_rows is None else total_rows
    import gc
    from rayforce_amd import _lib as L
    job = Job(name, eng, sharded, rows, row0)
    PATHS = ("plane_scatter", "plane_fallback", "plane_aggregate", "chunk_scatter", "chunk_aggregate", "mask_passes", "where_once")
    XSTATS = {"scopes_sampled": L.RFX_XSTAT_SCOPE_SAMPLED, "sampled_scopes_retried": L.RFX_XSTAT_SCOPE_RETRIED, "hash_tables_grown": L.RFX_XSTAT_HASH_GROWN}
    before, xbefore, rbefore = [eng.stat(i) for i in range(len(PATHS))], {k: eng.xstat(v) for k, v in XSTATS.items()}, rtc_counters(eng)
    dt, kms, res = timed(job, steps, warmup, world)
    paths = dict(zip(PATHS, [eng.stat(i) - b for i, b in enumerate(before)]))
    planner = {k: eng.xstat(v) - xbefore[k] for k, v in XSTATS.items()}
    rtc = {k: v - rbefore[k] for k, v in rtc_counters(eng).items()}
    checked = job.verify(res) if (world == 1 and sharded is None) else None
    ms_step = dt * 1e3 / steps
    if name in ("c3", "c3w", "q2", "k9", "q1", "q7", "w2") or world > 1:
        kms = ms_step  # several dependent kernels (partition, aggregate, rank, emit) / the merge collective: price the whole query
    value = total_rows / (dt / steps)
    alg_bytes = w["bytes_per_row"] * total_rows / world  # per launch, per GPU (SURVEY 8d figures, stated in DESIGN.md)
    achieved = alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    out = dict(workload=name, rows_per_gpu=rows, total_rows=total_rows, ms_per_step=ms_step, rows_per_s=value, kernel_ms=kms, achieved_GBps=achieved,
               frac=achieved / HBM_PEAK_GBPS, result=_brief(res), verified=checked, paths={k: v for k, v in paths.items() if v},
               planner={k: v for k, v in planner.items() if v}, rtc=rtc)
    del job, res
    gc.collect()  # (verify's recursive closures hold the columns in a reference cycle: 240 GB of them by the last workload otherwise)
    eng.trim()
    torch.cuda.empty_cache()
    return out


def rtc_counters(eng):
    """launches that went through run-time compiled plan kernels / plans compiled / loaded from the on-disk code-object cache, so far in this process"""
    rl, rc, rd, rw = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    eng.lib.rfx_hip_rtc_stats(C.byref(rl), C.byref(rc))
    eng.lib.rfx_hip_rtc_cache_stats(C.byref(rd), C.byref(rw))
    return {"launches_through_plan_kernels": int(rl.value), "plans_compiled": int(rc.value), "plans_loaded_from_disk": int(rd.value)}


def kernel_that_ran(label, rtc):
    """WORKLOADS' labels name both forms of a plan kernel; the run's own counters say which one launched."""
    if "prebuilt " not in label or not rtc:
        return label
    plan, _, rest = label.partition(" (")
    prebuilt = rest.split("prebuilt ", 1)[1].split(" without hiprtc")[0]
    if rtc.get("launches_through_plan_kernels", 0) > 0:
        return plan + " (compiled at run time for the plan" + (", loaded from the on-disk code-object cache" if rtc.get("plans_compiled", 0) == 0 else "") + ")"
    return prebuilt + " (prebuilt: no run-time compiled plan kernel launched)"


def roofline_block(name, r, world):
    w = dict(WORKLOADS[name])
    w["kernel"] = kernel_that_ran(w["kernel"], r.get("rtc"))
    return {"bound": "hbm", "achieved": r["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": r["frac"],
            "traffic": pmc_traffic(name) if world == 1 and r["total_rows"] == w["rows"] else None,
            "traffic_source": "profiles/pmc_traffic.json (replayed: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh over this workload, "
                              "not measured in this run)" if world == 1 and r["total_rows"] == w["rows"] and pmc_traffic(name) is not None else None,
            "kernel": w["kernel"], "kernel_ms": r["kernel_ms"],
            "algorithmic_bytes_per_launch": w["bytes_per_row"] * r["total_rows"] / world}


def _brief(res):
    if isinstance(res, dict):
        return {"groups": int(res["groups"])}
    vals, sel = res
    return {"values": vals, "selected": sel}


def pmc_traffic(name):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/), corrected as MI355X_MICROARCH.md
    prescribes; None when that workload has not been profiled."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(name)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ the operator boundary
def boundary_overhead(eng, rows=100_000_000, reps=10):
    """The same query through BOTH doors at one size: Engine.group_by on device-resident columns (what `value` times) and the C operator
    boundary -- rfx_select over host vectors laid out as RayforceDB objects (standalone host object model), columns pinned (rfx_pin:
    uploaded once, trusted until rfx_invalidate) -- result table built on the host included.  Reports ms per query of each and the
    difference: plan walk + cache lookups + result read-back + host table construction."""
    import ctypes as C
    from rayforce_amd import hostobj as H
    ops = H.lib()
    ops.rfx_host_bind()
    cols = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
    q = {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}
    for _ in range(2):
        eng.select({"from": cols, **q})
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.select({"from": cols, **q})
    eng.sync()
    ms_engine = (time.perf_counter() - t0) * 1e3 / reps
    host = {k: v.cpu().numpy() for k, v in cols.items()}
    del cols
    tab = H.table(host)
    t0 = time.perf_counter()
    p = ops.rfx_pin(tab)
    pin_s = time.perf_counter() - t0
    d = H.select_dict(q, tab)
    for _ in range(2):
        ops.rfx_host_drop(ops.rfx_select(d))
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ops.rfx_select(d)
        assert r and not H.is_error(r), H.error_text(r)
        ops.rfx_host_drop(r)
    ms_c = (time.perf_counter() - t0) * 1e3 / reps
    on_gpu = int(ops.rfx_last_select_on_gpu())
    u = ops.rfx_unpin(tab)
    for o in (p, u, d, tab):
        ops.rfx_host_drop(o)
    ops.rfx_cache_clear()
    return {"query": "select sum(v) by k where a < 100000 (c3w shape)", "rows": rows, "engine_ms": ms_engine, "rfx_select_ms": ms_c,
            "boundary_overhead_ms": ms_c - ms_engine, "answered_on_gpu": on_gpu, "pin_upload_s": pin_s,
            "pin_upload_GBps": 3 * rows * 8 / pin_s / 1e9}



# ------------------------------------------------------------------------------------------------ the product's door at full size
C_DOOR = {
    "c3w": ({"k": ("i64", 4, 1_000_000), "v": ("f64", 5), "a": ("i64", 2, 1_000_000)}, {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}),
    "c3": ({"k": ("i64", 4, 1_000_000), "v": ("f64", 5)}, {"by": "k", "s": ("sum", "v")}),
    "c2": ({"a": ("i64", 2, 1_000_000)}, {"where": ("<", "a", 100_000), "s": ("sum", "a")}),
    "c2b": ({"a": ("i64", 2, 1_000_000), "b": ("f64", 3)}, {"where": ("<", "a", 100_000), "s": ("sum", "b")}),
    "c5": ({c: ("f64", sd) for c, sd in zip("abcd", (6, 7, 8, 9))},
           {"where": ("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25)), "x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d")}),
    "q7": ({**{f"id{i + 1}": ("i64", 20 + i, m) for i, m in enumerate((100, 100, 1_000_000, 100, 100, 1_000_000))}, "v": ("f64", 5)},
           {"by": {f"id{i + 1}": f"id{i + 1}" for i in range(6)}, "s": ("sum", "v"), "c": ("count", "v")}),
}


def door_columns(eng, spec, rows, row0=0):
    return {c: (eng.gen_i64(rows, sp[1], sp[2], row0) if sp[0] == "i64" else eng.gen_f64(rows, sp[1], row0)) for c, sp in spec.items()}


def door_run(ops, H, d, steps, warmup, world=1):
    """K timed calls of rfx_select on the dict `d`, each returning the finished HOST result table (read-back and table construction inside the
    timed region); under a launcher bracketed by barriers, the maximum over the ranks."""
    # The timed region belongs to the C library: the Python host around it must not stop the world inside it.  A generation-2 collection of this process
    # (torch imported: ~1e6 tracked objects) takes ~10 ms -- BENCH_r05's ONE 14.2 ms step among twenty 4.7 ms ones (10 % of the mean) has exactly that
    # shape; collected NOW -- BEFORE the warm-up, so that the device does not sit idle (and clock down) between the warm-up and the first timed step: with
    # the collection in between, step 0 read 4.98 ms against 4.70 for the rest -- and switched off until the loop ends.
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    for _ in range(max(1, warmup)):
        r = ops.rfx_select(d)
        assert r and not H.is_error(r), H.error_text(r)
        ops.rfx_host_drop(r)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = None
    marks = [t0]
    for _ in range(steps):
        if r:
            ops.rfx_host_drop(r)
        r = ops.rfx_select(d)
        marks.append(time.perf_counter())  # (rfx_select returns the finished HOST table: every step ends synchronised)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    assert r and not H.is_error(r), H.error_text(r)
    if not int(ops.rfx_last_select_on_gpu()):
        raise SystemExit("bench.py: rfx_select handed the query back instead of answering it on the GPU")
    got = H.table_to_numpy(r)
    ops.rfx_host_drop(r)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    inorder = [(b - a) * 1e3 for a, b in zip(marks, marks[1:])]
    per = sorted(inorder)
    median = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    door_run.last_steps_ms = {"min": per[0], "median": median, "p90": per[min(len(per) - 1, int(0.9 * len(per)))], "max": per[-1],
                              "argmax": inorder.index(per[-1]), "in_order": [round(x, 4) for x in inorder]}
    return dt, got


def door_kernels(ops, H, d, steps=5):
    """HIP-event durations of the bracketed kernels of `steps` more calls (untimed by `value`), on the stream they are launched on
    (rfx_hip_ctx_profile / rfx_hip_profile_kernels on the operator layer's context): per-kernel averages in launch order."""
    from rayforce_amd import _lib as L
    x = C.c_void_p(ops.rfx_ops_exec())
    ctx = C.c_void_p(ops.rfx_exec_ctx(x, 0))
    L.check(ops.rfx_hip_ctx_profile(ctx, 1), "ctx_profile")
    runs, buf, n = [], (C.c_float * 8)(), C.c_int()
    try:
        for _ in range(steps):
            r = ops.rfx_select(d)
            assert r and not H.is_error(r), H.error_text(r)
            ops.rfx_host_drop(r)
            L.check(ops.rfx_hip_profile_kernels(ctx, buf, 8, C.byref(n)), "profile_kernels")
            runs.append([float(buf[i]) for i in range(n.value)])
    finally:
        ops.rfx_hip_ctx_profile(ctx, 0)
    runs = [r for r in runs if len(r) == len(runs[-1])]  # (a first call may run an extra sampled-scope kernel)
    return [sum(col) / len(col) for col in zip(*runs)] if runs else []


def door_clocks(ops, H, d, device_index, steps=10):
    """What the part ran at while answering: shader clock and socket power of THIS device (hwmon: freq1_input, power1_input, power1_cap), polled by a thread
    every millisecond over `steps` more calls, untimed by `value`.  Boxes of a pool differ most where kernels are bound by instruction issue:
    a record that says 0.58 instead of 0.64 should also say what clock it was measured at.  None when the box does not show the files."""
    import glob
    import threading
    try:
        props = torch.cuda.get_device_properties(device_index)
        want = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{getattr(props, 'pci_device_id', 0):02x}"
        hw = None
        for card in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(card)).lower().startswith(want):
                hits = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                hw = hits[0] if hits else None
        if not hw or not os.path.exists(os.path.join(hw, "freq1_input")):
            return None

        def rd(name):
            try:
                with open(os.path.join(hw, name)) as f:
                    return int(f.read().strip())
            except (OSError, ValueError):
                return None
        freq, power, stop = [], [], threading.Event()

        def poll():
            while not stop.is_set():
                a, b = rd("freq1_input"), rd("power1_input")
                if a:
                    freq.append(a / 1e6)
                if b:
                    power.append(b / 1e6)
                time.sleep(0.001)
        idle = rd("freq1_input")
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        for _ in range(steps):
            r = ops.rfx_select(d)
            assert r and not H.is_error(r), H.error_text(r)
            ops.rfx_host_drop(r)
        stop.set()
        th.join(timeout=1.0)
        if not freq:
            return None
        fs, ps = sorted(freq), sorted(power)
        cap = rd("power1_cap")
        return {"sclk_mhz": {"min": round(fs[0]), "median": round(fs[len(fs) // 2]), "max": round(fs[-1])},
                "power_w": {"median": round(ps[len(ps) // 2]), "max": round(ps[-1])} if ps else None, "power_cap_w": round(cap / 1e6) if cap else None,
                "sclk_mhz_before": round(idle / 1e6) if idle else None, "samples": len(fs), "device": want,
                "how": f"hwmon freq1_input / power1_input of the device, one sample per millisecond over {steps} more calls after the timed loop"}
    except Exception as e:  # noqa: BLE001  (a diagnostic: never the reason a bench line is missing)
        log(f"[bench] door_clocks failed: {e}")
        return None


def door_phases(ops, H, d, steps=5):
    """The planner's per-phase wall time (rfx_exec_timing: a sync at every phase end, so the sum sits a little above the untimed step) of
    `steps` more calls, untimed by `value`: ms per step of scope / pass / merge / rank / emit / fetch and the step's total."""
    from rayforce_amd import _lib as L
    x = C.c_void_p(ops.rfx_ops_exec())
    r = ops.rfx_select(d)  # (one unrecorded query: the first one after the host turned the last result into numpy arrays has been seen 20 ms slow)
    assert r and not H.is_error(r), H.error_text(r)
    ops.rfx_host_drop(r)
    ops.rfx_exec_timing(x, 1)
    t0 = time.perf_counter()
    walls = []
    for _ in range(steps):
        t1 = time.perf_counter()
        r = ops.rfx_select(d)
        walls.append((time.perf_counter() - t1) * 1e3)
        assert r and not H.is_error(r), H.error_text(r)
        ops.rfx_host_drop(r)
    wall = (time.perf_counter() - t0) * 1e3 / steps
    out = {k: int(ops.rfx_exec_stat(x, i)) / 1e6 / steps for k, i in L.RFX_XSTAT_PHASES}
    ops.rfx_exec_timing(x, 0)
    out["rfx_select_wall"] = wall  # + the dict walk, the plan and the host table's construction around the planner
    out["rfx_select_wall_min_max"] = [min(walls), max(walls)]
    return out


def door_same(name, got, want, what):
    import numpy as np
    for cname, w in want.items():
        w = w.cpu().numpy() if hasattr(w, "cpu") else np.asarray(w)
        g = got[cname]
        if w.dtype == np.float64:
            okc = g.shape == w.shape and bool(np.all(np.abs(g - w) <= 1e-9 * np.maximum(np.abs(w), 1e-300) + 0.0))
        else:
            okc = g.shape == w.shape and bool(np.array_equal(g, w))
        if not okc:
            raise SystemExit(f"bench.py: rfx_select({name}) column {cname} differs from {what}")


def c_door(name, eng, rows, steps, warmup, device_columns=False):
    """The workload through rfx_select -- the C operator boundary the reference's evaluator (or `loadfn`) binds -- at the workload's full size.
    Default: host vectors laid out as RayforceDB objects, pinned (uploaded once, trusted until rfx_invalidate); device_columns: the columns
    stay where the device generator left them and the table's columns are device handles.  The answer of the last call is checked against
    Engine's on the same data (itself checked against torch in run_workload)."""
    from rayforce_amd import hostobj as H
    ops = H.lib()
    ops.rfx_host_bind()
    spec, q = C_DOOR[name]
    pin_s, pin, keep = None, None, None
    if device_columns:
        keep = door_columns(eng, spec, rows)
        eng.sync()
        tab = H.device_table(keep)
    else:
        host = {}
        for cname in spec:
            dcol = door_columns(eng, {cname: spec[cname]}, rows)[cname]
            host[cname] = dcol.cpu().numpy()
            del dcol
        torch.cuda.empty_cache()
        tab = H.table(host)
        del host
        t0 = time.perf_counter()
        pin = ops.rfx_pin(tab)
        pin_s = time.perf_counter() - t0
    d = H.select_dict(q, tab)
    dt, got = door_run(ops, H, d, steps, warmup)
    steps_ms = dict(door_run.last_steps_ms)
    clocks = door_clocks(ops, H, d, eng.device.index) if "by" in q and not device_columns else None  # (right after the timed loop: the state `value` was measured in)
    phases = door_phases(ops, H, d) if "by" in q else None
    try:
        kernels_ms = door_kernels(ops, H, d)
    except Exception as e:  # noqa: BLE001
        log(f"[bench] door_kernels failed: {e}")
        kernels_ms = []
    cols = keep if device_columns else door_columns(eng, spec, rows)
    want = eng.select({"from": cols, **q})
    door_same(name, got, want, "Engine.select on the same data")
    del cols, want, keep
    det = None
    if "by" in q and not device_columns and os.environ.get("RFX_DETERMINISTIC") is None and os.environ.get("RFX_VALIDATE") is None:
        # the same query in the reproducible modes (grouped f64 sums as integer sums over the column's fixed-point image, one limb or two: DESIGN.md section 4):
        # the first call makes the image(s) (kept with the resident copy), the later ones run on them
        import numpy as np
        det = {}
        for mode, label in ((1, "one_limb"), (2, "two_limbs")):
            st0 = H.to_numpy(ops.rfx_stats(0))
            assert ops.rfx_ops_set_deterministic(mode) == 0
            try:
                t0 = time.perf_counter()
                r = ops.rfx_select(d)
                first_ms = (time.perf_counter() - t0) * 1e3
                assert r and not H.is_error(r), H.error_text(r)
                ops.rfx_host_drop(r)
                ddt, dgot = door_run(ops, H, d, min(steps, 10), 2)
                dsteps = dict(door_run.last_steps_ms)
                r = ops.rfx_select(d)
                again = H.table_to_numpy(r)
                ops.rfx_host_drop(r)
                same_bits = all(np.array_equal(np.ascontiguousarray(dgot[c]).view(np.uint8), np.ascontiguousarray(again[c]).view(np.uint8)) for c in dgot)
                err = max([float(np.max(np.abs(dgot[c] - got[c]) / np.maximum(np.abs(got[c]), 1e-300))) for c in got if got[c].dtype == np.float64] + [0.0])
                exact = all(bool(np.array_equal(dgot[c], got[c])) for c in got if got[c].dtype != np.float64)
                # (one limb rounds a cell to 2^(e + b - 62) ABSOLUTE: at the workload's full size that is far inside 1e-9 of every group's sum; a --rows run
                #  with a handful of rows per group may hold a group of tiny values beyond it -- reported, not judged; two limbs are held to 1e-9 at any size)
                judged = mode == 2 or rows >= 500_000_000
                if not (same_bits and exact and (err <= 1e-9 or not judged)):
                    raise SystemExit(f"bench.py: rfx_select({name}) in reproducible mode {mode}: bit-identical between two calls {same_bits}, "
                                     f"largest relative distance from the default path {err:.3g}")
                st = H.to_numpy(ops.rfx_stats(0))
                try:
                    dk = door_kernels(ops, H, d)
                except Exception as e:  # noqa: BLE001
                    log(f"[bench] door_kernels (reproducible mode) failed: {e}")
                    dk = []
                det[label] = {"first_call_ms": first_ms, "kernels_ms": dk, "phases_ms": door_phases(ops, H, d),
                              "median_ms": dsteps["median"], "ms_per_step": ddt * 1e3 / min(steps, 10), "steps_ms_in_order": dsteps["in_order"],
                              "images_made": int(st[15] - st0[15]), "images_found_again": int(st[16] - st0[16]), "largest_relative_distance_from_the_default_path": err,
                              "verified": "two calls bit-identical" + ("; every column within 1e-9 of the default path's" if judged else "")}
            finally:
                ops.rfx_ops_set_deterministic(0)
    for o in ([pin, ops.rfx_unpin(tab)] if pin else []) + [d, tab]:
        ops.rfx_host_drop(o)
    ops.rfx_cache_clear()
    torch.cuda.empty_cache()
    return {"door": "rfx_select (C operator boundary, include/rfx_ops.h) on " + ("device column handles" if device_columns else "pinned host columns") +
                    "; result table built on the host inside the timed region",
            "rows": rows, "steps": steps, "ms_per_step": dt * 1e3 / steps, "rows_per_s": rows / (dt / steps), "answered_on_gpu": 1,
            "median_ms": steps_ms["median"], "rows_per_s_median": rows / (steps_ms["median"] * 1e-3),
            "steps_ms": steps_ms, "phases_ms": phases, "kernels_ms": kernels_ms,
            "pin_upload_s": pin_s, "clocks": clocks, "deterministic_mode": det, "verified": "every result column equals Engine.select's on the same data (f64 within 1e-9)"}


# ------------------------------------------------------------------------------------------------ the Amdahl budget of the sharded tail
XGMI_LINK_GBPS = 76.8   # one xGMI link, one direction (7 links x 153.6 GB/s bidirectional per GPU)
XGMI_EFFICIENCY = 0.6   # what a fused RCCL exchange is assumed to reach of it (NOT measured: the builder's boxes have one GPU)
XGMI_LATENCY_MS = 0.04  # launch + synchronisation of one grouped exchange (the one-rank RCCL world on one GPU measures 0.03-0.05)


def predicted_scaling(name, eng, steps=5):
    """T(N) of one evaluator process over N devices (strong scaling: the SAME table split N ways), from MEASUREMENTS on this one device plus
    a stated model of the one term that needs a second device.  Per N: the query over rows / N through the door with the planner's phase
    timers (scope + pass = what every device does in parallel; rank = every device ranks the merged tables, redundantly; emit and fetch =
    the whole result here, 1 / N of it per device once sliced: rfx_exec_groups_fetch_all), the planner's per-phase hand-over to N - 1 worker
    threads measured with N shards on this device (RFX_SHARDS), and the merge modelled as a reduce-scatter + all-gather of the table bytes
    over N - 1 xGMI links at XGMI_EFFICIENCY."""
    from rayforce_amd import hostobj as H
    ops = H.lib()
    ops.rfx_host_bind()
    spec, q = C_DOOR[name]
    total = WORKLOADS[name]["rows"]
    out = {"model": "T(N) = scope+pass(rows/N, measured) + merge(N, MODELLED: 2 * table_bytes/N / (%.1f GB/s * %.1f) + %.2f ms) + rank(measured, every device ranks) "
                    "+ (emit + fetch)(measured)/N + host(measured: rfx_select_wall - planner total) + 6 phase hand-overs to N - 1 threads (measured: rfx_exec_probe_handover_us)" % (XGMI_LINK_GBPS, XGMI_EFFICIENCY, XGMI_LATENCY_MS),
           "strong_scaling_rows": total, "per_n": {}}
    base = None
    for n in (1, 2, 4, 8):
        rows = total // n
        cols = door_columns(eng, spec, rows)
        eng.sync()
        tab = H.device_table(cols)
        d = H.select_dict(q, tab)
        for _ in range(2):
            r = ops.rfx_select(d)
            assert r and not H.is_error(r), H.error_text(r)
            groups = len(next(iter(H.table_to_numpy(r).values())))
            ops.rfx_host_drop(r)
        if "by" in q:
            ph = door_phases(ops, H, d, steps)
        else:  # a scalar fold has no planner phases: the whole call, timed from outside (median), is the device's part
            walls = []
            for _ in range(max(steps, 5)):
                t1 = time.perf_counter()
                r = ops.rfx_select(d)
                walls.append((time.perf_counter() - t1) * 1e3)
                assert r and not H.is_error(r), H.error_text(r)
                ops.rfx_host_drop(r)
            wall = sorted(walls)[len(walls) // 2]
            ph = {"scope": 0.0, "pass": wall, "merge": 0.0, "rank": 0.0, "emit": 0.0, "fetch": 0.0, "total": wall, "rfx_select_wall": wall}
        for o in (d, tab):
            ops.rfx_host_drop(o)
        del cols
        ops.rfx_cache_clear()
        torch.cuda.empty_cache()
        narr = 2  # first rows + one accumulator array (sum of f64) per slot
        table_bytes = narr * 1_000_000 * 8 if "by" in q else 64 * len(q)
        merge = 0.0 if n == 1 else 2 * (table_bytes / n) / (XGMI_LINK_GBPS * 1e9 * XGMI_EFFICIENCY) * 1e3 + XGMI_LATENCY_MS
        host = max(0.0, ph["rfx_select_wall"] - ph["total"])
        # six phases of a sharded group-by hand the work to N - 1 worker threads (scope sample, scope / partition, pass, merge wait, rank + emit, fetch)
        handover = 0.0 if n == 1 else (6 if "by" in q else 2) * float(ops.rfx_exec_probe_handover_us(n, 2000)) / 1e3  # (a scalar fold: the pass and the gather of the partials)
        t = ph["scope"] + ph["pass"] + merge + ph["rank"] + (ph["emit"] + ph["fetch"]) / n + host + handover
        if n == 1:
            base = t
        out["per_n"][str(n)] = {"rows_per_device": rows, "groups": groups, "measured_ms": {k: round(v, 4) for k, v in ph.items() if not isinstance(v, list)}, "merge_model_ms": round(merge, 4),
                                "host_ms": round(host, 4), "handover_ms": round(handover, 4), "T_ms": round(t, 4), "speedup": round(base / t, 3)}
    return out


def door_property_check(name, got, cols_by_shard, q, reduce_sum, sliced=False):
    """N > 1: no unsharded copy exists to compare with -- the answer's invariants instead.  Scalar: selected sums / extrema from torch per shard,
    folded; grouped: the groups' sums add up to the selected rows' sum, the group count is the number of distinct selected keys (torch.unique
    per shard, folded by a presence table), first-occurrence order = every key new where it stands (no duplicates)."""
    import numpy as np
    where = q.get("where")

    def mask(t, w):
        if w[0] == "and":
            m = mask(t, w[1])
            for x in w[2:]:
                m = m & mask(t, x)
            return m
        col, c = t[w[1]], w[2]
        return {"<": col < c, ">": col > c, "!=": col != c}[w[0]]

    if "by" in q:
        loc_sum, pres = 0.0, None
        for t in cols_by_shard:
            with torch.cuda.device(t["v"].device):
                m = mask(t, where) if where else None
                v = t["v"][m] if m is not None else t["v"]
                k = t["k"][m] if m is not None else t["k"]
                loc_sum += float(v.sum())
                p = torch.zeros(1_000_000, dtype=torch.int64, device=k.device)
                p[k] = 1
                pres = p.cpu() if pres is None else torch.maximum(pres, p.cpu())
        tot = reduce_sum(loc_sum)
        pres = reduce_sum(pres.double()).clamp(max=1.0)
        groups = int(pres.sum())
        if sliced:  # `got` is this rank's range of the groups: the ranks' pieces together are every distinct key exactly once, their sums add up
            mine = torch.zeros(1_000_000, dtype=torch.float64)
            mine[torch.from_numpy(np.ascontiguousarray(got["k"]))] += 1.0
            seen = reduce_sum(mine)
            n_all, s_all = int(reduce_sum(float(len(got["k"])))), reduce_sum(float(got["s"].sum()))
            if n_all != groups or len(np.unique(got["k"])) != len(got["k"]) or float(seen.max()) > 1.0 or int(seen.sum()) != groups or not torch.equal(seen > 0, pres > 0):
                raise SystemExit(f"bench.py: {name}: the ranks' slices hold {n_all} groups ({int((seen > 1).sum())} keys twice) vs {groups} distinct selected keys")
            if not abs(s_all - tot) <= 1e-9 * abs(tot):
                raise SystemExit(f"bench.py: {name}: the slices' sums add up to {s_all!r}, the selected rows' to {tot!r}")
            return f"{groups} groups over the ranks' slices = the distinct selected keys, each exactly once, the groups' sums add up to the selected rows' sum (1e-9)"
        if len(got["k"]) != groups or len(np.unique(got["k"])) != groups:
            raise SystemExit(f"bench.py: {name}: {len(got['k'])} groups vs {groups} distinct selected keys")
        if not abs(float(got["s"].sum()) - tot) <= 1e-9 * abs(tot):
            raise SystemExit(f"bench.py: {name}: the groups' sums add up to {float(got['s'].sum())!r}, the selected rows' to {tot!r}")
        return f"{groups} groups = distinct selected keys, no key twice, the groups' sums add up to the selected rows' sum (1e-9)"
    sums = 0.0
    for t in cols_by_shard:
        with torch.cuda.device(next(iter(t.values())).device):
            m = mask(t, where)
            col = t[q[next(k for k in q if k != "where")][1]]
            sums += float(col[m].sum())
    tot = reduce_sum(sums)
    first = next(k for k in q if k != "where")
    if q[first][0] == "sum" and not abs(float(got[first][0]) - tot) <= 1e-9 * abs(tot):
        raise SystemExit(f"bench.py: {name}: sum {float(got[first][0])!r} vs torch over the shards {tot!r}")
    return "the aggregate equals torch's over the shards' selected rows (1e-9)" if q[first][0] == "sum" else "answered (avg / min / max: see tests/test_dist_multi_gpu.py)"


def one_process(args):
    """`python bench.py --gpus N` without a launcher: ONE process drives N devices through the sharded operator layer (rfx_ops_set_shards): the
    evaluator process of INTEGRATION.md.  RFX_BENCH_SAME_DEVICE=1 puts the N shards on device 0 (merged by the planner's kernel instead of RCCL):
    the same code on a one-GPU box."""
    _stdout_is_for_the_json_line_only()
    from rayforce_amd import hostobj as H, _lib as L
    from rayforce_amd.engine import Engine
    name, N = args.workload, args.gpus
    if name not in C_DOOR or name == "q7":
        raise SystemExit(f"bench.py: --gpus {N} in one process runs the BASELINE configs ({', '.join(k for k in C_DOOR if k != 'q7')})")
    same = bool(os.environ.get("RFX_BENCH_SAME_DEVICE"))
    if not same and torch.cuda.device_count() < N:
        raise SystemExit(f"bench.py: --gpus {N} but this box has {torch.cuda.device_count()} device(s)")
    devs = [0] * N if same else list(range(N))
    ops = H.lib()
    ops.rfx_host_bind()
    uniq = sorted(set(devs))
    L.check(ops.rfx_ops_set_shards((C.c_int * len(uniq))(*uniq), len(uniq), N), "ops_set_shards")
    spec, q = C_DOOR[name]
    base = args.rows or WORKLOADS[name]["rows"]
    total = base if args.scaling == "strong" else base * N
    engs, shards = {}, []
    r0, ln = C.c_int64(), C.c_int64()
    for s_ in range(N):
        e = engs.setdefault(devs[s_], Engine(devs[s_]))
        ops.rfx_exec_split(total, N, s_, C.byref(r0), C.byref(ln))
        shards.append(door_columns(e, spec, ln.value, r0.value))
        e.sync()
    cols = [H.device_vector(shards[0][c], ptrs=[sh[c].data_ptr() for sh in shards]) for c in spec]
    for cv, c in zip(cols, spec):  # the handle's length is the whole column's
        H.header(cv).len = total
    tab = ops.rfx_host_table(H.symbols(list(spec)), H.list_of(cols))
    d = H.select_dict(q, tab)
    dt, got = door_run(ops, H, d, args.steps, args.warmup)
    steps_ms = dict(door_run.last_steps_ms)
    phases = door_phases(ops, H, d) if "by" in q else None
    checked = door_property_check(name, got, shards, q, lambda x: x)
    x = C.c_void_p(ops.rfx_ops_exec())
    w = WORKLOADS[name]
    ms = dt * 1e3 / args.steps
    achieved = w["bytes_per_row"] * total / N / (ms * 1e-3) / 1e9
    # did RCCL see N ranks?  ncclCommCount of every shard's lead communicator (rfx_exec_comm_init_all -> ncclCommInitAll over the devices)
    seen = []
    w_, r_ = C.c_int(), C.c_int()
    for s_ in range(N):
        L.check(ops.rfx_dist_world(C.c_void_p(ops.rfx_exec_ctx(x, s_)), C.byref(w_), C.byref(r_)), "dist_world")
        seen.append([int(w_.value), int(r_.value)])
    ranks_seen = seen[0][0]
    fused = int(ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_RCCL))
    if len(uniq) > 1:  # several devices: the merge MUST have been the fused RCCL exchange among exactly that many ranks -- anything else is not an N-GPU run
        if fused == 0:
            raise SystemExit(f"bench.py: --gpus {N} over {len(uniq)} devices but the planner ran 0 fused RCCL exchanges (merges by kernel: "
                             f"{int(ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_KERNEL))}) -- refusing to report this as an {N}-GPU run")
        if ranks_seen != len(uniq) or sorted(r for _, r in seen) != list(range(len(uniq))):
            raise SystemExit(f"bench.py: the communicators of the {len(uniq)} devices report (count, rank) = {seen}")
    cpu = None
    if not args.no_cpu_baseline:  # the same bounded CPU sample as at N = 1 (the reference's CPU path does not change with the GPU count)
        try:
            cpu = cpu_baseline(name, min(args.cpu_sample_rows, base, 20_000_000), timeout=45, pools=(32,), sweep_budget_s=20)  # (a smaller sample than N = 1's: N > 1 runs follow back to back)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cpu_baseline failed: {e}")
    print(json.dumps({
        "metric": METRIC, "value": total / (dt / args.steps), "unit": "rows/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": w["dtype"], "data": "synthetic",
        "config": {"workload": f"{name}: {w['desc']}", "total_rows": total, "rows_per_gpu": total // N,
                   "sharding": f"ONE process, {N} shards on {len(uniq)} device(s): every shard's pass on its own host thread, partial tables merged by "
                               + ("the planner's device kernel" if len(uniq) == 1 else "one fused RCCL exchange over xGMI"),
                   "door": "rfx_select on per-shard device column handles (rfx_ops_set_shards); result table built on the host inside the timed region",
                   "verified": checked, "resident": "HBM (columns generated on every device)",
                   "planner": {"merges_by_kernel": int(ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_KERNEL)), "fused_rccl_exchanges": int(ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_RCCL)),
                               "sliced_results": int(ops.rfx_exec_stat(x, L.RFX_XSTAT_SLICED))},
                   "ranks_seen": ranks_seen, "communicators": seen},
        "median_ms": steps_ms["median"], "value_median": total / (steps_ms["median"] * 1e-3), "value_basis": "mean step of the timed loop (N > 1: the launch contract's max-over-ranks form); value_median beside it",
        "steps_ms": steps_ms, "phases_ms": phases,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": w["kernel"],
                     "kernel_ms": ms, "algorithmic_bytes_per_launch": w["bytes_per_row"] * total / N},
        "cpu_baseline": cpu}), flush=True)
    for e in engs.values():
        e.close()


def launcher_door(args, name, eng, world, rank, rows, row0, total_rows):
    """Under a launcher, world > 1: this rank's row range as device columns through rfx_select with the operator layer's context in the
    RCCL communicator (rfx_ops_dist_init): a scalar answer whole on every rank, a grouped one sliced over the ranks (every rank its range of the groups)."""
    from rayforce_amd import hostobj as H, _lib as L
    ops = H.lib()
    ops.rfx_host_bind()
    L.check(ops.rfx_ops_set_device(eng.device.index), "ops_set_device")
    transport = None
    if world > 1 and dist.get_backend() == "gloo":
        # test plumbing (RFX_BENCH_BACKEND=gloo RFX_BENCH_SAME_DEVICE=1: N ranks on ONE GPU, where RCCL refuses a communicator): the planner's
        # exchanges ride torch.distributed through the rfx_transport_t hooks -- the launch, the door and the record are the N > 1 code
        from rayforce_amd.dist import _TorchTransport
        import numpy as np
        v0 = H.vector(np.arange(4, dtype=np.int64))  # (any operator call brings the layer's context and planner up)
        r0 = ops.rfx_sum(v0)
        ops.rfx_host_drop(r0)
        ops.rfx_host_drop(v0)

        class _OpsCtx:
            lib = ops
            _ctx = C.c_void_p(ops.rfx_exec_ctx(C.c_void_p(ops.rfx_ops_exec()), 0))
        transport = _TorchTransport(_OpsCtx, None)
        L.check(ops.rfx_exec_set_transport(C.c_void_p(ops.rfx_ops_exec()), C.byref(transport.struct)), "exec_set_transport")
    else:
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            L.check(ops.rfx_dist_unique_id(buf), "dist_unique_id")
            ident = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        L.check(ops.rfx_ops_dist_init(world, rank, C.c_char_p(ident[0])), "ops_dist_init")
    spec, q = C_DOOR[name]
    # a grouped answer stays SLICED over the ranks: every rank reads back and returns only its range of the groups (rfx_ops_set_rank_slices; the ranks' tables
    # end to end are the answer) -- the whole 16 MB result on every rank is a constant that does not shrink with the ranks.  RFX_BENCH_WHOLE_RESULT=1: as before
    sliced = "by" in q and not os.environ.get("RFX_BENCH_WHOLE_RESULT")
    L.check(ops.rfx_ops_set_rank_slices(1 if sliced else 0), "ops_set_rank_slices")
    cols = door_columns(eng, spec, rows, row0)
    eng.sync()
    tab = H.device_table(cols)
    d = H.select_dict(q, tab)
    dt, got = door_run(ops, H, d, args.steps, args.warmup, world)

    def reduce_sum(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda") if isinstance(x, float) else x.cuda()
        if world > 1:
            dist.all_reduce(t)
        return float(t[0]) if isinstance(x, float) else t.cpu()

    checked = door_property_check(name, got, [cols], q, reduce_sum, sliced=sliced and world > 1)
    x = C.c_void_p(ops.rfx_ops_exec())
    w_, r_ = C.c_int(), C.c_int()
    if transport is None:
        ops.rfx_dist_world(C.c_void_p(ops.rfx_exec_ctx(x, 0)), C.byref(w_), C.byref(r_))
        if int(w_.value) != world:
            raise SystemExit(f"bench.py: the RCCL communicator spans {int(w_.value)} ranks, the launcher started {world}")
    calls = transport.calls if transport else int(ops.rfx_dist_calls(C.c_void_p(ops.rfx_exec_ctx(x, 0))))
    import hashlib
    import numpy as np
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(got[c]).tobytes() for c in sorted(got))).hexdigest()  # (bit-stable across rank counts in the reproducible mode only)
    out = {"ms_per_step": dt * 1e3 / args.steps, "rows_per_s": total_rows / (dt / args.steps), "verified": checked, "ranks_seen": transport.world if transport else int(w_.value),
           "result_digest": digest, "reproducible_mode": os.environ.get("RFX_DETERMINISTIC") == "1",
           "collectives_per_query": calls / (args.steps + max(1, args.warmup)),
           "result": ({"groups": int(reduce_sum(float(len(got[next(iter(got))])))) if sliced and world > 1 else len(got[next(iter(got))]),
                       "returned": "every rank its range of the groups" if sliced and world > 1 else "the whole answer on every rank"}
                      if "by" in q else {"values": [float(v[0]) for v in got.values()]})}
    ops.rfx_ops_set_rank_slices(0)
    if transport:
        ops.rfx_exec_set_transport(x, None)
        out["exchange"] = "torch.distributed (gloo) through rfx_transport_t: test plumbing, not a data path"
    else:
        ops.rfx_ops_dist_finalize()
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(name, sample_rows, timeout=120, pools=(128, 64, 32), sweep_budget_s=60):
    """The reference itself (oracle/_ref/rayforce, kind 'reference') or -- when it is not built -- the C restatement
    (kind 'port'), timed on this box's host cores over a bounded sample of the same workload."""
    import numpy as np
    from oracle import ref, rfo
    cores = os.cpu_count() or 1
    rfo.set_threads(cores)
    t0 = time.perf_counter()
    if name == "c2":
        cols = {"a": rfo.gen_i64(sample_rows, 2, 1_000_000)}
        q = "(select {s: (sum a) from: t where: (< a 100000)})"
        oq = {"where": ("<", "a", 100_000), "s": ("sum", "a")}
    elif name == "c1":
        cols = {"v": rfo.gen_f64(sample_rows, 1)}
        q = "(sum v)"
        oq = {"s": ("sum", "v")}
    elif name == "c3w":
        cols = {"k": rfo.gen_i64(sample_rows, 4, 1_000_000), "v": rfo.gen_f64(sample_rows, 5), "a": rfo.gen_i64(sample_rows, 2, 1_000_000)}
        q = "(select {s: (sum v) from: t where: (< a 100000) by: k})"
        oq = {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}
    elif name == "x6":
        cols = {"p": rfo.gen_f64(sample_rows, 12), "d": rfo.gen_f64(sample_rows, 13) * 0.1, "q": rfo.gen_i64(sample_rows, 14, 50)}
        q = "(select {s: (sum (* p d)) from: t where: (and (< q 24) (>= d 0.05) (<= d 0.07))})"
        oq = {"where": ("and", ("<", "q", 24), (">=", "d", 0.05), ("<=", "d", 0.07)), "s": ("sum", ("*", "p", "d"))}
    elif name == "k9":
        cols = {"k": rfo.gen_i64(sample_rows, 4, 1_000_000) * 1_000_003 - 77, "v": rfo.gen_f64(sample_rows, 5)}
        q = "(select {s: (sum v) from: t by: k})"
        oq = {"by": "k", "s": ("sum", "v")}
    elif name == "q2":
        cols = {"id1": rfo.gen_i64(sample_rows, 10, 100), "id2": rfo.gen_i64(sample_rows, 11, 100), "v": rfo.gen_f64(sample_rows, 5)}
        q = "(select {s: (sum v) from: t by: {id1: id1 id2: id2}})"
        oq = {"by": {"id1": "id1", "id2": "id2"}, "s": ("sum", "v")}
    elif name == "c2b":
        cols = {"a": rfo.gen_i64(sample_rows, 2, 1_000_000), "b": rfo.gen_f64(sample_rows, 3)}
        q = "(select {s: (sum b) from: t where: (< a 100000)})"
        oq = {"where": ("<", "a", 100_000), "s": ("sum", "b")}
    elif name == "c3":
        cols = {"k": rfo.gen_i64(sample_rows, 4, 1_000_000), "v": rfo.gen_f64(sample_rows, 5)}
        q = "(select {s: (sum v) from: t by: k})"
        oq = {"by": "k", "s": ("sum", "v")}
    else:
        cols = {c: rfo.gen_f64(sample_rows, s) for c, s in zip("abcd", (6, 7, 8, 9))}
        q = "(select {x: (avg d) y: (min d) z: (max d) from: t where: (and (< a 0.316228) (> b 0.683772) (!= c 0.25))})"
        oq = {"where": ("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25)), "x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d")}
    log(f"[cpu_baseline] generated {sample_rows} sample rows in {time.perf_counter() - t0:.1f}s")
    reps = 5
    if ref.available():
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
        # The reference sizes its pool to ALL cores (core/runtime.c:141-145) -- its own policy, reported as `all_cores` -- but on a 256-thread
        # box that pool often HURTS it (a 1e7-row sum: 11 ms on 256 threads, 1-2 ms on 8); so the sweep -c {all, 128, 64, 32} and `value` =
        # the BEST of them (the fairest CPU figure to stand beside the GPU's).  Its page-aligned chunking can also overshoot and crash
        # when the pool is large relative to the input, and it occasionally hangs with very wide pools: every attempt is bounded.
        runs = []
        try:
            with ref.Session(root=shm) as s:
                # materialise the mmapped column files into heap vectors first (the GPU path is timed HBM-resident too)
                for k, v in cols.items():
                    s.put(k, v)
                    s.eval(f"(set {k} (+ {k} 0))" if v.dtype == np.int64 else f"(set {k} (+ {k} 0.0))")
                names = " ".join(cols.keys())
                s.eval(f"(set t (table [{names}] (list {names})))")
                s.eval(f"(set warm {q})")
                s.out("ms", f"(enlist (timeit {reps} {q}))")
                t_sweep = time.perf_counter()
                for threads in list(dict.fromkeys([cores] + [t for t in pools if t < cores])):
                    if runs and time.perf_counter() - t_sweep > sweep_budget_s:
                        break
                    try:
                        out = s.run(threads=threads, timeout=timeout)
                        ms = float(out["ms"][0]) / reps
                        if ms > 0:
                            runs.append({"cores": threads, "ms_per_query": ms, "value": sample_rows / (ms * 1e-3)})
                    except Exception as e:  # noqa: BLE001
                        log(f"[cpu_baseline] reference run with {threads} threads failed ({str(e)[:120]})")
        except Exception as e:  # noqa: BLE001
            log(f"[cpu_baseline] reference session failed ({str(e)[:160]})")
        if runs:
            best = max(runs, key=lambda r: r["value"])
            allc = next((r for r in runs if r["cores"] == cores), None)
            return dict(value=best["value"], unit="rows/s", cores=best["cores"], kind="reference", ms_per_query=best["ms_per_query"],
                        all_cores=allc, sweep=runs,
                        sample=f"{name}: first {sample_rows} rows of the workload (same seeds), real RayforceDB build (oracle/_ref, gcc -O3 x86-64-v3); "
                               f"(timeit {reps} query) after one warm run at -c {[r['cores'] for r in runs]} of {cores} hardware threads; value = the best pool "
                               f"(-c {best['cores']}), all_cores = the reference's own default policy")
        log("[cpu_baseline] reference unusable here, falling back to the port")
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rfo.select({"from": cols, **oq})
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return dict(value=sample_rows / best, unit="rows/s", cores=cores, kind="port", ms_per_query=best * 1e3,
                sample=f"{name}: first {sample_rows} rows of the workload (same seeds), C restatement oracle/librfo.so with {cores} OpenMP threads, "
                       f"best of {reps}")


def shard_range(total, world, rank):
    """Row range [lo, hi) of `rank`: contiguous, sizes differ by at most one row."""
    return total * rank // world, total * (rank + 1) // world


def respawn(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one per GPU) under torch.distributed.run on 127.0.0.1."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:10])} ...")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3w", choices=list(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the workload's BASELINE row count in TOTAL, split over the ranks (SURVEY 8d C4/C5); weak: that many rows PER rank")
    ap.add_argument("--rows", type=int, default=0, help="override the workload's BASELINE row count (total under strong scaling, per rank under weak)")
    ap.add_argument("--cpu-sample-rows", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="use the row-sharded driver (collectives) even with one rank")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads in the 'also' field")
    ap.add_argument("--no-predict", action="store_true", help="skip the predicted_scaling block (the query over rows / N on this device for N = 1, 2, 4, 8 + the stated merge model)")
    ap.add_argument("--engine-door", action="store_true", help="report Engine.group_by / filter_aggr (ctypes host, device-resident results) as `value` instead of rfx_select")
    ap.add_argument("--spawn", action="store_true", help="--gpus N without a launcher: start N ranks under torch.distributed.run (one process per GPU) instead of one process over N devices")
    ap.add_argument("--dry-run", action="store_true", help="no device work: rendezvous, shard arithmetic and the JSON line only (CPU test of the launch contract)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--ab", default="", help="dev: comma list of tune flags to A/B in ONE process (same box, same clocks); prints one line per run")
    ap.add_argument("--tune-flags", type=int, default=0, help="rfx_hip_ctx_tune flags (kernel-variant experiments)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.spawn or args.dry_run:
            respawn(args.gpus)  # does not return
        return one_process(args)  # ONE process over the N devices: the sharded operator layer
    _stdout_is_for_the_json_line_only()  # (after the respawn decision: the ranks started above inherit the untouched stdout)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks -- refusing to report a {world}-GPU run as {args.gpus} GPUs")
    name = args.workload
    base_rows = args.rows or WORKLOADS[name]["rows"]
    total_rows = base_rows if args.scaling == "strong" else base_rows * world
    lo, hi = shard_range(total_rows, world, rank)
    rows, row0 = hi - lo, lo

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if args.dry_run else os.environ.get("RFX_BENCH_BACKEND", "nccl")  # "gloo" + RFX_BENCH_SAME_DEVICE=1: N ranks on ONE GPU
        if os.environ.get("RFX_BENCH_SAME_DEVICE"):
            local_rank = 0
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if not args.dry_run:
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    if args.dry_run:
        # the launch contract without a device: every rank reports its shard, rank 0 checks the shards tile [0, total) and prints the line
        shards = [None] * world
        if world > 1:
            dist.all_gather_object(shards, (row0, rows))
            dist.barrier()
        else:
            shards = [(row0, rows)]
        if rank == 0:
            pos = 0
            for r0, n in shards:
                assert r0 == pos, (shards, "shards must tile the table")
                pos += n
            assert pos == total_rows
            w = WORKLOADS[name]
            print(json.dumps({"metric": METRIC, "value": None, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": w["dtype"], "data": "synthetic", "dry_run": True,
                              "config": {"workload": f"{name}: {w['desc']}", "total_rows": total_rows, "rows_per_gpu": [n for _, n in shards],
                                         "sharding": f"row-range x{world}" if world > 1 else "single GPU"}}), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")

    from rayforce_amd.engine import Engine
    from rayforce_amd.dist import ShardedEngine
    eng = Engine(local_rank)
    if args.blocks_per_cu or args.tune_flags:
        eng.tune(blocks_per_cu=args.blocks_per_cu, flags=args.tune_flags)
    if args.sharded and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    sharded = ShardedEngine(eng, rows) if (world > 1 or args.sharded) else None
    ranks_seen = None
    if sharded is not None and getattr(sharded, "native", None) is not None:
        w_, r_ = C.c_int(), C.c_int()
        eng.lib.rfx_dist_world(eng._ctx, C.byref(w_), C.byref(r_))  # ncclCommCount / ncclCommUserRank of the context's communicator
        ranks_seen = int(w_.value)
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the RCCL communicator spans {ranks_seen} ranks, the launcher started {world}")

    if args.ab:
        job = Job(name, eng, sharded, rows, row0)
        for rep in range(3):
            for item in args.ab.split(","):
                fl, _, bpc = item.partition(":")
                eng.tune(blocks_per_cu=int(bpc or args.blocks_per_cu or 2), flags=int(fl))
                dt, kms, _ = timed(job, args.steps, 1, world)
                log(f"[ab] rep {rep} flags:bpc {item}: ms_per_step {dt * 1e3 / args.steps:.3f} kernel_ms {kms:.3f} "
                    f"GB/s {WORKLOADS[name]['bytes_per_row'] * rows / kms / 1e6:.0f}")
        return
    ldoor = None
    if (world > 1 or os.environ.get("RFX_BENCH_FORCE_LAUNCHER_DOOR")) and name in C_DOOR and name != "q7" and not args.engine_door:  # (the env: the N > 1 code on one rank)
        # every rank's shard through rfx_select, the operator layer's context inside the RCCL communicator: `value` is this door's
        if sharded is not None:
            sharded.close()
            sharded = None
        ldoor = launcher_door(args, name, eng, world, rank, rows, row0, total_rows)
        log(f"[bench] rank {rank}: {name} through rfx_select over {world} processes: {ldoor}")
        w = WORKLOADS[name]
        kms = ldoor["ms_per_step"]
        ach = w["bytes_per_row"] * total_rows / world / (kms * 1e-3) / 1e9
        main_r = dict(workload=name, rows_per_gpu=rows, total_rows=total_rows, ms_per_step=kms, rows_per_s=ldoor["rows_per_s"], kernel_ms=kms, achieved_GBps=ach,
                      frac=ach / HBM_PEAK_GBPS, result=ldoor["result"], verified=ldoor["verified"], paths={}, planner={}, rtc={})
        ranks_seen = ldoor["ranks_seen"]
    else:
        main_r = run_workload(name, eng, sharded, rows, row0, args.steps, args.warmup, world, total_rows)
    log(f"[bench] {name}: {main_r}")
    door = None
    if world == 1 and sharded is None and ldoor is None and name in C_DOOR and not args.engine_door:
        # the headline goes through the product's own door: the C operator rfx_select (Engine's time for the same query rides beside it)
        door = c_door(name, eng, rows, args.steps, args.warmup)
        log(f"[bench] {name} through rfx_select: {door}")
    also = {}
    FULL = ("c3w", "c2", "c2b", "c3", "c5")  # the BASELINE configs: the full step count, their own roofline block and CPU baseline
    if not args.no_also and world == 1 and not args.rows:
        for other in WORKLOADS:
            if other == name:
                continue
            try:
                full = other in FULL
                r = run_workload(other, eng, None, WORKLOADS[other]["rows"], 0, args.steps if full else max(3, args.steps // 4), args.warmup if full else 2, 1)
                also[other] = {k: r[k] for k in ("rows_per_gpu", "ms_per_step", "rows_per_s", "kernel_ms", "achieved_GBps", "frac", "verified", "paths", "planner", "rtc")}
                if other == "q7":  # the six-key row-hash shape through the C operator too (device column handles: the planner keeps its blocks)
                    dq = c_door("q7", eng, WORKLOADS["q7"]["rows"], max(3, args.steps // 4), 2, device_columns=True)
                    also[other]["rfx_select_ms_per_step"] = dq["ms_per_step"]
                    also[other]["rfx_select_verified"] = dq["verified"]
                if other == "g2":  # BOTH figures: `frac` prices the gather at LINE granularity (8.1 B per table row: what a monotone gather at 10 % must move);
                    # SURVEY App. A K4's own bytes are 8 B id + 8 B read + 8 B write per SELECTED row = 2.4 B per table row -- no gather can reach that on 128-byte lines
                    km = r["kernel_ms"] if r["kernel_ms"] > 0 else r["ms_per_step"]
                    also[other]["frac_survey_app_a_bytes"] = 2.4 * WORKLOADS[other]["rows"] / (km * 1e-3) / 1e9 / HBM_PEAK_GBPS
                also[other]["steps"] = args.steps if full else max(3, args.steps // 4)
                if full:
                    also[other]["roofline"] = roofline_block(other, r, 1)
                log(f"[bench] also {other}: {also[other]}")
            except Exception as e:  # noqa: BLE001
                also[other] = {"error": str(e)[:200]}

    boundary = None
    if rank == 0 and world == 1 and not args.no_also and not args.rows:
        try:
            boundary = boundary_overhead(eng)
            log(f"[bench] boundary: {boundary}")
        except Exception as e:  # noqa: BLE001
            boundary = {"error": str(e)[:200]}
    predicted, predicted_more = None, {}
    if rank == 0 and world == 1 and door and not args.rows and not args.no_predict:
        try:
            predicted = predicted_scaling(name, eng)
            log(f"[bench] predicted_scaling: {predicted}")
        except Exception as e:  # noqa: BLE001
            predicted = {"error": str(e)[:200]}
        for other in ("c3", "c5"):  # BASELINE configs[3] (C4 = C3 over 8 devices) and configs[4] (C5): the configs BASELINE.json names for 8 GPUs
            if other == name:
                continue
            try:
                predicted_more[other] = predicted_scaling(other, eng)
                log(f"[bench] predicted_scaling[{other}]: {predicted_more[other]}")
            except Exception as e:  # noqa: BLE001
                predicted_more[other] = {"error": str(e)[:200]}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(name, min(args.cpu_sample_rows, rows))
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cpu_baseline failed: {e}")
            cpu = None
        # the reference's CPU path beside the secondary workloads too (smaller samples: the whole run stays within minutes)
        t_cpu = time.perf_counter()
        for other in sorted(also, key=lambda o: (o not in FULL, o)):
            if other in ("c1", "c2", "c2b", "c3", "c3w", "q2", "x6", "k9", "c5") and "error" not in also[other]:
                if time.perf_counter() - t_cpu > 150:  # keep the default run within minutes
                    log(f"[bench] cpu_baseline({other}) skipped: time budget for the secondary baselines used up")
                    continue
                try:
                    cb = cpu_baseline(other, min(20_000_000, WORKLOADS[other]["rows"]), timeout=45, pools=(32,), sweep_budget_s=20)
                    also[other]["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "ms_per_query", "all_cores")}
                    also[other]["cpu_baseline"]["sample_rows"] = min(20_000_000, WORKLOADS[other]["rows"])
                except Exception as e:  # noqa: BLE001
                    log(f"[bench] cpu_baseline({other}) failed: {e}")

    if rank == 0:
        w = WORKLOADS[name]
        head = dict(main_r)
        if door:  # value = the C operator boundary; the Engine figures stay in `engine`
            # SURVEY 8d defines the metric on the MEDIAN step: `value`, `roofline.achieved / frac` come from it; `ms_per_step` stays the mean of the
            # timed loop (what a clock around the run sees) and the mean-based figures ride beside (value_mean, roofline.frac_mean)
            head.update(ms_per_step=door["ms_per_step"], rows_per_s=door["rows_per_s_median"], kernel_ms=door["median_ms"])
            head["achieved_GBps"] = w["bytes_per_row"] * total_rows / (door["median_ms"] * 1e-3) / 1e9
            head["frac"] = head["achieved_GBps"] / HBM_PEAK_GBPS
            head["rows_per_s_mean"] = door["rows_per_s"]
            head["frac_mean"] = w["bytes_per_row"] * total_rows / (door["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        rl, rc = C.c_int64(), C.c_int64()
        eng.lib.rfx_hip_rtc_stats(C.byref(rl), C.byref(rc))
        rd, rw = C.c_int64(), C.c_int64()
        eng.lib.rfx_hip_rtc_cache_stats(C.byref(rd), C.byref(rw))
        line = {
            "metric": METRIC,
            "value": head["rows_per_s"],
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": w["dtype"],
            "data": "synthetic",
            "config": {"workload": f"{name}: {w['desc']}", "rows_per_gpu": rows, "total_rows": total_rows,
                       "sharding": f"row-range x{world}" if world > 1 else "single GPU", "ranks_seen": ranks_seen, "resident": "HBM (columns generated on device)",
                       "result": main_r["result"], "verified": main_r["verified"], "paths": main_r["paths"],
                       "planner": main_r.get("planner"),
                       "door": door["door"] if door else ("rfx_select on this rank's row range as device column handles, the operator layer's context in the RCCL "
                                                          "communicator (rfx_ops_dist_init): a grouped answer stays sliced over the ranks (rfx_ops_set_rank_slices)" if ldoor else
                                                          "Engine (ctypes host over the library's planner, results stay on the device)")},
            "roofline": roofline_block(name, head, world),
            "cpu_baseline": cpu,
            "value_basis": ("median step of the timed loop (SURVEY 8d); ms_per_step = its mean, value_mean = rows / mean step" if door else "mean step of the timed loop"),
            "value_mean": head.get("rows_per_s_mean", head["rows_per_s"]),
            "median_ms": door["median_ms"] if door else None,
            "steps_ms_in_order": door["steps_ms"]["in_order"] if door else None,
            "rtc": {"launches_through_run_time_compiled_kernels": int(rl.value), "plans_compiled": int(rc.value),
                    "plans_loaded_from_the_disk_cache": int(rd.value), "code_objects_written": int(rw.value)},
        }
        if door:
            line["door"] = door
            line["engine"] = {k: main_r[k] for k in ("ms_per_step", "rows_per_s", "frac")}
            line["roofline"]["frac_mean"] = head["frac_mean"]
            line["roofline"]["basis"] = "whole query through rfx_select (every kernel, the plan walk, the result's read-back and host table), median step; dominant_kernel = that kernel alone by HIP events"
            if door.get("kernels_ms"):  # the dominant kernel alone: HIP events on the stream it is launched on, averaged over 5 more (untimed) steps
                km = max(door["kernels_ms"])
                line["roofline"]["dominant_kernel"] = {"avg_ms": km, "achieved": w["bytes_per_row"] * total_rows / (km * 1e-3) / 1e9,
                                                       "frac": w["bytes_per_row"] * total_rows / (km * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                       "all_bracketed_kernels_ms": door["kernels_ms"], "how": "hipEventRecord around the launch on the context's stream (rfx_hip_profile_kernels), 5 steps after the timed loop"}
        if ldoor:
            line["door"] = {k: ldoor[k] for k in ("ms_per_step", "rows_per_s", "verified", "collectives_per_query", "result_digest", "reproducible_mode")}
            if ldoor.get("exchange"):  # (the gloo test plumbing: the time is a host round trip per exchange, not the product's)
                line["door"]["exchange"] = ldoor["exchange"]
                line["config"]["door"] = line["config"]["door"].replace("in the RCCL communicator (rfx_ops_dist_init)", "exchanging through " + ldoor["exchange"])
        if also:
            line["also"] = also
        if boundary:
            line["boundary"] = boundary
        if predicted:
            line["predicted_scaling"] = predicted
        for other, pr in predicted_more.items():
            line["predicted_scaling_" + other] = pr
        print(json.dumps(line), flush=True)
        # the same run in <= 2 KB, LAST on stderr: a truncated tail of the record still carries every workload and the engine / door split
        def r3(v):
            return None if v is None else round(float(v), 3)
        compact = {"summary": name, "door_ms": r3(door["ms_per_step"]) if door else None, "engine_ms": r3(main_r["ms_per_step"]), "frac": r3(head["frac"]),
                   "median_ms": r3(door["median_ms"]) if door else None, "frac_mean": r3(head.get("frac_mean")),
                   "steps": [r3(v) for v in door["steps_ms"]["in_order"]] if door else None,
                   "kernels_ms": [r3(v) for v in door.get("kernels_ms") or []] if door else None,
                   "sclk_mhz": (door.get("clocks") or {}).get("sclk_mhz") if door else None, "power_w": (door.get("clocks") or {}).get("power_w") if door else None,
                   "phases_ms": {k: r3(v) for k, v in door["phases_ms"].items() if not isinstance(v, list)} if door and door.get("phases_ms") else None,
                   "also": {k: ([r3(v.get("ms_per_step")), r3(v.get("frac"))] + ([r3(v["rfx_select_ms_per_step"])] if "rfx_select_ms_per_step" in v else [])
                                if "error" not in v else "error") for k, v in also.items()},
                   "T_N": {n: [v["T_ms"], v["speedup"]] for n, v in predicted["per_n"].items()} if predicted and "per_n" in predicted else None,
                   **{"T_N_" + o: {n: [v["T_ms"], v["speedup"]] for n, v in pr["per_n"].items()} for o, pr in predicted_more.items() if "per_n" in pr},
                   "cpu": [cpu.get("value"), cpu.get("cores"), (cpu.get("all_cores") or {}).get("value")] if cpu else None}
        print(json.dumps(compact, separators=(",", ":")), file=sys.stderr, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def _stdout_is_for_the_json_line_only():
    """Libraries write to the process's stdout too (RCCL prints a version banner there when a communicator comes up): from here on file
    descriptor 1 points at stderr, and `print` keeps the real stdout -- which only the JSON line is printed to."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real, "w", buffering=1)


if __name__ == "__main__":
    main()
