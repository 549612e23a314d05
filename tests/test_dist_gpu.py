"""Two ranks, ONE GPU: the row-sharded driver end to end (HIP kernels per rank + the real merge code), with `gloo`
carrying the collectives because RCCL refuses two ranks on one device.  Checker: the unsharded CPU oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NULL = -(2**63)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rfo
        from rayforce_amd.dist import ShardedEngine
        from rayforce_amd.engine import Engine
        rfo.set_threads(4)
        n = 600_011
        full = {"k": rfo.gen_i64(n, 4, 50_000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5) - 0.5}
        full["a"][::97] = NULL
        full["v"][::89] = np.nan
        cut = [0, 250_007, n]
        eng = Engine(0)
        mine = {c: eng.column(x[cut[rank]:cut[rank + 1]]) for c, x in full.items()}
        sh = ShardedEngine(eng, cut[rank + 1] - cut[rank])
        assert sh.shard.row0 == cut[rank] and sh.shard.total_rows == n
        # scalar aggregates
        where = ("and", ("<", "a", 600_000), (">", "v", -0.4))
        aggs = [("sum", "a"), ("sum", "v"), ("min", "v"), ("max", "a"), ("avg", "v"), ("count", "a")]
        vals, sel = sh.filter_aggr(aggs, where, mine)
        want = rfo.select({"from": full, "where": where, **{f"o{i}": a for i, a in enumerate(aggs)}})
        for i, v in enumerate(vals):
            w = want[f"o{i}"][0]
            assert (abs(v - w) <= 1e-9 * abs(w)) if isinstance(v, float) else v == int(w), (aggs[i], v, w)
        assert sel == int(rfo.mask_of(where, full).sum())
        # where ids: global, ascending
        ids = sh.where(("<", "a", 50_000), mine)
        assert np.array_equal(ids.cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 50_000), full)))
        # dense group-by (partitioned path on each rank, tables all-reduced, ranked by GLOBAL first row)
        gaggs = [("sum", "v"), ("sum", "a"), ("min", "v"), ("max", "a"), ("avg", "a"), ("count", "v"), ("first", "a"), ("first", "v")]
        for w_ in (None, (">", "v", -0.25)):
            r = sh.group_by("k", gaggs, w_, mine)
            qq = {"from": full, "by": "k", **{f"o{i}": a for i, a in enumerate(gaggs)}}
            if w_:
                qq["where"] = w_
            want = rfo.select(qq)
            assert np.array_equal(r["keys"].cpu().numpy(), want["k"]), "keys / first-occurrence order"
            for i, res in enumerate(r["results"]):
                g, w = res.cpu().numpy(), want[f"o{i}"]
                if w.dtype == np.float64:
                    assert np.array_equal(np.isnan(g), np.isnan(w))
                    ok = ~np.isnan(w)
                    assert np.allclose(g[ok], w[ok], rtol=1e-9, atol=0), gaggs[i]
                else:
                    assert np.array_equal(g, w), gaggs[i]
        # sparse keys: hashed tables all-gathered and merged
        sparse = dict(full)
        sparse["k"] = full["k"] * 1_000_003 - 5
        mine_s = dict(mine)
        mine_s["k"] = eng.column(sparse["k"][cut[rank]:cut[rank + 1]])
        r = sh.group_by("k", [("sum", "v"), ("count", "a")], None, mine_s)
        want = rfo.select({"from": sparse, "by": "k", "s": ("sum", "v"), "c": ("count", "a")})
        assert np.array_equal(r["keys"].cpu().numpy(), want["k"]) and np.array_equal(r["results"][1].cpu().numpy(), want["c"])
        # more outputs than one table set carries: several launches, each all-reduced with ITS aggregates' reduce ops
        many = [("max", "a"), ("sum", "v"), ("min", "a"), ("avg", "v"), ("count", "a"), ("sum", "a"), ("min", "v"), ("max", "v"), ("avg", "a"), ("min", "a"), ("sum", "v")]
        r = sh.group_by("k", many, None, mine)
        want = rfo.select({"from": full, "by": "k", **{f"o{i}": a for i, a in enumerate(many)}})
        for i, res in enumerate(r["results"]):
            g, w = res.cpu().numpy(), want[f"o{i}"]
            if w.dtype == np.float64:
                ok = ~np.isnan(w)
                assert np.array_equal(np.isnan(g), np.isnan(w)) and np.allclose(g[ok], w[ok], rtol=1e-9, atol=0), many[i]
            else:
                assert np.array_equal(g, w), many[i]
        # key tuples beyond the composite key (the reference's row-hash path) across ranks (round 5): every rank groups its rows on the row hash, the
        # hashed tables are gathered and re-inserted, the tuples proven by a (min, max) pair per key column through the same exchange -- the maxima are
        # the result's key columns (index_group_list, core/index.c:2731-2790)
        wide = {"k1": rfo.gen_i64(n, 41, 50) * (1 << 50), "k2": rfo.gen_i64(n, 42, 40) * (1 << 45) - (1 << 50), "k3": rfo.gen_i64(n, 43, 3), "v": full["v"], "a": full["a"]}
        mine_w = {c: eng.column(x[cut[rank]:cut[rank + 1]]) for c, x in wide.items()}
        from rayforce_amd._lib import RfxError
        r = sh.group_by(["k1", "k2", "k3"], [("sum", "v"), ("count", "a")], None, mine_w)
        want = rfo.select({"from": wide, "by": {"k1": "k1", "k2": "k2", "k3": "k3"}, "s": ("sum", "v"), "c": ("count", "a")})
        for i, nm in enumerate(("k1", "k2", "k3")):
            assert np.array_equal(r["key_columns"][i].cpu().numpy(), want[nm]), nm
        assert np.array_equal(r["results"][1].cpu().numpy(), want["c"])
        gs, ws = r["results"][0].cpu().numpy(), want["s"]
        okv = ~np.isnan(ws)  # (v holds NaNs: a group's sum over them is NaN on both sides)
        assert np.array_equal(np.isnan(gs), np.isnan(ws)) and np.allclose(gs[okv], ws[okv], rtol=1e-9, atol=1e-12)
        # ... a null among the key tuples rides through the proof as max + 1 (MIN / MAX skip nulls) and comes back as the null
        wide_n = dict(wide)
        wide_n["k3"] = wide["k3"].copy()
        wide_n["k3"][5::97] = -(2**63)
        mine_n = dict(mine_w)
        mine_n["k3"] = eng.column(wide_n["k3"][cut[rank]:cut[rank + 1]])
        r = sh.group_by(["k1", "k2", "k3"], [("count", "a"), ("first", "a"), ("max", "a")], None, mine_n)  # (FIRST values beside the proven tuples)
        want = rfo.select({"from": wide_n, "by": {"k1": "k1", "k2": "k2", "k3": "k3"}, "c": ("count", "a"), "f": ("first", "a"), "m": ("max", "a")})
        for i, nm in enumerate(("k1", "k2", "k3")):
            assert np.array_equal(r["key_columns"][i].cpu().numpy(), want[nm]), nm
        for i, nm in enumerate(("c", "f", "m")):
            assert np.array_equal(r["results"][i].cpu().numpy(), want[nm]), nm
        # ... while composite keys that fit 64 bits group across the ranks like any dense / hashed key
        r = sh.group_by(["k", "k3"], [("sum", "v"), ("count", "a"), ("max", "a")], None, {**mine, "k3": mine_w["k3"]})
        want = rfo.select({"from": {**full, "k3": wide["k3"]}, "by": {"k": "k", "k3": "k3"}, "s": ("sum", "v"), "c": ("count", "a"), "m": ("max", "a")})
        for i, nm in enumerate(("k", "k3")):
            assert np.array_equal(r["key_columns"][i].cpu().numpy(), want[nm]), nm
        assert np.array_equal(r["results"][1].cpu().numpy(), want["c"]) and np.array_equal(r["results"][2].cpu().numpy(), want["m"])
        assert sh.transport.calls > 0
        sh.close()
        # The same two ranks through the C OPERATOR (rfx_select on this rank's rows as host columns, the planner's exchanges through the transport hooks),
        # every rank keeping only ITS range of the groups (rfx_ops_set_rank_slices): the ranks' tables end to end are the oracle's answer, in its order --
        # in the default mode and with reproducible sums (one more small gather: the ranks agree on the scale)
        import ctypes as C
        from rayforce_amd import hostobj as H, _lib as L
        from rayforce_amd.dist import _TorchTransport
        ops = H.lib()
        ops.rfx_host_bind()
        v0 = H.vector(np.arange(4, dtype=np.int64))  # (any operator call brings the layer's context and planner up)
        ops.rfx_host_drop(ops.rfx_sum(v0))
        ops.rfx_host_drop(v0)
        xo = C.c_void_p(ops.rfx_ops_exec())

        class _OpsCtx:
            lib = ops
            _ctx = C.c_void_p(ops.rfx_exec_ctx(xo, 0))
        tr = _TorchTransport(_OpsCtx, None)
        L.check(ops.rfx_exec_set_transport(xo, C.byref(tr.struct)), "exec_set_transport")
        rk = C.c_int(-1)
        assert ops.rfx_exec_ranks(xo, C.byref(rk)) == 2 and rk.value == rank
        whole = dict(full, w=rfo.gen_f64(n, 6) * 100.0 - 30.0)
        tab = H.table({c: np.ascontiguousarray(x[cut[rank]:cut[rank + 1]]) for c, x in whole.items()})
        qd = {"s": ("sum", "w"), "x": ("avg", "w"), "c": ("count", "a"), "m": ("max", "a"), "where": (">", "w", -10.0), "by": "k"}
        d = H.select_dict(qd, tab)
        want = rfo.select({"from": whole, **qd})
        ncalls = {}
        for slices in (0, 1):
            for det in (0, 1, 2):
                assert ops.rfx_ops_set_rank_slices(slices) == 0 and ops.rfx_ops_set_deterministic(det) == 0
                c0 = tr.calls
                r = ops.rfx_select(d)
                assert r and not H.is_error(r), H.error_text(r)
                assert int(ops.rfx_last_select_on_gpu()) == 1
                ncalls[slices, det] = tr.calls - c0
                got = H.table_to_numpy(r)
                ops.rfx_host_drop(r)
                g0, gn = C.c_int64(0), C.c_int64(len(want["k"]))
                if slices:
                    ops.rfx_exec_split(len(want["k"]), 2, rank, C.byref(g0), C.byref(gn))
                    assert 0 < gn.value < len(want["k"])
                for nm, w in want.items():
                    w = w[g0.value:g0.value + gn.value]
                    assert got[nm].shape == w.shape, (slices, det, nm, got[nm].shape, w.shape)
                    assert np.allclose(got[nm], w, rtol=1e-9, atol=1e-9) if w.dtype == np.float64 else np.array_equal(got[nm], w), (slices, det, nm)
        # ... the hashed paths too (sparse keys; a key tuple beyond a 64-bit composite key): whole and sliced, default and two limbs
        whole2 = dict(whole, k=whole["k"] * 1_000_003 - 5, k2=rfo.gen_i64(n, 44, 7) * (1 << 50), k3=rfo.gen_i64(n, 45, 5) * (1 << 45))
        tab2 = H.table({c: np.ascontiguousarray(x[cut[rank]:cut[rank + 1]]) for c, x in whole2.items()})
        for qd2 in ({"s": ("sum", "w"), "c": ("count", "a"), "by": "k"}, {"s": ("sum", "w"), "m": ("max", "a"), "by": {"k": "k", "k2": "k2", "k3": "k3"}}):
            d2 = H.select_dict(qd2, tab2)
            want2 = rfo.select({"from": whole2, **qd2})
            for slices in (0, 1):
                for det in (0, 2):
                    assert ops.rfx_ops_set_rank_slices(slices) == 0 and ops.rfx_ops_set_deterministic(det) == 0
                    r = ops.rfx_select(d2)
                    assert r and not H.is_error(r), H.error_text(r)
                    assert int(ops.rfx_last_select_on_gpu()) == 1, (qd2, slices, det)
                    got = H.table_to_numpy(r)
                    ops.rfx_host_drop(r)
                    g0, gn = C.c_int64(0), C.c_int64(len(want2["k"]))
                    if slices:
                        ops.rfx_exec_split(len(want2["k"]), 2, rank, C.byref(g0), C.byref(gn))
                    for nm, w in want2.items():
                        w = w[g0.value:g0.value + gn.value]
                        assert got[nm].shape == w.shape, (qd2, slices, det, nm, got[nm].shape, w.shape)
                        assert np.allclose(got[nm], w, rtol=1e-9, atol=1e-9) if w.dtype == np.float64 else np.array_equal(got[nm], w), (qd2, slices, det, nm)
            ops.rfx_host_drop(d2)
        ops.rfx_host_drop(tab2)
        assert ncalls[0, 0] >= 3 and ncalls[1, 0] == ncalls[0, 0] and ncalls[0, 1] == ncalls[0, 0] + 2 == ncalls[1, 1], ncalls  # (one gather per rewritten aggregate ...)
        assert ncalls[0, 2] >= ncalls[0, 1] and ncalls[1, 2] == ncalls[0, 2], ncalls  # (... however many limbs; more aggregates may mean one more table exchange)
        ops.rfx_ops_set_rank_slices(0)
        ops.rfx_ops_set_deterministic(0)
        ops.rfx_exec_set_transport(xo, None)
        for o in (d, tab):
            ops.rfx_host_drop(o)
        eng.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _nccl_worker(port, q):
    """ONE rank under NCCL: the exchange runs through the library's C entry points (rfx_dist.hip, RCCL communicator in the context)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from oracle import rfo
        from rayforce_amd.dist import ShardedEngine
        from rayforce_amd.engine import Engine
        rfo.set_threads(4)
        n = 500_009
        full = {"k": rfo.gen_i64(n, 4, 60_000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5) - 0.5}
        full["a"][::97] = NULL
        eng = Engine(0)
        mine = {c: eng.column(x) for c, x in full.items()}
        sh = ShardedEngine(eng, n)
        assert sh.native is not None, "under NCCL the exchange must be the C one"
        lib = eng.lib
        c0 = lib.rfx_dist_calls(eng._ctx)
        aggs = [("sum", "a"), ("sum", "v"), ("min", "v"), ("max", "a"), ("avg", "v"), ("count", "a")]
        where = ("<", "a", 600_000)
        vals, sel = sh.filter_aggr(aggs, where, mine)
        assert lib.rfx_dist_calls(eng._ctx) - c0 == 1, "scalar aggregates: ONE all-gather"
        want = rfo.select({"from": full, "where": where, **{f"o{i}": a for i, a in enumerate(aggs)}})
        for i, v in enumerate(vals):
            w = want[f"o{i}"][0]
            assert (abs(v - w) <= 1e-9 * abs(w)) if isinstance(v, float) else v == int(w), (aggs[i], v, w)
        assert sel == int(rfo.mask_of(where, full).sum())
        c0 = lib.rfx_dist_calls(eng._ctx)
        r = sh.group_by("k", [("sum", "v")], None, mine)
        assert lib.rfx_dist_calls(eng._ctx) - c0 == 3, "select sum(v) by k: the scope gather + first MIN and sums SUM in one fused exchange"
        want = rfo.select({"from": full, "by": "k", "s": ("sum", "v")})
        assert np.array_equal(r["keys"].cpu().numpy(), want["k"]) and np.allclose(r["results"][0].cpu().numpy(), want["s"], rtol=1e-9, atol=0)
        gaggs = [("sum", "v"), ("sum", "a"), ("min", "v"), ("max", "a"), ("avg", "a"), ("count", "v"), ("first", "a")]
        r = sh.group_by("k", gaggs, (">", "v", -0.25), mine)
        want = rfo.select({"from": full, "by": "k", "where": (">", "v", -0.25), **{f"o{i}": a for i, a in enumerate(gaggs)}})
        assert np.array_equal(r["keys"].cpu().numpy(), want["k"])
        for i, res in enumerate(r["results"]):
            g, w = res.cpu().numpy(), want[f"o{i}"]
            if w.dtype == np.float64:
                ok = ~np.isnan(w)
                assert np.array_equal(np.isnan(g), np.isnan(w)) and np.allclose(g[ok], w[ok], rtol=1e-9, atol=0), gaggs[i]
            else:
                assert np.array_equal(g, w), gaggs[i]
        ids = sh.where(("<", "a", 50_000), mine)
        assert np.array_equal(ids.cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 50_000), full)))
        sparse = {**mine, "k": eng.column(full["k"] * 1_000_003 - 5)}
        r = sh.group_by("k", [("sum", "v"), ("count", "a")], None, sparse)
        want = rfo.select({"from": {**full, "k": full["k"] * 1_000_003 - 5}, "by": "k", "s": ("sum", "v"), "c": ("count", "a")})
        assert np.array_equal(r["keys"].cpu().numpy(), want["k"]) and np.array_equal(r["results"][1].cpu().numpy(), want["c"])
        sh.close()
        eng.close()
        q.put("ok")
    except Exception:  # noqa: BLE001
        import traceback
        q.put(traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_c_exchange_over_rccl_one_rank(built):
    """The C entry points of the exchange (rfx_dist_*) under a real RCCL communicator: one rank (RCCL refuses two per device), every
    collective really issued; dense group-by = two calls in one ncclGroup, scalar aggregates = one all-gather."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=300)
    p.join(timeout=60)
    assert msg == "ok", msg


@pytest.mark.parametrize("workload", ["c3w", "c2"])
def test_bench_under_the_launcher_two_ranks_one_gpu(workload):
    """The driver's N > 1 command line -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with two ranks on ONE
    GPU: every rank's row range through rfx_select, the planner's exchanges over gloo (RFX_BENCH_BACKEND: RCCL refuses two ranks on a
    device), rank 0's single JSON line on stdout.  The time is a host round trip per exchange; the launch, the door, the property check
    of the answer and the record are the N > 1 code."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RFX_BENCH_SAME_DEVICE="1", RFX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "40000000", "--workload", workload],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # stdout carries the one line only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["config"]["ranks_seen"] == 2 and rec["config"]["total_rows"] == 40_000_000
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["roofline"]["frac"] > 0 and rec["cpu_baseline"] is None
    assert rec["door"]["verified"] and rec["door"]["collectives_per_query"] >= 1 and "gloo" in rec["door"]["exchange"]
    if workload == "c3w":
        assert 950_000 < rec["config"]["result"]["groups"] <= 1_000_000 and rec["door"]["collectives_per_query"] == 3  # scope, tables, first rows' order
        # every rank read back and returned only ITS range of the groups (rfx_ops_set_rank_slices); the property check folded the ranks' pieces
        assert rec["config"]["result"]["returned"].startswith("every rank its range") and "over the ranks' slices" in rec["door"]["verified"]


def test_reproducible_sums_across_rank_counts():
    """RFX_DETERMINISTIC=1 under the launcher: the ranks agree on ONE scale for the fixed-point sums (max |x| and the row count over all ranks' row ranges: one
    more small all-gather per query), so two ranks over the halves of the table return the very bits one rank returns over the whole of it."""
    import json
    import subprocess
    import sys
    recs = []
    for ranks in (2, 1):
        env = dict(os.environ, RFX_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RFX_BENCH_WHOLE_RESULT="1")  # (the digest is of the whole answer)
        env.update({"RFX_BENCH_SAME_DEVICE": "1", "RFX_BENCH_BACKEND": "gloo"} if ranks > 1 else {"RFX_BENCH_FORCE_LAUNCHER_DOOR": "1"})
        cmd = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--rows", "30000001", "--workload", "c3w", "--no-also", "--no-cpu-baseline", "--no-predict"]
        if ranks > 1:
            cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + cmd
        out = subprocess.run([sys.executable] + cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        recs.append(json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1]))
    two, one = recs
    assert two["door"]["reproducible_mode"] and one["door"]["reproducible_mode"] and two["door"]["verified"] and one["door"]["verified"]
    assert two["door"]["collectives_per_query"] == 4  # scope, tables, first rows' order + the scale
    assert two["config"]["result"]["groups"] == one["config"]["result"]["groups"] and two["door"]["result_digest"] == one["door"]["result_digest"]
