"""N > 1 DEVICES over RCCL: one rank per GPU, the exchange through the library's own C entry points (rfx_dist_*: RCCL communicator
inside the context), BASELINE configs[3] (C4: the C3 group-by row-range sharded) and configs[4] (C5: three predicates, avg / min / max
over four f64 columns) against the UNSHARDED oracle, collective calls counted.  Skipped where the box has fewer devices than ranks
(the builder's boxes have one; the round-end driver's SCALE box has eight)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import ctypes as C
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from oracle import rfo
        from rayforce_amd.dist import ShardedEngine
        from rayforce_amd.engine import Engine
        rfo.set_threads(4)
        eng = Engine(rank)
        lib = eng.lib
        # ---- C4: select sum(v) by k, k i64 seed 4 mod 1e6, v f64 seed 5, rows [lo, hi) of the table on this rank
        n = 8_000_024
        lo, hi = n * rank // world, n * (rank + 1) // world
        sh = ShardedEngine(eng, hi - lo)
        assert sh.native is not None, "under NCCL the exchange must be the library's C one (rfx_dist_*)"
        w_, r_ = C.c_int(), C.c_int()
        assert lib.rfx_dist_world(eng._ctx, C.byref(w_), C.byref(r_)) == 0 and (w_.value, r_.value) == (world, rank), "ncclCommCount / ncclCommUserRank"
        assert sh.shard.row0 == lo and sh.shard.total_rows == n
        mine = {"k": eng.gen_i64(hi - lo, 4, 1_000_000, lo), "v": eng.gen_f64(hi - lo, 5, lo)}
        c0 = lib.rfx_dist_calls(eng._ctx)
        r = sh.group_by("k", [("sum", "v")], None, mine)
        assert lib.rfx_dist_calls(eng._ctx) - c0 == 3, "dense group-by: the scope gather + first MIN and sums SUM in one fused exchange"
        if rank == 0:
            want = rfo.select({"from": {"k": rfo.gen_i64(n, 4, 1_000_000), "v": rfo.gen_f64(n, 5)}, "by": "k", "s": ("sum", "v")})
            assert np.array_equal(r["keys"].cpu().numpy(), want["k"]), "keys / first-occurrence order across the shards"
            assert np.allclose(r["results"][0].cpu().numpy(), want["s"], rtol=1e-9, atol=0)
        # the metric's shape on the shards: filter -> group-by -> sum
        mine["a"] = eng.gen_i64(hi - lo, 2, 1_000_000, lo)
        r = sh.group_by("k", [("sum", "v"), ("count", "v")], ("<", "a", 100_000), mine)
        if rank == 0:
            full = {"k": rfo.gen_i64(n, 4, 1_000_000), "v": rfo.gen_f64(n, 5), "a": rfo.gen_i64(n, 2, 1_000_000)}
            want = rfo.select({"from": full, "where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v"), "c": ("count", "v")})
            assert np.array_equal(r["keys"].cpu().numpy(), want["k"]) and np.array_equal(r["results"][1].cpu().numpy(), want["c"])
            assert np.allclose(r["results"][0].cpu().numpy(), want["s"], rtol=1e-9, atol=0)
        # ---- C5: avg, min, max(d) where a < 0.316228 and b > 0.683772 and c != 0.25, four f64 columns seeds 6..9
        cols = {c: eng.gen_f64(hi - lo, s, lo) for c, s in zip("abcd", (6, 7, 8, 9))}
        where = ("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25))
        c0 = lib.rfx_dist_calls(eng._ctx)
        vals, sel = sh.filter_aggr([("avg", "d"), ("min", "d"), ("max", "d")], where, cols)
        assert lib.rfx_dist_calls(eng._ctx) - c0 == 1, "scalar aggregates: ONE all-gather of the partials"
        if rank == 0:
            full = {c: rfo.gen_f64(n, s) for c, s in zip("abcd", (6, 7, 8, 9))}
            want = rfo.select({"from": full, "where": where, "x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d")})
            assert abs(vals[0] - want["x"][0]) <= 1e-9 * abs(want["x"][0]) and vals[1] == want["y"][0] and vals[2] == want["z"][0]
            assert sel == int(rfo.mask_of(where, full).sum())
        ids = sh.where(("<", "a", 0.001), cols)  # ids: global, ascending, the ranks' pieces in rank order
        if rank == 0:
            assert np.array_equal(ids.cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 0.001), full)))
        sh.close()
        eng.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])  # (1: the same body on the one device the builder's boxes have)
def test_c4_c5_shards_over_rccl(built, world):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} devices (this box has {torch.cuda.device_count() if torch.cuda.is_available() else 0})")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


# ---------------------------------------------------------------- ONE evaluator process over the node's devices (RFX_DEVICES)
_ONE_PROCESS = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import rfo
from rayforce_amd import hostobj as H, _lib as L
from test_gpu_parity import same_f64
import ctypes as C
ops = H.lib()
ops.rfx_host_bind()
n = 4_000_037
host = {"k": rfo.gen_i64(n, 4, 100_000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5), "b": rfo.gen_f64(n, 3), "c": rfo.gen_f64(n, 8), "d": rfo.gen_f64(n, 9)}
host["ks"] = host["k"] * 1_000_003 + 17  # sparse keys: the hashed tables, merged by an all-gather
tab = H.table(host)
queries = [
    {"s": ("sum", "a"), "c": ("count", "a"), "where": ("<", "a", 100_000)},                                     # C4's shape
    {"x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d"), "where": ("and", ("<", "v", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25))},  # C5's
    {"s": ("sum", "v"), "by": "k", "where": ("<", "a", 100_000)},                                               # the metric's
    {"s": ("sum", "v"), "f": ("first", "a"), "m": ("max", "a"), "by": "k"},
    {"s": ("sum", "v"), "c": ("count", "a"), "by": "ks"},
    {"s": ("sum", "b"), "by": "k", "where": ("and", ("<", ("+", "v", "b"), 0.7), (">", ("-", "a", "k"), 1000))},
    {"where": ("<", "a", 3000)},
]
def ask(q):
    d = H.select_dict(q, tab)
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    assert ops.rfx_last_select_on_gpu() == 1, (q, ops.rfx_ops_last_error())
    out = H.table_to_numpy(r)
    ops.rfx_host_drop(r); ops.rfx_host_drop(d)
    return out
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
                assert np.array_equal(g, w, equal_nan=w.dtype == np.float64), (name, q)
x = C.c_void_p(ops.rfx_ops_exec())
if WORLD > 1:
    assert ops.rfx_ops_shards() == WORLD and ops.rfx_exec_shards(x) == WORLD
    devs = {ops.rfx_hip_ctx_device(C.c_void_p(ops.rfx_exec_ctx(x, s))) for s in range(WORLD)}
    assert len(devs) == WORLD, devs                                   # one shard per device
    assert ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_RCCL) > 0          # the fused exchange over xGMI merged the tables
    assert ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_KERNEL) == 0       # (no two shards share a device)
    assert ops.rfx_exec_stat(x, L.RFX_XSTAT_SLICED) > 0               # every device emitted and read back ITS range of the groups (the FIRST query stays whole)
else:  # the same script on the one device of the builder's boxes: two shards on it, merged by the kernel
    assert ops.rfx_ops_shards() == 2 and ops.rfx_exec_stat(x, L.RFX_XSTAT_MERGES_KERNEL) > 0
print("ONE-PROCESS-OK")
'''


@pytest.mark.parametrize("world", [1, 2, 4, 8])  # (1: the script itself on one device, as two shards of it)
def test_one_evaluator_process_over_the_devices(built, world):
    """RFX_DEVICES=0,..,world-1: ONE process, one shard per device -- what a RayforceDB evaluator that owns the node does.  rfx_select splits the
    host table row-range over the devices at first touch, every device's pass runs on its own host thread, the group tables merge in one fused
    RCCL exchange (ncclCommInitAll): answers against the UNSHARDED oracle, the planner's counters say who merged."""
    import os, subprocess, sys
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} devices (this box has {torch.cuda.device_count() if torch.cuda.is_available() else 0})")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RFX_DEVICES=",".join(str(i) for i in range(world)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RFX_SHARDS", None)
    if world == 1:
        env["RFX_SHARDS"] = "2"
    code = f"ROOT = {root!r}\nWORLD = {world}\n" + _ONE_PROCESS
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ONE-PROCESS-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
