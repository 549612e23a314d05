"""On-disk columns into HBM: RayforceDB column files and splayed tables written by the REAL reference are
loaded with the pipelined pinned-staging path and queried; results against the oracle."""
import os

import numpy as np
import pytest

from oracle import ref, rfo

pytestmark = pytest.mark.gpu


def test_column_file_roundtrip(eng, tmp_path):
    for n in (0, 1, 1000, 5_000_011):  # the last one spans several 32 MB staging chunks
        a = rfo.gen_i64(n, 3, 2**40) - 2**39
        v = rfo.gen_f64(n, 4)
        pa, pv = str(tmp_path / f"a{n}"), str(tmp_path / f"v{n}")
        ref.write_col(pa, a)
        ref.write_col(pv, v)
        da, dv = eng.load_column(pa), eng.load_column(pv)
        assert da.dtype.is_floating_point is False and dv.dtype.is_floating_point
        assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(dv.cpu().numpy(), v)
    up = eng.upload(v)
    assert np.array_equal(up.cpu().numpy(), v)


def test_bad_files_are_refused(eng, tmp_path):
    from rayforce_amd._lib import RfxError
    p = tmp_path / "junk"
    p.write_bytes(b"not a column file at all")
    with pytest.raises(RfxError, match="not a RayforceDB column file"):
        eng.load_column(str(p))
    with pytest.raises(RfxError, match="cannot open"):
        eng.load_column(str(tmp_path / "missing"))
    q = tmp_path / "short"
    ref.write_col(str(q), np.arange(100, dtype=np.int64))
    q.write_bytes(q.read_bytes()[:200])
    with pytest.raises(RfxError, match="truncated"):
        eng.load_column(str(q))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/rayforce not built")
def test_splayed_table_written_by_the_reference(eng, tmp_path):
    n = 200_003
    host = {"k": rfo.gen_i64(n, 4, 3000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5)}
    d = str(tmp_path / "tab") + "/"
    with ref.Session() as s:
        s.table("t", host)
        s.eval(f'(set "{d}" t)')  # io_set_table_splayed, core/io.c:1194
        s.run(threads=8)
    assert sorted(os.listdir(d)) == [".d", "a", "k", "v"]
    t = eng.load_splayed(d)
    assert list(t) == ["k", "a", "v"]
    got = eng.select({"from": t, "where": ("<", "a", 500_000), "by": "k", "s": ("sum", "v"), "x": ("sum", ("*", "a", "v"))})
    want = rfo.select({"from": host, "where": ("<", "a", 500_000), "by": "k", "s": ("sum", "v"), "x": ("sum", ("*", "a", "v"))})
    assert np.array_equal(got["k"].cpu().numpy(), want["k"])
    assert np.allclose(got["s"].cpu().numpy(), want["s"], rtol=1e-9, atol=0) and np.allclose(got["x"].cpu().numpy(), want["x"], rtol=1e-9, atol=0)
    only = eng.load_splayed(d, ["v"])
    assert list(only) == ["v"] and np.array_equal(only["v"].cpu().numpy(), host["v"])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/rayforce not built")
def test_parted_table_written_by_the_reference(eng, tmp_path):
    """`(get-parted root 'tab)` layout (core/vary.c:185-392): one splayed table per date directory, written by the reference;
    loaded whole and with partition pruning on the virtual Date column; a pruned partition is never opened (its files are
    deleted before the pruned load); the reference's own answers over its parted table stand beside the device's."""
    import datetime
    import shutil
    root = str(tmp_path / "db") + "/"
    dates = ["2024.01.03", "2024.01.01", "2024.02.29", "2024.01.02"]  # directory order is not date order
    lens = [70_001, 50_000, 33_333, 0 + 90_007]
    hosts = {}
    for i, (d, n) in enumerate(zip(dates, lens)):  # one reference process per partition (a session stages its columns by name)
        hosts[d] = {"k": rfo.gen_i64(n, 40 + i, 500), "a": rfo.gen_i64(n, 50 + i, 1_000_000), "v": rfo.gen_f64(n, 60 + i)}
        with ref.Session() as s:
            s.table("t", hosts[d])
            s.eval(f'(set "{root}{d}/tab/" t)')
            s.run(threads=8)
    with ref.Session() as s:
        s.eval(f"(set p (get-parted \"{root}\" 'tab))")
        s.eval("(set r1 (select {s: (sum v) c: (count a) from: p where: (== Date 2024.01.02)}))")
        s.out("s1", "(at r1 's)")
        s.out("c1", "(at r1 'c)")
        r = s.run(threads=8)
    order = sorted(dates)
    whole = {c: np.concatenate([hosts[d][c] for d in order]) for c in ("k", "a", "v")}
    day = lambda d: (datetime.date(*map(int, d.split("."))) - datetime.date(2000, 1, 1)).days
    whole["Date"] = np.concatenate([np.full(len(hosts[d]["k"]), day(d), np.int64) for d in order])
    t = eng.load_parted(root, "tab")
    assert list(t) == ["Date", "k", "a", "v"]
    for c in whole:
        assert np.array_equal(t[c].cpu().numpy(), whole[c]), c
    q = {"where": ("<", "a", 400_000), "by": "Date", "s": ("sum", "v"), "c": ("count", "a"), "m": ("max", "k")}
    got, want = eng.select({"from": t, **q}), rfo.select({"from": whole, **q})
    assert np.array_equal(got["Date"].cpu().numpy(), want["Date"]) and np.array_equal(got["c"].cpu().numpy(), want["c"])
    assert np.allclose(got["s"].cpu().numpy(), want["s"], rtol=1e-9, atol=0)
    # the reference's answers over ITS parted table
    one = eng.load_parted(root, "tab", where=("==", "Date", "2024.01.02"))
    assert one["v"].numel() == 90_007 and int(one["Date"][0]) == day("2024.01.02")
    g1 = eng.select({"from": one, "s": ("sum", "v"), "c": ("count", "a")})
    assert int(g1["c"][0]) == int(r["c1"][0]) and abs(float(g1["s"][0]) - float(r["s1"][0])) <= 1e-9 * abs(float(r["s1"][0]))
    # (a Date predicate AND-ed with a column predicate is not compared with the reference: it answers twice one partition's sum
    #  there -- 2 x 124.92 where the rows add up to 371.63 on a two-partition probe -- so the oracle stands in)
    late = eng.load_parted(root, "tab", ["a", "v"], where=(">=", "Date", "2024.01.02"))
    g2 = eng.select({"from": late, "where": ("<", "a", 500_000), "s": ("sum", "v")})
    sel = (whole["Date"] >= day("2024.01.02")) & (whole["a"] < 500_000)
    assert abs(float(g2["s"][0]) - float(whole["v"][sel].sum())) <= 1e-9 * float(whole["v"][sel].sum())
    # pruning happens on the directory list: a partition that fails the predicate is not even opened
    shutil.rmtree(os.path.join(root, "2024.01.01", "tab"))
    again = eng.load_parted(root, "tab", ["a", "v"], where=("or", ("==", "Date", "2024.02.29"), ("and", (">", "Date", "2024.01.01"), ("<", "Date", day("2024.01.03")))))
    assert again["v"].numel() == 90_007 + 33_333
    from rayforce_amd._lib import RfxError
    with pytest.raises(RfxError, match="Date column only"):
        eng.load_parted(root, "tab", where=("<", "a", 5))


def test_large_blocks_come_back_through_pinned_staging(eng):
    """rfx_hip_d2h_pipelined (what rfx_exec_groups_fetch_all uses for result columns >= 64 MB): 32 MB chunks through pinned staging, the destination written
    by several host threads -- every byte where it belongs, ragged tail included; small blocks take the plain copy."""
    import ctypes as C
    import numpy as np
    import torch
    for n in (1 << 20, (200 << 20) // 8 + 12_345):
        src = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
        dst = np.empty(n, dtype=np.int64)
        torch.cuda.synchronize()
        rc = eng.lib.rfx_hip_d2h_pipelined(eng._ctx, C.c_void_p(dst.ctypes.data), C.c_void_p(src.data_ptr()), C.c_size_t(n * 8))
        assert rc == 0
        assert dst[0] == 1 and dst[-1] == 3 * (n - 1) + 1 and np.array_equal(dst, np.arange(n, dtype=np.int64) * 3 + 1)
