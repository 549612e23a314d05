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
