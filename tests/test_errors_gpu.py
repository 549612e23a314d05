"""Error behaviour of the flat ABI on a live device: bad arguments are refused with a code + message, never a crash, and
nothing silently falls back."""
import ctypes as C

import pytest
import torch

from rayforce_amd import _lib as L
from rayforce_amd._lib import RfxError

pytestmark = pytest.mark.gpu


def test_argument_validation(eng):
    lib = eng.lib
    a = eng.gen_i64(1000, 1, 100)
    v = eng.gen_f64(1000, 2)
    with pytest.raises(RfxError, match="length mismatch"):
        eng.filter_aggr([("sum", v)], ("<", a, eng.gen_i64(999, 1, 100)))
    with pytest.raises(RfxError, match="unsupported column dtype"):
        eng.filter_aggr([("sum", a.to(torch.int32))], None)
    # more comparisons than one fused pass holds (RFX_MAX_PREDS) are not an error since the nested-tree work: the tree goes through
    # materialised masks, like any tree the fused descriptors cannot express -- and must still be right
    ids = eng.where(("and", *[("<", a, 90 - i) for i in range(9)]))
    assert torch.equal(ids, torch.nonzero(a < 82).flatten())  # (the flat ABI still refuses nine descriptors: RFX_ELIMIT below)
    with pytest.raises(RfxError, match="needs a column"):
        eng.filter_aggr([("sum", None)], None, nrows=10)
    # raw ABI: bad operator / aggregate kind / logic are RFX_EINVAL with a message
    p = (L.Pred * 1)()
    p[0].d_col, p[0].col_type, p[0].rhs_type, p[0].op = a.data_ptr(), L.RFX_I64, L.RFX_I64, 99
    out = torch.empty(128, dtype=torch.uint8, device=eng.device)
    ag = (L.Agg * 1)()
    ag[0].d_col, ag[0].col_type, ag[0].kind = a.data_ptr(), L.RFX_I64, L.RFX_AGG_SUM
    assert lib.rfx_hip_filter_aggr(eng._ctx, p, 1, L.RFX_AND, ag, 1, 1000, 0, out.data_ptr()) == -2
    assert b"comparison operator" in lib.rfx_hip_last_error()
    p[0].op = L.RFX_LT
    ag[0].kind = 42
    assert lib.rfx_hip_filter_aggr(eng._ctx, p, 1, L.RFX_AND, ag, 1, 1000, 0, out.data_ptr()) == -2
    ag[0].kind = L.RFX_AGG_SUM
    assert lib.rfx_hip_filter_aggr(eng._ctx, p, 1, 7, ag, 1, 1000, 0, out.data_ptr()) == -2
    assert lib.rfx_hip_filter_aggr(eng._ctx, p, 9, L.RFX_AND, ag, 1, 1000, 0, out.data_ptr()) == -5  # RFX_ELIMIT
    # call-sequence errors
    assert lib.rfx_hip_where_emit(eng._ctx, 0, out.data_ptr()) in (0, -6)  # fine after a begin, RFX_ESTATE otherwise
    ms = C.c_float()
    eng.profile(False)
    assert lib.rfx_hip_last_kernel_ms(eng._ctx, C.byref(ms)) == -6
    # the engine is still healthy afterwards
    assert eng.sum(a) == int(a.sum())


def test_hash_table_overflow_is_reported(eng):
    """An open-addressed table that is too small must say so instead of dropping groups."""
    n = 10_000
    k = eng.column(torch.arange(n, dtype=torch.int64).numpy() * 1_000_003)
    v = eng.gen_f64(n, 3)
    ag = (L.Agg * 1)()
    ag[0].d_col, ag[0].col_type, ag[0].kind = v.data_ptr(), L.RFX_F64, L.RFX_AGG_SUM
    from flat_util import group_tables
    t, store, layout = group_tables(eng, ag, 1, 0, 1024, hashed=True)  # 1024 slots for 10 000 distinct keys
    L.check(eng.lib.rfx_hip_hash_tables_init(eng._ctx, ag, C.byref(t)))
    rc = eng.lib.rfx_hip_group_hash_accumulate(eng._ctx, k.data_ptr(), None, 0, L.RFX_AND, ag, n, 0, C.byref(t))
    assert rc == -5 and b"hash table full" in eng.lib.rfx_hip_last_error()
