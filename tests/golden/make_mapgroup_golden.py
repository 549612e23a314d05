#!/usr/bin/env python
"""Golden group INDEXES and grouped aggregates from the compiled reference library -- build container only.

    python tests/golden/make_mapgroup_golden.py     # writes tests/golden/mapgroup_golden.npz

The 7-slot group index (index_group_build, core/index.c:1696-1699) is what an FN_AGGR built-in receives inside a lazy
TYPE_MAPGROUP pair (core/group.c:26-46).  This script calls the reference's own index_group(keys, filter) and
aggr_{sum,min,max,avg,count,first}(val, index) (core/aggr.c) through ctypes on oracle/_ref/librayforce_ref.so and stores, per case:
inputs (keys, values, optional filter ids), the index's slots (type, group count, group ids / key table, shift, first ids) and
every aggregate's result.  Both index flavours occur: SHIFT (range <= 524 288: key table + source column) and IDS (per-row ids).
The fixture is data only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rfo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
T_I64, T_F64, T_LIST = 5, 10, 0
NULL = -(2**63)


class Obj(C.Structure):
    _fields_ = [("mmod", C.c_uint8), ("order", C.c_uint8), ("type", C.c_int8), ("attrs", C.c_uint8), ("rc", C.c_uint32), ("len", C.c_int64)]


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librayforce_ref.so"))
    lib.ray_init.restype = C.c_int32
    assert lib.ray_init() == 0
    lib.vector.restype = C.c_void_p
    lib.vector.argtypes = [C.c_int8, C.c_int64]
    for f in ("index_group", "aggr_sum", "aggr_min", "aggr_max", "aggr_avg", "aggr_count", "aggr_first"):
        getattr(lib, f).restype = C.c_void_p
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p]
    null_obj = C.addressof(Obj.in_dll(lib, "__NULL_OBJ"))

    def vec(a):
        a = np.ascontiguousarray(a)
        o = lib.vector(T_F64 if a.dtype == np.float64 else T_I64, a.size)
        C.memmove(o + 16, a.ctypes.data, a.nbytes)
        return o

    def arr(o):
        h = Obj.from_address(o)
        if h.type not in (T_I64, T_F64):
            return None
        dt = {T_I64: np.int64, T_F64: np.float64}[h.type]
        return np.frombuffer((C.c_char * (h.len * 8)).from_address(o + 16), dtype=dt).copy()

    def slot(index, i):
        return C.c_void_p.from_address(index + 16 + 8 * i).value

    def atom_i64(o):
        return C.c_int64.from_address(o + 8).value

    import hashlib

    def digest(a):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()

    arrays, cases = {}, []
    # The fixture stays small: inputs are (generator, seed) pairs (tests/golden_cases.py: mapgroup_inputs rebuilds them), the index's
    # per-row / per-slot arrays and every INTEGER result are stored as sha-256 digests (bit-exact comparison needs no more), f64
    # results as a sample (the first 512 groups and every 61st).
    # (sizes that keep clear of the reference's index_scope_i64 chunking defect, DESIGN.md "Reference defects observed")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_cases import MAPGROUP_CASES, mapgroup_inputs, mapgroup_sample
    for ci, (n, keys, off, filt) in enumerate(MAPGROUP_CASES):
        k, vi, vf, ids = mapgroup_inputs(ci)
        ko, fo = vec(k), (vec(ids) if filt else null_obj)
        index = lib.index_group(ko, fo)
        assert Obj.from_address(index).type == T_LIST and Obj.from_address(index).len == 7
        itype, groups = atom_i64(slot(index, 0)), atom_i64(slot(index, 1))
        pre = f"mg{ci}_"
        arrays[pre + "group_ids_sha"] = digest(arr(slot(index, 2)))
        firsts = arr(slot(index, 6))
        if firsts is not None:
            arrays[pre + "first_ids_sha"] = digest(firsts)
        shift = atom_i64(slot(index, 3)) if itype == 1 else NULL
        for col, cv, isf in (("vi", vec(vi), False), ("vf", vec(vf), True)):
            for fn in ("sum", "min", "max", "avg", "count", "first"):
                if fn == "first" and firsts is None:
                    continue
                r = arr(getattr(lib, "aggr_" + fn)(cv, index))
                assert len(r) == groups
                if r.dtype == np.float64:
                    arrays[f"{pre}{fn}_{col}_sample"] = r[mapgroup_sample(groups)]
                else:
                    arrays[f"{pre}{fn}_{col}_sha"] = digest(r)
        cases.append({"case": ci, "n": n, "index_type": int(itype), "groups": int(groups), "shift": int(shift), "filtered": bool(filt), "firsts": firsts is not None})
        print(cases[-1])
    arrays["cases"] = np.array([[c["case"], c["n"], c["index_type"], c["groups"], c["shift"], int(c["filtered"]), int(c["firsts"])] for c in cases], np.int64)
    np.savez_compressed(os.path.join(HERE, "mapgroup_golden.npz"), **arrays)
    print("wrote", len(arrays), "arrays")


if __name__ == "__main__":
    main()
