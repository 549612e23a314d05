"""RayforceDB column files -> device columns: single column files (core/binary.c:263-311), splayed tables
(core/io.c:1194-1364) and the get-parted layout (core/vary.c:185-392), moved with the library's pipelined pinned-staging copy.
Host-side directory walking only; the bytes go through rfx_column_file_stat / rfx_hip_column_file_load."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib as L
from ._lib import RfxError

# ------------------------------------------------------------------ on-disk columns
_FILE_DTYPES = {5: torch.int64, 6: torch.int64, 9: torch.int64, 10: torch.float64}  # i64, symbol ids, timestamp, f64

def load_column(eng, path: str) -> torch.Tensor:
    """A RayforceDB column file (core/binary.c:263-311) -> device column, moved with the pipelined pinned-staging path."""
    t, n = C.c_int32(), C.c_int64()
    L.check(eng.lib.rfx_column_file_stat(path.encode(), C.byref(t), C.byref(n)), "column_file_stat")
    out = torch.empty(n.value, dtype=_FILE_DTYPES[t.value], device=eng.device)
    L.check(eng.lib.rfx_hip_column_file_load(eng._ctx, path.encode(), out.data_ptr(), n.value), "column_file_load")
    return out

def load_splayed(eng, directory: str, columns: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
    """A splayed table (core/io.c:1194-1364): `<dir>/.d` is the serialised symbol vector of the column names, every
    column is its own file.  Loads the 8-byte columns (all, or the ones asked for) into HBM."""
    import os
    names = _splayed_names(directory)
    want = list(columns) if columns is not None else names
    missing = [c for c in want if c not in names]
    if missing:
        raise RfxError(f"no such column(s) in {directory}: {missing}")
    return {c: load_column(eng, os.path.join(directory, c)) for c in want}

def load_parted(eng, root: str, table: str, columns: Optional[Sequence[str]] = None, where=None) -> Dict[str, torch.Tensor]:
    """A parted table, `(get-parted root 'table)` (core/vary.c:185-392): `<root>/<YYYY.MM.DD>/<table>/` is one splayed
    table per date (a `sym` entry beside them is skipped), partitions in ascending date order, all with the same columns;
    the result has the virtual `Date` column first (days since 2000.01.01, core/date.c:124-135, as i64 here) and every
    8-byte column concatenated over the partitions.

    `where` -- a comparison on `Date`, or and / or of such: ("==", "Date", "2024.01.02"), dates as text or day numbers --
    is the reference's partition pruning (cmp_map on the MAPCOMMON column, core/cmp.c:341-358): it is evaluated on the
    directory list, and a partition that fails it is never opened, let alone uploaded."""
    import datetime
    import os
    epoch = datetime.date(2000, 1, 1)

    def days(x) -> int:
        if isinstance(x, int):
            return x
        y, m, d = (int(p) for p in str(x).split("."))
        return (datetime.date(y, m, d) - epoch).days

    parts = []
    for name in os.listdir(root):
        if name == "sym":
            continue
        try:
            parts.append((days(name), name))
        except (ValueError, TypeError):
            raise RfxError(f"{root}: partition directory {name!r} is not a date (YYYY.MM.DD)") from None
    parts.sort()
    if not parts:
        raise RfxError(f"{root}: no partitions")

    def keep(p, d) -> bool:
        if p[0] in ("and", "or"):
            r = [keep(q, d) for q in p[1:]]
            return all(r) if p[0] == "and" else any(r)
        op, lhs, rhs = p
        if lhs != "Date":
            raise RfxError("load_parted prunes on the virtual Date column only; filter other columns in the query")
        c = days(rhs)
        return {"==": d == c, "!=": d != c, "<": d < c, ">": d > c, "<=": d <= c, ">=": d >= c}[op]

    kept = [(d, nm) for d, nm in parts if where is None or keep(where, d)]
    first_dir = os.path.join(root, (kept or parts)[0][1], table)  # schema: first partition that is read at all
    names = list(_splayed_names(first_dir))
    want = list(columns) if columns is not None else names
    missing = [c for c in want if c not in names]
    if missing:
        raise RfxError(f"no such column(s) in {first_dir}: {missing}")
    # lengths and types from the headers only, then one device column per name and every file straight into its slice
    lens, types = [], {}
    for d, nm in kept:
        n_here = None
        for c in want:
            t, n = C.c_int32(), C.c_int64()
            L.check(eng.lib.rfx_column_file_stat(os.path.join(root, nm, table, c).encode(), C.byref(t), C.byref(n)), "column_file_stat")
            if types.setdefault(c, t.value) != t.value:
                raise RfxError(f"column {c} changes type between partitions")
            if n_here is not None and n.value != n_here:
                raise RfxError(f"columns of partition {nm} differ in length")
            n_here = n.value
        lens.append(n_here or 0)
    total = sum(lens)
    out = {"Date": eng.empty(total)}
    for c in want:
        out[c] = torch.empty(total, dtype=_FILE_DTYPES[types[c]] if c in types else torch.int64, device=eng.device)
    row = 0
    for (d, nm), n_here in zip(kept, lens):
        if n_here:
            out["Date"][row:row + n_here].fill_(d)  # plumbing: a constant per partition
            for c in want:
                L.check(eng.lib.rfx_hip_column_file_load(eng._ctx, os.path.join(root, nm, table, c).encode(),
                                                           out[c].data_ptr() + row * 8, n_here), "column_file_load")
        row += n_here
    return out

def _splayed_names(directory: str):
    import os
    raw = open(os.path.join(directory, ".d"), "rb").read()
    # serialised object: 16-byte IPC header (magic fa de fa ce, version, payload size), then type, attrs, len:i64, strings
    if len(raw) < 26 or raw[:4] != bytes.fromhex("fadeface") or raw[16] != 6:
        raise RfxError(f"{directory}/.d is not a serialised symbol vector")
    cnt = int.from_bytes(raw[18:26], "little")
    return [b.decode() for b in raw[26:].split(b"\0")[:cnt]]

