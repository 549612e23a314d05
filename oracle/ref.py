"""oracle/ref.py -- drive the REAL reference (compiled into oracle/_ref/ by `make -C oracle ref`).

*** TEST INFRASTRUCTURE ONLY ***  Used (a) in the build container to capture golden vectors and to validate the C
restatement, (b) on the GPU box -- where only the prebuilt oracle/_ref/rayforce travels -- as the "reference" CPU
baseline of bench.py and for the drop-in plugin test.  Nothing here reads /root/reference at run time.

Data goes in and out through the reference's own column-file format (core/binary.c:263-311 writer, core/unary.c:48-136
reader): 16-byte header {mmod=0xfd, order=0, type, attrs=0, rc=0, len:i64} + raw little-endian payload.
"""
from __future__ import annotations

import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(_HERE, "_ref", "rayforce")
_TYPES = {np.dtype(np.int8): 1, np.dtype(np.bool_): 1, np.dtype(np.int32): 4, np.dtype(np.int64): 5, np.dtype(np.float64): 10}
_DTYPES = {1: np.int8, 4: np.int32, 5: np.int64, 7: np.int32, 8: np.int32, 10: np.float64, 9: np.int64, 6: np.int64}
LAST_STDERR = ""  # of the last run_script (a plugin loaded into the reference may trace there)


def available() -> bool:
    return os.path.exists(BIN) and os.access(BIN, os.X_OK)


def build(ref_root: str = "/root/reference") -> bool:
    """Compile the reference from its own sources (only possible where they are mounted)."""
    if not os.path.isdir(os.path.join(ref_root, "core")):
        return available()
    subprocess.run(["make", "-s", "-C", _HERE, "ref", f"REF={ref_root}", "-j8"], check=True, stdout=subprocess.DEVNULL)
    return available()


def write_col(path: str, a: np.ndarray, tp: int | None = None) -> None:
    """tp: the reference's vector type code when the dtype does not say it (int32 payloads: 4 I32, 7 DATE = days, 8 TIME = milliseconds)."""
    a = np.ascontiguousarray(a)
    with open(path, "wb") as f:
        f.write(struct.pack("<BBbBIq", 0xFD, 0, _TYPES[a.dtype] if tp is None else tp, 0, 0, a.size))
        f.write(a.tobytes())


def read_col(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        hdr = f.read(16)
        _, _, tp, _, _, n = struct.unpack("<BBbBIq", hdr)
        if tp < 0:  # atom written as a 1-element payload of the header union
            raise ValueError("atom file")
        return np.frombuffer(f.read(), dtype=_DTYPES[tp], count=n).copy()


def run_script(text: str, threads: int | None = None, timeout: float = 600.0, cwd: str | None = None) -> str:
    if not available():
        raise RuntimeError("oracle/_ref/rayforce is not built (run `make -C oracle ref` where /root/reference is mounted)")
    with tempfile.NamedTemporaryFile("w", suffix=".rfl", delete=False) as f:
        f.write(text)
        script = f.name
    try:
        cmd = [BIN, "-f", script]
        if threads:
            cmd += ["-c", str(threads)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=cwd, stdin=subprocess.DEVNULL)
        if p.returncode != 0:
            raise RuntimeError(f"reference exited with {p.returncode}: {p.stdout[-2000:]} {p.stderr[-2000:]}")
        global LAST_STDERR
        LAST_STDERR = p.stderr
        return p.stdout
    finally:
        os.unlink(script)


class Session:
    """A scratch directory of column files + a Rayfall script builder."""

    def __init__(self, root: str | None = None):
        self.dir = tempfile.mkdtemp(prefix="rfref_", dir=root)
        self.lines: list[str] = []
        self.outs: list[str] = []

    def close(self):
        shutil.rmtree(self.dir, ignore_errors=True)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def put(self, name: str, a: np.ndarray, tp: int | None = None) -> None:
        path = os.path.join(self.dir, f"in_{name}")
        write_col(path, a, tp)
        self.lines.append(f'(set {name} (get "{path}"))')

    def table(self, tname: str, cols: dict) -> None:
        for k, v in cols.items():
            self.put(k, v)
        names = " ".join(cols.keys())
        self.lines.append(f"(set {tname} (table [{names}] (list {names})))")

    def eval(self, expr: str) -> None:
        self.lines.append(expr)

    def out(self, name: str, expr: str) -> None:
        """Write the VECTOR value of expr to a column file; read back by run()."""
        path = os.path.join(self.dir, f"out_{name}")
        self.lines.append(f'(set "{path}" {expr})')
        self.outs.append(name)

    def run(self, threads: int | None = None, timeout: float = 600.0) -> dict:
        stdout = run_script("\n".join(self.lines) + "\n", threads=threads, timeout=timeout)
        res = {"_stdout": stdout}
        for name in self.outs:
            res[name] = read_col(os.path.join(self.dir, f"out_{name}"))
        return res
