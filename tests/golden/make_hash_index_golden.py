#!/usr/bin/env python
"""Pin hash_index_u64 (core/hash.h:86-97, `inline`: no symbol in the compiled reference) -- build container only.

    python tests/golden/make_hash_index_golden.py

Compiles tests/golden/hash_index_harness.c against the reference's own header, runs it on the fixture's hash keys with the
reference's seed (U64_HASH_SEED) and with chained seeds (the row-hash use: h = hash_index_u64(h, key) column after column), and
adds the arrays hash_index_u64 / hash_index_u64_seeds / hash_index_u64_seeded to tests/golden/ref_golden.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/core"


def main():
    npz = os.path.join(HERE, "ref_golden.npz")
    arrays = dict(np.load(npz))
    keys = arrays["hash_keys"].astype(np.int64).view(np.uint64)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "hih")
        subprocess.run(["gcc", "-O2", "-std=c17", "-w", "-I", REF, "-include", os.path.join(REF, "def.h"), os.path.join(HERE, "hash_index_harness.c"), "-o", exe],
                       check=True)

        def run(seeds):
            pairs = np.stack([seeds, keys], axis=1).astype(np.uint64).tobytes()
            out = subprocess.run([exe], input=pairs, capture_output=True, check=True).stdout
            return np.frombuffer(out, dtype=np.uint64).copy()

        seed = np.full(keys.shape, 0x9DDFEA08EB382D69, np.uint64)
        h1 = run(seed)
        arrays["hash_index_u64"] = h1
        arrays["hash_index_u64_seeds"] = h1            # second link of the chain: the previous column's hash is the seed
        arrays["hash_index_u64_seeded"] = run(h1)
    np.savez_compressed(npz, **arrays)
    print(f"hash_index_u64: {len(keys)} vectors, first {h1[0]:#x}; chained first {arrays['hash_index_u64_seeded'][0]:#x}")


if __name__ == "__main__":
    sys.exit(main())
