/* tests/golden/hash_index_harness.c -- FIXTURE GENERATION ONLY (build container; never compiled into the product or the tests).
 * hash_index_u64 is `inline` in the reference's core/hash.h, so the compiled reference exports no symbol for it: this harness
 * includes the reference's OWN header where it lies (-I /root/reference/core) and evaluates it on (seed, key) pairs read from
 * stdin (binary u64 pairs), writing the hashes to stdout (binary u64).  tests/golden/make_hash_index_golden.py drives it and
 * stores the vectors in tests/golden/ref_golden.npz (arrays hash_index_u64, hash_index_u64_seeded). */
#include <stdio.h>
#include "hash.h"

int main(void) {
    unsigned long long in[2];
    while (fread(in, sizeof(in), 1, stdin) == 1) {
        unsigned long long out = (unsigned long long)hash_index_u64((u64_t)in[0], (u64_t)in[1]);
        fwrite(&out, sizeof(out), 1, stdout);
    }
    return 0;
}
