// rfx_group_part.hip -- radix-partitioned dense group-by for ranges whose tables do not fit one workgroup's LDS.
// (placeholder until the partitioned kernels land: reports "not applicable" so the caller uses device atomics)
#include "rfx_scalar_kernel.hpp"

int rfx_group_part_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t) {
    (void)c; (void)P; (void)key_idx; (void)t;
    return RFX_ESTATE;
}
