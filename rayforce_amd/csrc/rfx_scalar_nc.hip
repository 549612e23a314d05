// rfx_scalar_nc.hip -- instantiates k_filter_aggr for ONE distinct-column count (RFX_NC) and the 4 aggregate-slot
// buckets.  Compiled 8 times (RFX_NC = 1..8) so the instantiations build in parallel.
#include "rfx_scalar_kernel.hpp"
#ifndef RFX_NC
#error "compile with -DRFX_NC=<1..8>"
#endif

template <int NC, int NA>
static void launch_filter_aggr(rfx_ctx *c, const Plan &P, int grid, Acc *ws) {
    // loads in flight per lane = NC * U (16 B each); probe_hw: U=4 with nt loads is the streaming sweet spot
    constexpr int U = (NC <= 2) ? 4 : (NC <= 4 ? 2 : 1);
    hipLaunchKernelGGL((k_filter_aggr<NC, NA, U>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, ws);
}

#define RFX_CAT2(a, b) a##b
#define RFX_CAT(a, b) RFX_CAT2(a, b)
int RFX_CAT(rfx_launch_filter_aggr_nc, RFX_NC)(rfx_ctx *c, const Plan &P, int grid, Acc *ws, int *na_stride) {
    if (P.nagg <= 1) { *na_stride = 2; launch_filter_aggr<RFX_NC, 1>(c, P, grid, ws); }
    else if (P.nagg <= 2) { *na_stride = 3; launch_filter_aggr<RFX_NC, 2>(c, P, grid, ws); }
    else if (P.nagg <= 4) { *na_stride = 5; launch_filter_aggr<RFX_NC, 4>(c, P, grid, ws); }
    else { *na_stride = 9; launch_filter_aggr<RFX_NC, 8>(c, P, grid, ws); }
    return RFX_OK;
}
