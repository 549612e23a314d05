// rfx_scalar_nc.hip -- instantiates k_filter_aggr for ONE distinct-column count (RFX_NC) and the 4 aggregate-slot
// buckets.  Compiled 8 times (RFX_NC = 1..8) so the instantiations build in parallel.
#include "rfx_scalar_kernel.hpp"
#ifndef RFX_NC
#error "compile with -DRFX_NC=<1..8>"
#endif

// Fast shapes (<= 4 distinct columns, <= 4 predicates, <= 4 aggregates) get tight instantiations; everything else
// runs the one fully general instantiation of its column count.
template <int NC, int NA, int NP, int NX = 0, bool DEEP = false>
static void launch_shape(rfx_ctx *c, const Plan &P, int grid, Acc *ws) {
    // 16-byte loads in flight per lane = NC * U.  In-process sweep on MI355X (bench.py --ab): U = 4 wins
    // for 1..4 columns (C2 6.77, C2b 6.92, C5 6.10 TB/s); the matching workgroups-per-CU choice is rfx_scalar_grid().
    constexpr int U = (NC <= 4) ? 4 : (NC <= 6) ? 2 : 1;
    hipLaunchKernelGGL((k_filter_aggr<NC, NA, U, NP, NX, DEEP>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, ws);
}

#define RFX_CAT2(a, b) a##b
#define RFX_CAT(a, b) RFX_CAT2(a, b)
int RFX_CAT(rfx_launch_filter_aggr_nc, RFX_NC)(rfx_ctx *c, const Plan &P, int grid, Acc *ws, int *na_stride) {
    if (P.nx > 0) { // aggregates over element-wise expressions, folded on the fly
        bool deep = false; // any expression tree (more than one operation)?  Those take the general evaluator's instantiation.
        for (int i = 0; i < P.nx; i++) deep |= P.xs[i].nops > 1;
        if (deep) {
            *na_stride = 9;
            launch_shape<RFX_NC, 8, 8, RFX_MAX_EXPRS, true>(c, P, grid, ws);
            return RFX_OK;
        }
#if RFX_NC <= 4
        if (P.npred <= 4 && P.nagg <= 4) {
            if (P.nx == 1 && P.nagg == 1) { // one aggregate over one expression (TPC-H Q6 itself): one accumulator, not four
                *na_stride = 2;
                launch_shape<RFX_NC, 1, 4, 1>(c, P, grid, ws);
                return RFX_OK;
            }
            *na_stride = 5;
            if (P.nx == 1) launch_shape<RFX_NC, 4, 4, 1>(c, P, grid, ws); // the common one-expression query (TPC-H Q6 shape)
            else launch_shape<RFX_NC, 4, 4, RFX_MAX_EXPRS>(c, P, grid, ws);
            return RFX_OK;
        }
#endif
        *na_stride = 9;
        launch_shape<RFX_NC, 8, 8, RFX_MAX_EXPRS>(c, P, grid, ws);
        return RFX_OK;
    }
#if RFX_NC <= 4
    if (P.npred <= 4 && P.nagg <= 4) {
        if (P.nagg <= 1) {
            *na_stride = 2;
            if (P.npred <= 1) launch_shape<RFX_NC, 1, 1>(c, P, grid, ws);
            else launch_shape<RFX_NC, 1, 4>(c, P, grid, ws);
        } else {
            *na_stride = 5;
            if (P.npred <= 1) launch_shape<RFX_NC, 4, 1>(c, P, grid, ws);
            else launch_shape<RFX_NC, 4, 4>(c, P, grid, ws);
        }
        return RFX_OK;
    }
#endif
    *na_stride = 9;
    launch_shape<RFX_NC, 8, 8>(c, P, grid, ws);
    return RFX_OK;
}
