// Test hooks declared in include/hqtick_debug.h: host-side building blocks that can be unit-tested without a GPU.
// They are NOT a CPU implementation of the tick (hqtick_run has none): only the exact MILP solver and the
// hashbrown-order helper are reachable here.
#include "../../include/hqtick_debug.h"

#include "hb_order.h"
#include "milp.h"
#include "price_emul.h"

static thread_local int g_last_canonical = 0;
extern "C" int hqtick_debug_milp_was_canonical(void) { return g_last_canonical; }

extern "C" int hqtick_debug_milp_solve(int ncols, const double *obj, const uint8_t *col_kind, int nrows, const uint8_t *row_type,
                                       const double *rhs, const int *row_off, const int *row_col, const double *row_coef,
                                       double time_limit_s, int canonical, double *x_out, double *obj_out, int *is_optimal,
                                       long *nodes_out) {
    hqmilp::Model m;
    m.obj.assign(obj, obj + ncols);
    m.kind.assign(col_kind, col_kind + ncols);
    m.rtype.assign(row_type, row_type + nrows);
    m.rhs.assign(rhs, rhs + nrows);
    m.roff.assign(row_off, row_off + nrows + 1);
    int nnz = nrows ? row_off[nrows] : 0;
    m.rcol.assign(row_col, row_col + nnz);
    m.rcoef.assign(row_coef, row_coef + nnz);
    hqmilp::Result r = hqmilp::solve(m, time_limit_s, canonical != 0);
    g_last_canonical = r.canonical ? 1 : 0;
    if (nodes_out) *nodes_out = r.nodes;
    if (!r.feasible) return 0;
    for (int j = 0; j < ncols; j++) x_out[j] = r.x[j];
    *obj_out = r.objective;
    *is_optimal = r.optimal ? 1 : 0;
    return 1;
}

extern "C" void hqtick_debug_map_order_u32(const uint32_t *keys, uint32_t n, uint32_t *out_pos) {
    std::vector<uint32_t> order;
    hqhb::insertion_order_u32(keys, n, order);
    for (uint32_t i = 0; i < n; i++) out_pos[i] = order[i];
}

// The solver on a model that carries the builder's structure hints, with the price sweeps of the coupled solve run through the emulated wavefront
// (csrc/price_emul.cpp): the CPU tests' way into csrc/price.cpp + csrc/price_core.h.  col_group / row_implied may be null; min_cols = smallest
// component the sweeps take (0: the default).  stats_out[4] (optional): sweeps, rounds, certified-by-sweeps flag (sweeps > 0 and optimal), canonical.
extern "C" int hqtick_debug_milp_solve_priced(int ncols, const double *obj, const uint8_t *col_kind, const int32_t *col_group, int nrows, const uint8_t *row_type,
                                              const uint8_t *row_implied, const double *rhs, const int *row_off, const int *row_col, const double *row_coef,
                                              double time_limit_s, int use_sweeps, uint32_t min_cols, double *x_out, double *obj_out, int *is_optimal, double *stats_out) {
    hqmilp::Model m;
    m.obj.assign(obj, obj + ncols);
    m.kind.assign(col_kind, col_kind + ncols);
    m.rtype.assign(row_type, row_type + nrows);
    m.rhs.assign(rhs, rhs + nrows);
    m.roff.assign(row_off, row_off + nrows + 1);
    int nnz = nrows ? row_off[nrows] : 0;
    m.rcol.assign(row_col, row_col + nnz);
    m.rcoef.assign(row_coef, row_coef + nnz);
    if (col_group) m.col_group.assign(col_group, col_group + ncols);
    if (row_implied) m.row_implied.assign(row_implied, row_implied + nrows);
    hqprice::EmulatedSweeper emu;
    if (min_cols) emu.min_cols = min_cols;
    hqmilp::Result r = hqmilp::solve(m, time_limit_s, true, hqmilp::REFERENCE_MIP_REL_GAP, use_sweeps ? &emu : nullptr);
    g_last_canonical = r.canonical ? 1 : 0;
    if (stats_out) { stats_out[0] = r.price_sweeps; stats_out[1] = r.price_rounds; stats_out[2] = r.price_total_us; stats_out[3] = r.canonical ? 1.0 : 0.0; }
    if (!r.feasible) return 0;
    for (int j = 0; j < ncols; j++) x_out[j] = r.x[j];
    *obj_out = r.objective;
    *is_optimal = r.optimal ? 1 : 0;
    return 1;
}
