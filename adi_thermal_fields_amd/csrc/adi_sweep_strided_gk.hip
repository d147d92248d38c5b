// adi_sweep_strided_gk.hip -- the GENERAL kernels of the strided-axis sweeps (adi_strided_general.hpp) that carry the deferred
// interface correction of a slab decomposition (adi_sweep_corrected, SweepScal::c_*: the axis-1 sweep of a slab adds the
// rank-two update of the sharded-axis solve to every value it loads).  Only these instantiations
// contain corr_apply; the ordinary sweeps of adi_sweep_strided.hip / _gc.hip are compiled without it.  8 / 16 rows per
// thread, never fused; the source of the coefficients is decided at run time here (a slab-only path).
#include "adi_strided_general.hpp"

namespace adi {

template <int M>
static void gk_m(bool has_dir, bool has_q, const StridedPlan &P, unsigned ggrid, const double *in, const uint8_t *flags,
                 const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                 const double *xlo, const double *xhi, const SweepScal &s, const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_dir && has_q) launch_strided_general_t<M, true, true, false, 0, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_q) launch_strided_general_t<M, false, true, false, 0, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_dir) launch_strided_general_t<M, true, false, false, 0, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_general_t<M, false, false, false, 0, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_general_corr(int m, bool has_dir, bool has_q, const StridedPlan &P, unsigned ggrid, const double *in,
                          const uint8_t *flags, const double *coeff, const uint8_t *dmask, const double *dval, const double *qf,
                          double *out, const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                          const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (m == 8) gk_m<8>(has_dir, has_q, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else gk_m<16>(has_dir, has_q, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

}  // namespace adi
