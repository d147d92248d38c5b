// adi_sweep_strided_gc.hip -- the GENERAL kernels of the strided-axis sweeps (adi_strided_general.hpp) for packs built from
// per-face SCALARS (SweepScal::fconst, h_face_consts of include/adi_hip.h): the Robin coefficient / Neumann flux of a row
// exposed along the sweep axis follows from its flags byte, and these instantiations contain no load of coeff / qflux at
// all.  8 / 16 rows per thread, with and without the explicit stage folded in (they drain what the FAST kernels of
// adi_sweep_strided_fc.hip queue: surface tiles, Dirichlet cells).  A translation unit of its own: the build stays parallel.
#include "adi_strided_general.hpp"

namespace adi {

template <int M, bool HAS_DIR, bool HAS_Q>
static void gc_t(bool fuse, const StridedPlan &P, unsigned ggrid, const double *in, const uint8_t *flags, const double *coeff,
                 const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g, const double *xlo,
                 const double *xhi, const SweepScal &s, const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (fuse) launch_strided_general_t<M, HAS_DIR, HAS_Q, true, 1, false>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_general_t<M, HAS_DIR, HAS_Q, false, 1, false>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

template <int M>
static void gc_m(bool has_dir, bool has_q, bool fuse, const StridedPlan &P, unsigned ggrid, const double *in, const uint8_t *flags,
                 const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                 const double *xlo, const double *xhi, const SweepScal &s, const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_dir && has_q) gc_t<M, true, true>(fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_q) gc_t<M, false, true>(fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_dir) gc_t<M, true, false>(fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else gc_t<M, false, false>(fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_general_fc(int m, bool has_dir, bool has_q, bool fuse, const StridedPlan &P, unsigned ggrid, const double *in,
                        const uint8_t *flags, const double *coeff, const uint8_t *dmask, const double *dval, const double *qf,
                        double *out, const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                        const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (m == 8) gc_m<8>(has_dir, has_q, fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else gc_m<16>(has_dir, has_q, fuse, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

}  // namespace adi
