// adi_sweep_strided_x.hip -- the FAST kernel of the strided-axis sweeps (adi_strided_fast.hpp) with 20, 24 and 28 rows per
// thread: lines of 320 / 384 / 448 rows (16 segments, 256-thread workgroups) and 640 / 768 / 896 rows (32 segments).  With 16
// or 32 rows per thread those lines fill 20 - 28 of 32 segment slots and the rest of every workgroup is padding (strided_plan).
// A translation unit of its own so that the build stays parallel.
#include "adi_strided_fast.hpp"

namespace adi {

template <int MF>
static void exact_t(bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                    const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                    const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_dir && has_q) launch_strided_fast<MF, true, true, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_q) launch_strided_fast<MF, false, true, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_dir) launch_strided_fast<MF, true, false, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_fast<MF, false, false, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_fast_exact(int mf, bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                        const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                        const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                        const Fuse &fz)
{
    if (mf == 20) exact_t<20>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (mf == 24) exact_t<24>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else exact_t<28>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

}  // namespace adi
