// adi_sweep_strided_y.hip -- the FAST kernel of the strided-axis sweeps (adi_strided_fast.hpp) with 18, 22, 26 and 30 rows per
// thread: lines of 288 / 352 / 416 / 480 rows (16 segments) and 576 / 704 / 832 / 960 rows (32 segments); with
// adi_sweep_strided_x.hip (20 / 24 / 28 rows) every multiple of 32 rows from 256 to 512 is an exact fit of the unfused strided
// sweeps too -- the lengths the padded extents (adi_recommended_dims) round ragged lines up to.
#include "adi_strided_fast.hpp"

namespace adi {

template <int MF>
static void exact2_t(bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                    const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                    const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_dir && has_q) launch_strided_fast<MF, true, true, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_q) launch_strided_fast<MF, false, true, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (has_dir) launch_strided_fast<MF, true, false, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_fast<MF, false, false, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_fast_exact2(int mf, bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                        const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                        const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                        const Fuse &fz)
{
    if (mf == 18) exact2_t<18>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (mf == 22) exact2_t<22>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (mf == 26) exact2_t<26>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
    else exact2_t<30>(has_dir, has_q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
}

}  // namespace adi
