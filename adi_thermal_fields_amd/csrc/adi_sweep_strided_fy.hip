// adi_sweep_strided_fy.hip -- the FUSED explicit + axis-0 FAST kernel (adi_strided_fast.hpp, FUSE = true) with 9, 11, 13 and 15
// rows per thread: lines of 144 / 176 / 208 / 240 rows (16 segments) and 288 / 352 / 416 / 480 rows (32 segments); with
// adi_sweep_strided_fx.hip every multiple of 16 from 128 to 256 rows and every multiple of 32 from 256 to 512 is an exact fit,
// which is what the padded extents (adi_recommended_dims) round ragged lines up to.  See adi_sweep_strided_fx.hip:  The fused kernel holds
// at most 16 rows per thread (its loader hands the k-halo columns round the 16 lanes of a DPP row and it sits at the 128-VGPR
// budget), so with 16 rows those lines fill 10 - 14 of 16 (20 - 28 of 32) segment slots of every workgroup: 127 - 164 Gcell/s
// against 195 - 200 at 256 / 512 rows.  No Dirichlet cells (those tiles would go to the GENERAL kernel anyway); with and without
// the coefficients taken from the flags (FC).  A translation unit of its own so that the build stays parallel.
#include "adi_strided_fast.hpp"

namespace adi {

template <int MF, bool HAS_Q>
static void fy_t(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff, const double *qf,
                 double *out, const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue,
                 hipStream_t st, const Fuse &fz)
{
    if (s.fconst) launch_strided_fast_t<MF, false, HAS_Q, true, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_fast_t<MF, false, HAS_Q, true, false>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_fast_fused_exact_odd(int mf, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                              const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                              const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_q) {
        if (mf == 9) fy_t<9, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 11) fy_t<11, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 13) fy_t<13, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else fy_t<15, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
    } else {
        if (mf == 9) fy_t<9, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 11) fy_t<11, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 13) fy_t<13, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
        else fy_t<15, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
    }
}

}  // namespace adi
