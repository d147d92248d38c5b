// adi_sweep_strided_fx.hip -- the FUSED explicit + axis-0 FAST kernel (adi_strided_fast.hpp, FUSE = true) with 10, 12 and 14
// rows per thread: lines of 160 / 192 / 224 rows (16 segments) and 320 / 384 / 448 rows (32 segments).  The fused kernel holds
// at most 16 rows per thread (its loader hands the k-halo columns round the 16 lanes of a DPP row and it sits at the 128-VGPR
// budget), so with 16 rows those lines fill 10 - 14 of 16 (20 - 28 of 32) segment slots of every workgroup: 127 - 164 Gcell/s
// against 195 - 200 at 256 / 512 rows.  No Dirichlet cells (those tiles would go to the GENERAL kernel anyway); with and without
// the coefficients taken from the flags (FC).  A translation unit of its own so that the build stays parallel.
#include "adi_strided_fast.hpp"

namespace adi {

template <int MF, bool HAS_Q>
static void fx_t(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff, const double *qf,
                 double *out, const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue,
                 hipStream_t st, const Fuse &fz)
{
    if (s.fconst) launch_strided_fast_t<MF, false, HAS_Q, true, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
    else launch_strided_fast_t<MF, false, HAS_Q, true, false>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
}

void strided_fast_fused_exact(int mf, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                              const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                              const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_q) {
        if (mf == 10) fx_t<10, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 12) fx_t<12, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else fx_t<14, true>(P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
    } else {
        if (mf == 10) fx_t<10, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
        else if (mf == 12) fx_t<12, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
        else fx_t<14, false>(P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
    }
}

}  // namespace adi
