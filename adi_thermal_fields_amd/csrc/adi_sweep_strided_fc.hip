// adi_sweep_strided_fc.hip -- the FAST kernels of the strided-axis sweeps (adi_strided_fast.hpp) for packs built from per-face
// SCALARS (SweepScal::fconst, h_face_consts of include/adi_hip.h): the Robin coefficient / Neumann flux of a row exposed
// along the sweep axis follows from its flags byte, so these instantiations contain no load of coeff / qflux at all.  On a
// curved solid such a load can only be issued once the flags have arrived -- a second memory latency in every wave that
// holds a surface row -- and even its presence behind a run-time branch cost the 512^3 ellipsoid 0.02 ms per strided sweep.
// 8 / 16 / 32 rows per thread, no Dirichlet cells, with and without the explicit stage folded in; a translation unit of its
// own so that the build stays parallel.
#include "adi_strided_fast.hpp"

namespace adi {

template <bool HAS_Q>
static void fc_t(int mf, bool fuse, const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                 const double *qf, double *out, const LineGeom &g, const double *xlo, const double *xhi, SweepScal s,
                 unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (mf == 32) launch_strided_fast_t<32, false, HAS_Q, false, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
    else if (mf == 16) {
        if (fuse) launch_strided_fast_t<16, false, HAS_Q, true, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
        else launch_strided_fast_t<16, false, HAS_Q, false, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
    } else {
        if (fuse) launch_strided_fast_t<8, false, HAS_Q, true, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
        else launch_strided_fast_t<8, false, HAS_Q, false, true>(P, in, flags, coeff, nullptr, nullptr, qf, out, g, xlo, xhi, s, queue, st, fz);
    }
}

void strided_fast_fc(int mf, bool has_q, bool fuse, const StridedPlan &P, const double *in, const uint8_t *flags,
                     const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                     const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (has_q) fc_t<true>(mf, fuse, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
    else fc_t<false>(mf, fuse, P, in, flags, coeff, nullptr, out, g, xlo, xhi, s, queue, st, fz);
}

}  // namespace adi
