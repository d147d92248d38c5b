// adi_sweep_contig_x.hip -- the FAST kernel of the contiguous-axis sweep (adi_contig_dev.hpp) with 20, 24 and 28 rows per lane:
// lines of 320 / 384 / 448 rows (16 segments) and 640 / 768 / 896 (32); 64 segments would be 1280 / 1536 / 1792 rows, beyond the
// kMaxFastLine = 1024 rows the in-register kernels are given, so that branch is never taken.  With 16 rows per lane such
// lines need 20 - 28 of 32 lanes of the in-wave interface solve and the rest of every wave is padding: 222 - 241
// Gcell/s against 324 - 333 with the exact fit (345 at 512 rows).  A translation unit of its own so that the build stays
// parallel (12 more instantiations of a 100 - 145 VGPR kernel).
#include "adi_contig_dev.hpp"

namespace adi {

template <int MF>
static void exact_t(bool has_dir, bool has_q, const double *in, const uint8_t *flags, const double *coeff, const uint8_t *dmask,
                    const double *dval, const double *qf, double *out, const Lay &L, SweepScal s, bool vec, long nunits_f,
                    unsigned *queue, hipStream_t st)
{
    if (has_dir && has_q) launch_contig_fast<MF, true, true>(in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
    else if (has_q) launch_contig_fast<MF, false, true>(in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
    else if (has_dir) launch_contig_fast<MF, true, false>(in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
    else launch_contig_fast<MF, false, false>(in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
}

void contig_fast_exact(int mf, bool has_dir, bool has_q, const double *in, const uint8_t *flags, const double *coeff,
                       const uint8_t *dmask, const double *dval, const double *qf, double *out, const Lay &L, SweepScal s,
                       bool vec, long nunits_f, unsigned *queue, hipStream_t st)
{
    if (mf == 20) exact_t<20>(has_dir, has_q, in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
    else if (mf == 24) exact_t<24>(has_dir, has_q, in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
    else exact_t<28>(has_dir, has_q, in, flags, coeff, dmask, dval, qf, out, L, s, vec, nunits_f, queue, st);
}

}  // namespace adi
