// adi_sweep_strided.hip -- K2: the batched tridiagonal sweeps along the strided axes (memory axes 0 and 1): sweep_axis0/1
// + thomas_solve of adi3d_numba_coeff.py:133-202, :121-130 (identity-row form of adi3d_gpu_coeff.py:154-191), the form
// with the explicit stage (lap1D_x/y/z + R0, :240-298) folded into the loads of the axis-0 sweep, and K4, the
// thread-per-line sweep for lines beyond 1024 rows.  Hand-written HIP for gfx950; HBM-bound, no MFMA.
#include "adi_strided_general.hpp"

namespace adi {

// ------------------------------------------------------------------------------------------------
// K4: generic fallback, one thread per line, normalised Thomas (adi3d_gpu_coeff.py:140-152) with the
// forward-pass c', d' kept in an HBM workspace.  Used only for lines longer than kMaxFastLine rows.
// ------------------------------------------------------------------------------------------------
template <bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_sweep_generic(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, long inner_stride, const double *__restrict__ xlo,
    const double *__restrict__ xhi, double *__restrict__ wc, double *__restrict__ wd, SweepScal s)
{
    const long lid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= (long)g.n_inner * g.n_outer) return;
    const long o = lid / g.n_inner, kc = lid - o * g.n_inner;
    const long base = o * g.outer_stride + kc * inner_stride;
    const int n = g.n;
    double cp = 0.0, dp = 0.0;
    const double2 cw = corr_weights(s, o);            // deferred interface correction (SweepScal::c_*), off in ordinary sweeps
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * g.stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        double vin = in[p];
        const long cell = (long)r * g.stride + kc * inner_stride;
        if (cw.x != 0.0) vin = __builtin_fma(cw.x, s.c_lo[cell], vin);
        if (cw.y != 0.0) vin = __builtin_fma(cw.y, s.c_hi[cell], vin);
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> g.lbit) & 1u, (f >> (g.lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0, vin,
                                     coeff[p], HAS_DIR ? dval[p] : 0.0, HAS_Q ? qf[p] : 0.0, s, a, b, c, d);
        if (r == 0) { if (xlo != nullptr) d -= a * xlo[lid]; a = 0.0; }
        if (r == n - 1) { if (xhi != nullptr) d -= c * xhi[lid]; c = 0.0; }
        const double inv = 1.0 / (b - a * cp);
        cp = c * inv;
        dp = (d - a * dp) * inv;
        wc[p] = cp;
        wd[p] = dp;
    }
    double x = 0.0;
    for (int r = n - 1; r >= 0; --r) {
        const long p = base + (long)r * g.stride;
        x = wd[p] - wc[p] * x;
        out[p] = x;
    }
}


// ------------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------------

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_strided(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                           const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                           const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                           const Fuse &fz)
{
    unsigned ggrid = (unsigned)P.ntiles_g;
    if (queue != nullptr) {
        const bool nofb = s.nofb != 0;                 // promise: no tile will be queued (see SweepScal)
        if (nofb) queue = nullptr;
        else (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        if (P.Mf == 20 || P.Mf == 24 || P.Mf == 28)      // exact fits: instantiated in adi_sweep_strided_x.hip
            strided_fast_exact(P.Mf, HAS_DIR, HAS_Q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 18 || P.Mf == 22 || P.Mf == 26 || P.Mf == 30)   // ... and adi_sweep_strided_y.hip
            strided_fast_exact2(P.Mf, HAS_DIR, HAS_Q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 10 || P.Mf == 12 || P.Mf == 14)  // exact fits of the fused kernel (no Dirichlet cells): adi_sweep_strided_fx.hip
            strided_fast_fused_exact(P.Mf, HAS_Q, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 9 || P.Mf == 11 || P.Mf == 13 || P.Mf == 15)   // ... and the odd row counts: adi_sweep_strided_fy.hip
            strided_fast_fused_exact_odd(P.Mf, HAS_Q, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 32) launch_strided_fast<32, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 16) launch_strided_fast<16, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else launch_strided_fast<8, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        ggrid = P.ntiles_g < 1024 ? (unsigned)P.ntiles_g : 1024u;
        if (nofb) return;
    }
    // 8 / 16 rows per thread: the source of the coefficients and the deferred correction are compiled in or out (see
    // adi_strided_general.hpp); shorter segments decide both at run time
    if constexpr (M >= 8) {
        if (!FUSE && s.c_w != nullptr)
            strided_general_corr(M, HAS_DIR, HAS_Q, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (s.fconst)
            strided_general_fc(M, HAS_DIR, HAS_Q, FUSE, P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else
            launch_strided_general_t<M, HAS_DIR, HAS_Q, FUSE, 2, false>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi,
                                                                        s, queue, st, fz);
    } else {
        launch_strided_general_t<M, HAS_DIR, HAS_Q, FUSE, 0, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s,
                                                                   queue, st, fz);
    }
}

template <bool HAS_DIR, bool HAS_Q>
static void strided_sweep_t(const SweepArgs &a, const Lay &L, const LineGeom &g, const SweepScal &s, double *out,
                            const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                            const Fuse *fzp)
{
    StridedPlan P = strided_plan(g, s.sparse != 0 && work != nullptr, false, fzp != nullptr, fzp != nullptr && !HAS_DIR);
    unsigned *queue = nullptr;
    if (P.Mf && use_fast(s, work, work_bytes, P.ntiles_f)) queue = (unsigned *)work;
    else if (P.Mf) P = strided_plan(g, false, false);
    if (fzp != nullptr) {
        Fuse fz = *fzp;
        if (queue != nullptr && !fuse_fast_ok(P, L, fz)) { queue = nullptr; P = strided_plan(g, false, false); }
        fuse_tile_order(fz, P, L);
        switch (P.Mg) {
            case 2: launch_strided<2, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 4: launch_strided<4, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 8: launch_strided<8, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            default: launch_strided<16, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        }
        return;
    }
    const Fuse fz = Fuse();
    switch (P.Mg) {
        case 2: launch_strided<2, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        case 4: launch_strided<4, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        case 8: launch_strided<8, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        default: launch_strided<16, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
    }
}

void strided_sweep(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const LineGeom &g, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                   const Fuse *fz)
{
    if (has_dir && has_q) strided_sweep_t<true, true>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else if (has_q) strided_sweep_t<false, true>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else if (has_dir) strided_sweep_t<true, false>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else strided_sweep_t<false, false>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
}

void generic_sweep(bool has_dir, bool has_q, const SweepArgs &a, const LineGeom &g, long inner_stride, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, double *wc, double *wd, hipStream_t st)
{
    const long nl = (long)g.n_inner * g.n_outer;
    const dim3 grid((unsigned)((nl + 255) / 256)), block(256);
    if (has_dir && has_q) hipLaunchKernelGGL((k_sweep_generic<true, true>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else if (has_q) hipLaunchKernelGGL((k_sweep_generic<false, true>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else if (has_dir) hipLaunchKernelGGL((k_sweep_generic<true, false>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else hipLaunchKernelGGL((k_sweep_generic<false, false>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
}

}  // namespace adi
