// adi_cyl.hip -- cylindrical (r, phi, z) backward-Euler ADI step for MI355X (gfx950).
//
// Replaces adi3d_cyl_phi_v3.py:332-350 (scheme="be": r-Thomas -> periodic phi solve -> z-Thomas)
// and the void-clamping wrapper adi_step_masked (quick_spiral_deposition_gif_v5.py:31-70).
// Field layout C-order (nr, nphi, nz): r is the slowest axis, z is contiguous.
//
//   r sweep   strided axis 0, lines indexed by the flattened (phi, z) index; the tridiagonal
//             coefficients depend on the radius index only (build_coeff_r, :155-202) and come from a
//             small per-plan table; the source term and the void pre-clamp are fused into the load.
//   phi sweep strided axis 1, periodic.  The reference solves it spectrally (phi_solve_spectral,
//             :302-329), i.e. it inverts the circulant tridiagonal (-f_i, 1+2 f_i, -f_i) exactly; here
//             the same system is solved as an ordinary tridiagonal plus a Sherman-Morrison rank-one
//             correction whose vector z_i = A'^-1 u depends on the radius only and is tabulated at
//             plan creation (valid for any nphi, power of two or not).  f_0 = 0: the axis row is identity.
//   z sweep   contiguous axis 2 (build_coeff_z, :255-298): constant coefficients, the two end rows
//             carry the neumann0 / dirichlet / robin closure; the void post-clamp is fused into the store.
//
// Algorithmic HBM traffic: 16 B/cell/sweep (+8 with a source, +1 per masked pass).
#include <cstdlib>
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "adi_common.hpp"
#include "adi_core.hpp"

struct adi_cyl_plan {
    int nr, nphi, nz, device;
    long sx;  // plane (one radius) stride in elements
    double rho, cp, dt;
    // r sweep
    double *d_ar, *d_br, *d_cr;  // [nr]
    double r_add_last;
    // phi sweep
    double *d_fac;      // [nr]
    double *d_zt;       // [nr * nphi]  Sherman-Morrison vector per radius
    double *d_smden;    // [nr]         1 / (1 + v.z)
    // z sweep
    double zf;                   // theta*alpha*dt/dz^2
    double zb0, zbN;             // diagonal of the first / last row
    double za_N, zc_0;           // off-diagonals of the end rows (0 for dirichlet)
    double zadd0, zaddN;         // RHS increments (robin)
    int zdir0, zdirN;            // dirichlet flags
    double zT0, zTN;             // dirichlet values
};

#ifndef ADI_CYL_NT
#define ADI_CYL_NT 1
#endif

namespace adi {

struct CylZ {
    double f, b0, bN, aN, c0, add0, addN, T0, TN;
    int dir0, dirN;
};

// ---- r sweep / phi sweep: strided kernel with table-driven rows ---------------------------------
// MODE 0: r sweep (axis 0).  MODE 1: phi sweep (axis 1, periodic, Sherman-Morrison).
template <int M, int MODE>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512) void k_cyl_strided(
    const double *__restrict__ in, double *__restrict__ out, int n, long stride, int n_inner, long outer_stride,
    int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ ta, const double *__restrict__ tb, const double *__restrict__ tc, double add_last,
    const double *__restrict__ S, double s_scale, const uint8_t *__restrict__ active_mask, double T_void,
    const double *__restrict__ fac, const double *__restrict__ zt, const double *__restrict__ smden)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < n_inner;
    const long base = to * outer_stride + kcol;
    const int r0 = sg * M;

    double a[M], b[M], c[M], d[M];
    double f = 0.0, b0 = 1.0;
    if (MODE == 1) {
        f = fac[to];
        b0 = 1.0 + 2.0 * f;
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const int row = r0 + r;
        const bool ok = active && row < n;
        const long p = base + (long)row * stride;
        double v = ok ? in[p] : 0.0;
        if (MODE == 0) {
            if (active_mask != nullptr && ok && active_mask[p] == 0) v = T_void;  // T_work[void] = ambient, :56-57
            if (S != nullptr && ok) v = v + s_scale * S[p];                    // R0 = Tn + dt*(S/(rho cp)), :339
            a[r] = (row < n) ? ta[row] : 0.0;
            b[r] = (row < n) ? tb[row] : 1.0;
            c[r] = (row < n) ? tc[row] : 0.0;
            if (row == n - 1) v = v + add_last;                                // rhs_r[:, -1] += ..., :201
        } else {
            const bool inr = row < n;
            if (n == 2) {  // both neighbours are the same cell
                a[r] = (inr && row == 1) ? -2.0 * f : 0.0;
                c[r] = (inr && row == 0) ? -2.0 * f : 0.0;
                b[r] = inr ? b0 : 1.0;
            } else {
                a[r] = (inr && row > 0) ? -f : 0.0;
                c[r] = (inr && row < n - 1) ? -f : 0.0;
                // Sherman-Morrison split with gamma = -b0: b'_0 = 2 b0, b'_{n-1} = b0 + f^2 / b0
                b[r] = !inr ? 1.0 : (row == 0 ? 2.0 * b0 : (row == n - 1 ? b0 + f * f / b0 : b0));
            }
        }
        d[r] = v;
    }

    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);

    const int ld = Lp + 1;
    const int plane = LINES * ld;
    double *sX1 = sm, *sX2 = sm + plane, *sCS = sm + 2 * plane, *sX4 = sm + 3 * plane;
    double *sGF = sm + 4 * plane, *sAF = sm + 5 * plane, *sCF = sm + 6 * plane, *sXS = sm + 7 * plane;
    {
        const int w = kk * ld + sg;
        const double aS = a[M - 1];
        sX1[w] = -aS * k.aL;
        sX2[w] = __builtin_fma(-aS, k.cL, b[M - 1]);
        sCS[w] = c[M - 1];
        sX4[w] = __builtin_fma(-aS, k.gL, d[M - 1]);
        sGF[w] = k.gF;
        sAF[w] = k.aF;
        sCF[w] = k.cF;
    }
    __syncthreads();
    {
        const int pl = tid >> (__ffs(Lp) - 1), ps = tid & (Lp - 1);   // Lp is a power of two
        const int w = pl * ld + ps;
        const double cS = sCS[w];
        const bool hasn = ps < Lp - 1;
        const double gFn = hasn ? sGF[w + 1] : 0.0, aFn = hasn ? sAF[w + 1] : 0.0, cFn = hasn ? sCF[w + 1] : 0.0;
        const double ra = sX1[w];
        const double rb = __builtin_fma(-cS, aFn, sX2[w]);
        const double rc = -cS * cFn;
        const double rd = __builtin_fma(-cS, gFn, sX4[w]);
        sXS[w] = pcr_solve(ra, rb, rc, rd, ps, Lp);
    }
    __syncthreads();
    const double xS = sXS[kk * ld + sg];
    const double xL = (sg > 0) ? sXS[kk * ld + sg - 1] : 0.0;
    double x[M];
    back_solve<M>(a, c, d, ip, xL, xS, x);

    if (MODE == 1 && n > 2) {
        // x = y - z * (v.y) / (1 + v.z),  v = (1, 0, ..., 0, beta/gamma) with beta/gamma = f / b0
        double *sY0 = sm, *sYN = sm + LINES;  // the condensation arrays are dead after the second barrier
#pragma unroll
        for (int r = 0; r < M; ++r) {
            if (r0 + r == 0) sY0[kk] = x[r];
            if (r0 + r == n - 1) sYN[kk] = x[r];
        }
        __syncthreads();
        const double mu = (sY0[kk] + (f / b0) * sYN[kk]) * smden[to];
        const double *z = zt + to * (long)n;
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (r0 + r < n) x[r] = __builtin_fma(-mu, z[r0 + r], x[r]);
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (active && (r0 + r) < n) {
#if ADI_CYL_NT
            __builtin_nontemporal_store(x[r], out + base + (long)(r0 + r) * stride);   // written once, whole 128-byte pieces
#else
            out[base + (long)(r0 + r) * stride] = x[r];
#endif
        }
}

// ---- z sweep: contiguous kernel, constant coefficients with end closures -----------------------
template <int M, bool VEC>
__global__ __launch_bounds__(256) void k_cyl_contig(const double *__restrict__ in, double *__restrict__ out,
                                                   long nlines, int n, int Lp, CylZ z,
                                                   const uint8_t *__restrict__ active_mask, double T_void,
                                                   double T_inner, long lines_per_r0, long sx)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lw = 64 >> (__ffs(Lp) - 1);
    const int li = lane & (Lp - 1);
    const unsigned line = (blockIdx.x * (blockDim.x >> 6) + wave) * (unsigned)lw + ((unsigned)lane >> (__ffs(Lp) - 1));
    const bool active = line < (unsigned long)nlines;
    const int r0 = li * M;
    const unsigned pi = line / (unsigned)lines_per_r0;   // radius index; lines_per_r0 = nphi
    const long base = (long)pi * sx + (long)(line - pi * (unsigned)lines_per_r0) * n + r0;

    double a[M], b[M], c[M], d[M];
    if (VEC) {
        if (active && r0 < n) {
            const double2 *q = reinterpret_cast<const double2 *>(in + base);
#pragma unroll
            for (int i = 0; i < M / 2; ++i) {
                const double2 t = q[i];
                d[2 * i] = t.x;
                d[2 * i + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = 0.0;
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) d[r] = (active && r0 + r < n) ? in[base + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const int row = r0 + r;
        const bool inr = row < n;
        double av = inr ? -z.f : 0.0, bv = inr ? 1.0 + 2.0 * z.f : 1.0, cv = inr ? -z.f : 0.0;
        if (row == 0) {                 // bottom closure, adi3d_cyl_phi_v3.py:271-283
            av = 0.0; bv = z.b0; cv = z.c0;
            d[r] = z.dir0 ? z.T0 : d[r] + z.add0;
        }
        if (row == n - 1) {             // top closure, :285-296 (applied last, as in the reference, when n == 1)
            av = (n == 1) ? 0.0 : z.aN; bv = z.bN; cv = 0.0;
            d[r] = z.dirN ? z.TN : d[r] + z.addN;
        }
        a[r] = av; b[r] = bv; c[r] = cv;
    }
    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);
    const double gFn = __shfl_down(k.gF, 1, Lp), aFn = __shfl_down(k.aF, 1, Lp), cFn = __shfl_down(k.cF, 1, Lp);
    double ra, rb, rc, rd;
    reduced_row(a[M - 1], b[M - 1], c[M - 1], d[M - 1], k, gFn, aFn, cFn, ra, rb, rc, rd);
    const double xS = pcr_solve(ra, rb, rc, rd, li, Lp);
    double xL = __shfl_up(xS, 1, Lp);
    if (li == 0) xL = 0.0;
    double x[M];
    back_solve<M>(a, c, d, ip, xL, xS, x);

    if (active_mask != nullptr && active) {   // Tnp1[void] = ambient_void; Tnp1[0, ~active[0]] = ambient_inner (:61-68)
        const bool axis_row = line < lines_per_r0;
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (r0 + r < n && active_mask[base + r] == 0) x[r] = axis_row ? T_inner : T_void;
    }
    if (VEC) {
        if (active && r0 < n) {
            double2 *q = reinterpret_cast<double2 *>(out + base);
#pragma unroll
            for (int i = 0; i < M / 2; ++i) q[i] = make_double2(x[2 * i], x[2 * i + 1]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (active && r0 + r < n) out[base + r] = x[r];
    }
}

// elementwise pass used when a sweep degenerates (nphi == 1) or the grid is too long for the fast path
__global__ __launch_bounds__(256) void k_copy(const double *__restrict__ in, double *__restrict__ out, size_t n)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[p];
}

static int strided_rows(int n)
{
    static const int force = [] { const char *e = getenv("ADI_CYL_M"); return e ? atoi(e) : 0; }();
    if (force == 4 || force == 8 || force == 16)
        if ((n + force - 1) / force <= 64) return force;
    return n <= 16 ? 2 : (n <= 32 ? 4 : (n <= 512 ? 8 : 16));
}
static int contig_rows(int n) { return n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : 16)); }

template <int M, int MODE>
static void launch_cyl_strided(const double *in, double *out, int n, long stride, int n_inner, long n_outer,
                               long outer_stride, const adi_cyl_plan *pl, const double *S, double s_scale,
                               const uint8_t *act, double T_void, hipStream_t st)
{
    const int Lp = next_pow2((n + M - 1) / M);
    static const int min_lines = [] { const char *e = getenv("ADI_CYL_LINES"); return e ? atoi(e) : 16; }();
    static const int min_threads = [] { const char *e = getenv("ADI_CYL_THREADS"); return e ? atoi(e) : 512; }();
    // 16 adjacent lines = whole 128-byte pieces per row; 512 threads per tile measured best on 128 x 256 x 512
    // (0.219 -> 0.205 ms per step against 8 lines / 256 threads)
    int lines = min_lines;
    while (lines * Lp < min_threads) lines <<= 1;
    while (lines > 8 && (lines * Lp > (M <= 8 ? 1024 : 512) || 64 * lines * (Lp + 1) > 48 * 1024)) lines >>= 1;
    const int tiles_inner = (n_inner + lines - 1) / lines;
    const long ntiles = (long)tiles_inner * n_outer;
    const size_t lds = (size_t)8 * lines * (Lp + 1) * sizeof(double);
    hipLaunchKernelGGL((k_cyl_strided<M, MODE>), dim3((unsigned)ntiles), dim3(lines * Lp), lds, st, in, out, n, stride,
                       n_inner, outer_stride, Lp, lines, tiles_inner, ntiles, pl->d_ar, pl->d_br, pl->d_cr,
                       pl->r_add_last, S, s_scale, act, T_void, pl->d_fac, pl->d_zt, pl->d_smden);
}

template <int MODE>
static void dispatch_cyl_strided(const double *in, double *out, int n, long stride, int n_inner, long n_outer,
                                 long outer_stride, const adi_cyl_plan *pl, const double *S, double s_scale,
                                 const uint8_t *act, double T_void, hipStream_t st)
{
    switch (strided_rows(n)) {
        case 2: launch_cyl_strided<2, MODE>(in, out, n, stride, n_inner, n_outer, outer_stride, pl, S, s_scale, act, T_void, st); break;
        case 4: launch_cyl_strided<4, MODE>(in, out, n, stride, n_inner, n_outer, outer_stride, pl, S, s_scale, act, T_void, st); break;
        case 8: launch_cyl_strided<8, MODE>(in, out, n, stride, n_inner, n_outer, outer_stride, pl, S, s_scale, act, T_void, st); break;
        default: launch_cyl_strided<16, MODE>(in, out, n, stride, n_inner, n_outer, outer_stride, pl, S, s_scale, act, T_void, st); break;
    }
}

template <int M>
static void launch_cyl_contig(const double *in, double *out, long nlines, int n, const CylZ &z, const uint8_t *act,
                              double T_void, double T_inner, long lines_per_r0, long sx, hipStream_t st)
{
    const int Lp = next_pow2((n + M - 1) / M);
    const int lw = 64 / Lp;
    const long waves = (nlines + lw - 1) / lw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    const bool vec = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && (n % M == 0) && (sx % 2 == 0);
    if (vec)
        hipLaunchKernelGGL((k_cyl_contig<M, true>), dim3(grid), dim3(256), 0, st, in, out, nlines, n, Lp, z, act, T_void, T_inner, lines_per_r0, sx);
    else
        hipLaunchKernelGGL((k_cyl_contig<M, false>), dim3(grid), dim3(256), 0, st, in, out, nlines, n, Lp, z, act, T_void, T_inner, lines_per_r0, sx);
}

}  // namespace adi

using namespace adi;

static void plan_free(adi_cyl_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    void *ptrs[] = {p->d_ar, p->d_br, p->d_cr, p->d_fac, p->d_zt, p->d_smden};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
}

extern "C" {

int adi_cyl_plan_create(int nr, int nphi, int nz, long plane_stride, double dr, double dphi, double dz, double rho,
                        double cp, double k,
                        double dt, double robin_h, double robin_Tinf, int kind_bot, int kind_top, double h_bot,
                        double h_top, double Tinf_bot, double Tinf_top, double T_bot, double T_top,
                        adi_cyl_plan **out)
{
    ADI_REQUIRE(out, "adi_cyl_plan_create: null output");
    ADI_REQUIRE(nr > 0 && nphi > 0 && nz > 0, "adi_cyl_plan_create: bad grid");
    ADI_REQUIRE(kind_bot >= 0 && kind_bot <= 2, "unknown zbc.kind_bot");  // ValueError, adi3d_cyl_phi_v3.py:283
    ADI_REQUIRE(kind_top >= 0 && kind_top <= 2, "unknown zbc.kind_top");  // :296
    ADI_REQUIRE(nr <= kMaxFastLine && nphi <= kMaxFastLine && nz <= kMaxFastLine,
                "adi_cyl_plan_create: axis longer than %d cells is not supported", kMaxFastLine);
    adi_cyl_plan *p = new (std::nothrow) adi_cyl_plan();
    if (!p) return set_err(ADI_ERR_HIP, "out of host memory");
    memset(p, 0, sizeof(*p));
    ADI_HIP_TRY(hipGetDevice(&p->device));
    p->nr = nr; p->nphi = nphi; p->nz = nz; p->rho = rho; p->cp = cp; p->dt = dt;
    p->sx = plane_stride ? plane_stride : (long)nphi * nz;
    if (p->sx < (long)nphi * nz) { delete p; return set_err(ADI_ERR_ARG, "adi_cyl_plan_create: plane_stride < nphi*nz"); }
    const double alpha = k / (rho * cp);  // Material.alpha, :48-50
    const double theta = 1.0;             // BE branch calls the builders with theta = 1.0 (:341, :348)

    // ---- r coefficients: build_coeff_r, adi3d_cyl_phi_v3.py:155-202 -------------------------------
    std::vector<double> ar(nr), br(nr), cr(nr), r_i(nr), r_imh(nr), r_iph(nr);
    for (int i = 0; i < nr; ++i) {
        const double r = ((double)i + 0.5) * dr;       // GridCyl.r, :38
        r_i[i] = fmax(r, 1e-15);
        r_imh[i] = fmax(r - 0.5 * dr, 1e-15);
        r_iph[i] = r + 0.5 * dr;
    }
    const double fac = theta * alpha * dt;
    for (int i = 1; i < nr - 1; ++i) {
        const double ai = -fac * (r_imh[i] / (r_i[i] * dr * dr));
        const double ci = -fac * (r_iph[i] / (r_i[i] * dr * dr));
        ar[i] = ai; cr[i] = ci; br[i] = 1.0 - (ai + ci);
    }
    {
        const double c0 = -fac * (r_iph[0] / (r_i[0] * dr * dr));
        ar[0] = 0.0; br[0] = 1.0 - c0; cr[0] = c0;
        const int N = nr - 1;
        const double aN = -fac * (r_imh[N] / (r_i[N] * dr * dr));
        double bN = 1.0 + fac * (r_imh[N] / (r_i[N] * dr * dr));
        p->r_add_last = 0.0;
        if (robin_h != 0.0) {
            bN += fac * (r_iph[N] * (robin_h / k)) / (r_i[N] * dr);
            p->r_add_last = fac * (r_iph[N] * (robin_h / k)) / (r_i[N] * dr) * robin_Tinf;
        }
        // the reference writes the outer row last, so for nr == 1 it overrides the axis row (:187-190)
        ar[N] = aN; br[N] = bN; cr[N] = 0.0;
        if (nr == 1) ar[0] = 0.0;  // a[0] multiplies nothing in thomas_batch
    }

    // ---- phi: fac_i, Sherman-Morrison vectors; phi_solve_spectral, :302-329 ------------------------
    std::vector<double> pf(nr, 0.0), zt((size_t)nr * nphi, 0.0), smden(nr, 1.0);
    for (int i = 1; i < nr; ++i) {
        const double r = ((double)i + 0.5) * dr;
        pf[i] = theta * alpha * dt / (r * r * dphi * dphi);
    }
    if (nphi > 2) {
        std::vector<double> cpv(nphi), dpv(nphi);
        for (int i = 0; i < nr; ++i) {
            const double f = pf[i], b0 = 1.0 + 2.0 * f;
            // A' z = u,  u = (gamma, 0, ..., 0, alpha_c) with gamma = -b0, alpha_c = -f
            const int n = nphi;
            auto bb = [&](int j) { return j == 0 ? 2.0 * b0 : (j == n - 1 ? b0 + f * f / b0 : b0); };
            auto uu = [&](int j) { return j == 0 ? -b0 : (j == n - 1 ? -f : 0.0); };
            cpv[0] = (-f) / bb(0);
            dpv[0] = uu(0) / bb(0);
            for (int j = 1; j < n; ++j) {
                const double den = bb(j) - (-f) * cpv[j - 1];
                cpv[j] = (j < n - 1 ? -f : 0.0) / den;
                dpv[j] = (uu(j) - (-f) * dpv[j - 1]) / den;
            }
            double *z = &zt[(size_t)i * n];
            z[n - 1] = dpv[n - 1];
            for (int j = n - 2; j >= 0; --j) z[j] = dpv[j] - cpv[j] * z[j + 1];
            smden[i] = 1.0 / (1.0 + z[0] + (f / b0) * z[n - 1]);
        }
    }

    // ---- z closure: build_coeff_z, :255-298 ---------------------------------------------------------
    {
        const double f = theta * alpha * dt / (dz * dz);
        p->zf = f;
        p->zc_0 = -f; p->za_N = -f; p->zadd0 = 0.0; p->zaddN = 0.0; p->zdir0 = 0; p->zdirN = 0;
        p->zT0 = T_bot; p->zTN = T_top;
        if (kind_bot == ADI_ZBC_NEUMANN0) { p->zb0 = 1.0 + f; }
        else if (kind_bot == ADI_ZBC_DIRICHLET) { p->zb0 = 1.0; p->zc_0 = 0.0; p->zdir0 = 1; }
        else { const double beta = h_bot / k; p->zb0 = 1.0 + f * (1.0 + beta * dz); p->zadd0 = (theta * alpha * dt) * (beta / dz) * Tinf_bot; }
        if (kind_top == ADI_ZBC_NEUMANN0) { p->zbN = 1.0 + f; }
        else if (kind_top == ADI_ZBC_DIRICHLET) { p->zbN = 1.0; p->za_N = 0.0; p->zdirN = 1; }
        else { const double beta = h_top / k; p->zbN = 1.0 + f * (1.0 + beta * dz); p->zaddN = (theta * alpha * dt) * (beta / dz) * Tinf_top; }
    }

    auto up = [&](double **dst, const std::vector<double> &v) -> bool {
        if (hipMalloc((void **)dst, v.size() * sizeof(double)) != hipSuccess) return false;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&p->d_ar, ar) || !up(&p->d_br, br) || !up(&p->d_cr, cr) || !up(&p->d_fac, pf) || !up(&p->d_zt, zt) ||
        !up(&p->d_smden, smden)) {
        plan_free(p);
        return set_err(ADI_ERR_HIP, "adi_cyl_plan_create: device table upload failed");
    }
    *out = p;
    return ADI_OK;
}

int adi_cyl_plan_destroy(adi_cyl_plan *plan)
{
    plan_free(plan);
    return ADI_OK;
}

int adi_cyl_step(const adi_cyl_plan *pl, const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b,
                 const double *d_S, const uint8_t *d_active, double T_void, double T_inner, void *stream)
{
    ADI_REQUIRE(pl && d_T_in && d_T_out && d_tmp_a && d_tmp_b, "adi_cyl_step: null argument");
    ADI_REQUIRE(d_tmp_a != d_tmp_b && d_tmp_a != d_T_in && d_tmp_b != d_T_in && d_T_out != d_tmp_b &&
                    d_T_out != d_tmp_a && d_T_out != d_T_in,
                "adi_cyl_step: the four field buffers must be distinct");
    hipStream_t st = as_stream(stream);
    const int nr = pl->nr, nphi = pl->nphi, nz = pl->nz;
    const long plane = (long)nphi * nz;
    ADI_REQUIRE(plane <= 0x7fffffffL, "adi_cyl_step: (nphi, nz) plane too large");
    // r sweep: T_in -> tmp_a (source and void pre-clamp fused)
    const double s_scale = pl->dt * (1.0 / (pl->rho * pl->cp));
    dispatch_cyl_strided<0>(d_T_in, d_tmp_a, nr, pl->sx, (int)plane, 1, 0, pl, d_S, s_scale, d_active, T_void, st);
    ADI_CHECK_LAUNCH();
    // phi sweep: tmp_a -> tmp_b   (nphi == 1: the reference returns a copy, :303-304)
    const double *zin = d_tmp_a;
    if (nphi > 1) {
        dispatch_cyl_strided<1>(d_tmp_a, d_tmp_b, nphi, nz, nz, nr, pl->sx, pl, nullptr, 0.0, nullptr, 0.0, st);
        ADI_CHECK_LAUNCH();
        zin = d_tmp_b;
    }
    // z sweep: -> T_out (void post-clamp fused)
    CylZ z;
    z.f = pl->zf; z.b0 = pl->zb0; z.bN = pl->zbN; z.aN = pl->za_N; z.c0 = pl->zc_0; z.add0 = pl->zadd0;
    z.addN = pl->zaddN; z.T0 = pl->zT0; z.TN = pl->zTN; z.dir0 = pl->zdir0; z.dirN = pl->zdirN;
    const long nlines = (long)nr * nphi;
    switch (contig_rows(nz)) {
        case 2: launch_cyl_contig<2>(zin, d_T_out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
        case 4: launch_cyl_contig<4>(zin, d_T_out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
        case 8: launch_cyl_contig<8>(zin, d_T_out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
        default: launch_cyl_contig<16>(zin, d_T_out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

}  // extern "C"
