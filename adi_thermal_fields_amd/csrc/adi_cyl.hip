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
// Cache policy of this translation unit: PLAIN loads and stores (the Cartesian kernels stream their outputs with nt stores).
// The cylindrical sweeps run in place on a field of 134 MB at BASELINE configs[3] -- inside the 256 MB Infinity Cache -- and a
// streaming store evicts exactly the lines the next sweep is about to read: 0.145 -> 0.133 ms per step in the loop, 0.154 ->
// 0.143 ms with the reference's step semantics (scripts/cyl_probe.py, nt against plain).
#define ADI_STORE_AUX 0
#define ADI_LOAD_NT_CONTIG 0
#define ADI_CYL_NT 0
#include <cstdlib>
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "adi_cart_dev.hpp"

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
    void *d_uphi;       // [nr] UniC<phi_M>: constants of the uniform interior block (-f_i, 1+2f_i, -f_i) per radius
    int phi_M;          // rows per thread of the phi FAST kernel (0: not available for this nphi)
    // r sweep, FAST form: the factorisation of every segment of the (line-independent) r operator, built once
    double *d_rfac;     // [nseg][RF_STRIDE] per-segment tables, see CylRSeg
    int r_M, r_nseg;
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
    const double *in, double *out, int n, long stride, int n_inner, long outer_stride,
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
__global__ __launch_bounds__(256) void k_cyl_contig(const double *in, double *out,
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

// ---- FAST forms ------------------------------------------------------------------------------------------------
// The three operators of the BE step have coefficients that do not depend on the data and hardly on the position:
//   phi  (-f_i, 1+2f_i, -f_i) along the line, f_i by radius      -> the uniform-interior model of adi_core.hpp with one
//        UniC per radius (block-uniform: a tile lies in one radius plane), rows 0 / n-1 carry the Sherman-Morrison split
//   z    (-f, 1+2f, -f) everywhere, closures in rows 0 / n-1    -> the same model, one UniC per plan
//   r    a_i, b_i, c_i by radius index, the same for every line -> the whole factorisation of every segment (pivots,
//        multipliers, the six condensation numbers and the PCR multipliers of the separator system) is data-independent:
//        computed once on the host, the kernel only propagates right-hand sides
// No reciprocal is left in any of the three kernels; 50-60 VGPRs; every input read once, every output written once.

// phi sweep: tile = LINES adjacent z-lines x all nphi rows of one radius plane
template <int M>
__global__ __launch_bounds__(1024) void k_cyl_phi_fast(
    const double *in, double *out, int n, long stride, int n_inner, long outer_stride,
    int Lp, int LINES, int tiles_inner, long ntiles, const UniC<M> *__restrict__ utab,
    const double *__restrict__ fac, const double *__restrict__ zt, const double *__restrict__ smden)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // radius index: block-uniform
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);
    const int kcol = ti * LINES + kk;                                 // (host: n_inner % LINES == 0, Lp * M == n)
    const long base = to * outer_stride + kcol;
    const int r0 = sg * M;
    const UniC<M> U = utab[to];                                       // scalar loads: one radius per tile
    const double f = fac[to], b0 = 1.0 + 2.0 * f;

    double d[M];
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(ADI_LOAD_PRIO);
#endif
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = in[base + (long)(r0 + r) * stride];
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // Sherman-Morrison split with gamma = -b0 (see k_cyl_strided): b'_0 = 2 b0, b'_{n-1} = b0 + f^2 / b0
    const bool firstseg = r0 == 0, lastseg = r0 + M == n;
    const double a0 = firstseg ? 0.0 : U.s, bf = firstseg ? 2.0 * b0 : U.bu;
    const double aS = U.s, bS = lastseg ? b0 + f * f / b0 : U.bu, cS = lastseg ? 0.0 : U.s;
    Cond k;
    double kappa;
    condense_uniform<M>(U, a0, bf, d, k, kappa);
    double xL, xS;
    tile_separators(sm, tid, kk, sg, Lp, LINES, aS, bS, cS, d[M - 1], k, xL, xS);
    back_solve_uniform<M>(U, a0, kappa, d, xL, xS);
    // x = y - z * (v.y) / (1 + v.z),  v = (1, 0, ..., 0, beta/gamma) with beta/gamma = f / b0
    double *sY0 = sm, *sYN = sm + LINES;      // (the separator arrays are dead: every thread has read its xS / xL...
    __syncthreads();                          //  ...once all of them have passed this barrier)
    if (firstseg) sY0[kk] = d[0];
    if (lastseg) sYN[kk] = d[M - 1];
    __syncthreads();
    const double mu = (sY0[kk] + (f / b0) * sYN[kk]) * smden[to];
    const double *z = zt + to * (long)n + r0;
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const double x = __builtin_fma(-mu, z[r], d[r]);
#if ADI_CYL_NT
        __builtin_nontemporal_store(x, out + base + (long)(r0 + r) * stride);
#else
        out[base + (long)(r0 + r) * stride] = x;
#endif
    }
}

// z sweep: one wave = 64/Lp lines, lane li owns rows [li*M, li*M+M); coalesced loads / stores through a wave-private
// LDS strip (adi_cart_dev.hpp, coal_load).  Closures without a Dirichlet end (neumann0 / robin): row 0 and row n-1
// differ from the uniform row in their diagonal and right-hand side only.
template <int M>
__global__ __launch_bounds__(256) void k_cyl_z_fast(const double *in, double *out, long nlines,
                                                   int n, int Lp, CylZ z, UniC<M> U,
                                                   const uint8_t *__restrict__ active_mask, double T_void, double T_inner,
                                                   long lines_per_r0, long sx)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ __align__(16) double strips[4 * 32 * (M + 2)];
    double *strip = strips + wave * 32 * (M + 2);
    const int sh = __ffs(Lp) - 1;
    const int lw = 64 >> sh;
    const int li = lane & (Lp - 1);
    const unsigned line = (blockIdx.x * (blockDim.x >> 6) + wave) * (unsigned)lw + ((unsigned)lane >> sh);
    if ((long)(blockIdx.x * (blockDim.x >> 6) + wave) * lw >= nlines) return;       // (host: nlines % lw == 0)
    const int r0 = li * M;
    const unsigned pi = line / (unsigned)lines_per_r0;               // radius index; lines_per_r0 = nphi (host: % lw == 0)
    const long base = (long)pi * sx + (long)(line - pi * (unsigned)lines_per_r0) * n + r0;
    const long wbase = __shfl(base, 0);                              // the wave's 64*M doubles are contiguous from lane 0's
    double d[M];
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(ADI_LOAD_PRIO);
#endif
    coal_load<M>(in + wbase, strip, lane, d);
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const bool firstseg = li == 0, lastseg = r0 + M == n;
    const double a0 = firstseg ? 0.0 : U.s, bf = firstseg ? z.b0 : U.bu;
    if (firstseg) d[0] = d[0] + z.add0;                              // bottom closure, adi3d_cyl_phi_v3.py:271-283
    const double aS = lastseg ? z.aN : U.s, bS = lastseg ? z.bN : U.bu, cS = lastseg ? 0.0 : U.s;
    if (lastseg) d[M - 1] = d[M - 1] + z.addN;                       // top closure, :285-296
    Cond k;
    double kappa;
    condense_uniform<M>(U, a0, bf, d, k, kappa);
    const double gFn = __shfl_down(k.gF, 1, Lp), aFn = __shfl_down(k.aF, 1, Lp), cFn = __shfl_down(k.cF, 1, Lp);
    double ra, rb, rc, rd;
    reduced_row(aS, bS, cS, d[M - 1], k, gFn, aFn, cFn, ra, rb, rc, rd);
    const double xS = pcr_solve(ra, rb, rc, rd, li, Lp);
    double xL = __shfl_up(xS, 1, Lp);
    if (li == 0) xL = 0.0;
    back_solve_uniform<M>(U, a0, kappa, d, xL, xS);
    if (active_mask != nullptr) {   // Tnp1[void] = ambient_void; Tnp1[0, ~active[0]] = ambient_inner (:61-68)
        const bool axis_row = line < lines_per_r0;
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (active_mask[base + r] == 0) d[r] = axis_row ? T_inner : T_void;
    }
    coal_store<M>(out + wbase, strip, lane, d);
}

// r sweep.  Per segment (M rows, row M-1 = separator) the host stores, RF_STRIDE doubles each:
//   wf[M-1]  forward multipliers  a_r * ip_{r-1}  (wf[0] unused)      wb[M-1] backward multipliers c_r * jp_{r+1}
//   ip[M-1]  inverse pivots of the top-down factorisation             cr[M-1] the c_r of the interior rows
//   ipl, jpf the last / first inverse pivot of the two eliminations   aF, cF, aL, cL the matrix part of the condensation
//   a0       a of the segment's first row                             aS, cS   separator couplings
//   pk1[6], pk2[6], pinv  the PCR multipliers of this separator's row of the (line-independent) separator system
constexpr int RF_MAXM = 16;
constexpr int RF_STRIDE = 4 * (RF_MAXM - 1) + 10 + 13;
struct CylRSegView {
    const double *p;
    __device__ __forceinline__ double wf(int r) const { return p[r]; }
    __device__ __forceinline__ double wb(int r) const { return p[(RF_MAXM - 1) + r]; }
    __device__ __forceinline__ double ip(int r) const { return p[2 * (RF_MAXM - 1) + r]; }
    __device__ __forceinline__ double cr(int r) const { return p[3 * (RF_MAXM - 1) + r]; }
    __device__ __forceinline__ double sc(int i) const { return p[4 * (RF_MAXM - 1) + i]; }     // ipl jpf aF cF aL cL a0 aS cS bS
    __device__ __forceinline__ double pk(int i) const { return p[4 * (RF_MAXM - 1) + 10 + i]; }
};

// tile = 64 adjacent lines (one wave = one segment of 64 lines: the tables are wave-uniform -> scalar loads) x all
// segments; 512-byte row pieces
template <int M>
__global__ __launch_bounds__(1024) void k_cyl_r_fast(
    const double *in, double *out, int n, long stride, int n_inner, int nseg, int Lp,
    long ntiles, const double *__restrict__ rfac, double add_last, const double *__restrict__ S, double s_scale,
    const uint8_t *__restrict__ active_mask, double T_void)
{
    extern __shared__ __align__(16) double sm[];          // 2 x [Lp][64] separator right-hand sides + [Lp][64] first-row data
    constexpr int MI = M - 1;
    const int tid = threadIdx.x;
    const int kk = tid & 63;
    const int sg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const long base = tile * 64 + kk;                     // (host: n_inner % 64 == 0)
    const int r0 = sg * M;
    CylRSegView T;
    T.p = rfac + (size_t)sg * RF_STRIDE;
    double d[M];
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(ADI_LOAD_PRIO);
#endif
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const long p = base + (long)(r0 + r) * stride;
        double v = in[p];
        if (active_mask != nullptr && active_mask[p] == 0) v = T_void;        // T_work[void] = ambient, :56-57
        if (S != nullptr) v = v + s_scale * S[p];                            // R0 = Tn + dt*(S/(rho cp)), :339
        d[r] = v;
    }
#if ADI_LOAD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (r0 + M == n) d[M - 1] = d[M - 1] + add_last;                          // rhs_r[:, -1] += ..., :201
    // condensation of the right-hand side (the two recurrences of condense<M>, pivots from the table)
    double y = d[0], zb = d[MI - 1];
#pragma unroll
    for (int r = 1; r < MI; ++r) y = __builtin_fma(-T.wf(r), y, d[r]);
#pragma unroll
    for (int r = MI - 2; r >= 0; --r) zb = __builtin_fma(-T.wb(r), zb, d[r]);
    const double gL = y * T.sc(0), gF = zb * T.sc(1);
    // separator row: rd = dS - aS*gL - cS*gF(next segment); the matrix part of the row and of the whole PCR is tabulated
    double *sG = sm + 2 * Lp * 64;                        // gF of every segment, for the segment before it
    sG[sg * 64 + kk] = gF;
    __syncthreads();
    double rd = __builtin_fma(-T.sc(7), gL, d[M - 1]);
    if (sg + 1 < nseg) rd = __builtin_fma(-T.sc(8), sG[(sg + 1) * 64 + kk], rd);
    // PCR over the nseg separators of a line (the lanes of a line sit in different WAVES here: through LDS, one barrier
    // per level; log2(Lp) <= 6 levels).  rd_i <- rd_i - k1_i rd_{i-dl} - k2_i rd_{i+dl}
    int lev = 0;
    for (int dl = 1; dl < Lp; dl <<= 1, ++lev) {
        double *cur = sm + (lev & 1) * (Lp * 64);         // two buffers in turn: one barrier per level
        cur[sg * 64 + kk] = rd;
        __syncthreads();
        const double lo = (sg - dl >= 0) ? cur[(sg - dl) * 64 + kk] : 0.0;
        const double hi = (sg + dl < nseg) ? cur[(sg + dl) * 64 + kk] : 0.0;
        rd = __builtin_fma(-T.pk(6 + lev), hi, __builtin_fma(-T.pk(lev), lo, rd));
    }
    const double xS = rd * T.pk(12);
    sG[sg * 64 + kk] = xS;                                // (sG was last read before the first PCR barrier)
    __syncthreads();
    const double xL = (sg > 0) ? sG[(sg - 1) * 64 + kk] : 0.0;
    // back-substitution of the interior rows (back_solve<M> with tabulated pivots)
    d[0] = __builtin_fma(-T.sc(6), xL, d[0]);
#pragma unroll
    for (int r = 1; r < MI; ++r) d[r] = __builtin_fma(-T.wf(r), d[r - 1], d[r]);
    d[MI - 1] = __builtin_fma(-T.cr(MI - 1), xS, d[MI - 1]) * T.ip(MI - 1);
#pragma unroll
    for (int r = MI - 2; r >= 0; --r) d[r] = __builtin_fma(-T.cr(r), d[r + 1], d[r]) * T.ip(r);
    d[M - 1] = xS;
#pragma unroll
    for (int r = 0; r < M; ++r) {
#if ADI_CYL_NT
        __builtin_nontemporal_store(d[r], out + base + (long)(r0 + r) * stride);
#else
        out[base + (long)(r0 + r) * stride] = d[r];
#endif
    }
}

// elementwise pass used when a sweep degenerates (nphi == 1) or the grid is too long for the fast path
__global__ __launch_bounds__(256) void k_copy(const double *in, double *out, size_t n)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[p];
}

static int strided_rows(int n)
{
    return n <= 16 ? 2 : (n <= 32 ? 4 : (n <= 512 ? 8 : 16));
}
static int contig_rows(int n) { return n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : 16)); }

template <int M, int MODE>
static void launch_cyl_strided(const double *in, double *out, int n, long stride, int n_inner, long n_outer,
                               long outer_stride, const adi_cyl_plan *pl, const double *S, double s_scale,
                               const uint8_t *act, double T_void, hipStream_t st)
{
    const int Lp = next_pow2((n + M - 1) / M);
    const int min_lines = 16, min_threads = 512;
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
    void *ptrs[] = {p->d_ar, p->d_br, p->d_cr, p->d_fac, p->d_zt, p->d_smden, p->d_uphi, p->d_rfac};
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
    return adi_cyl_plan_create_annular(nr, nphi, nz, plane_stride, dr, dphi, dz, 0.0, rho, cp, k, dt, robin_h, robin_Tinf, kind_bot,
                                       kind_top, h_bot, h_top, Tinf_bot, Tinf_top, T_bot, T_top, out);
}

int adi_cyl_plan_create_annular(int nr, int nphi, int nz, long plane_stride, double dr, double dphi, double dz, double r_in,
                                double rho, double cp, double k,
                                double dt, double robin_h, double robin_Tinf, int kind_bot, int kind_top, double h_bot,
                                double h_top, double Tinf_bot, double Tinf_top, double T_bot, double T_top,
                                adi_cyl_plan **out)
{
    ADI_REQUIRE(r_in >= 0.0, "adi_cyl_plan_create: negative inner radius");
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
        const double r = r_in + ((double)i + 0.5) * dr;       // GridCyl.r, :38 (+ the inner radius of an annular grid)
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
        const double r = r_in + ((double)i + 0.5) * dr;
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

    // ---- FAST forms: constants of the uniform phi blocks per radius, factorisation tables of the r operator ----------
    p->phi_M = 0; p->d_uphi = nullptr; p->d_rfac = nullptr; p->r_M = 0; p->r_nseg = 0;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    std::vector<UniC<8>> u8;
    std::vector<UniC<16>> u16;
    // 16 rows per thread where the line allows it: the in-wave PCR over the separators costs the same per separator
    // whatever the segment length (measured on 128 x 256 x 512: 87 us with 8 rows, 59 us with 16, 32-line tiles)
    if (nphi >= 128 && nphi % 16 == 0 && pow2(nphi / 16) && nphi / 16 <= 32) {
        p->phi_M = 16;
        for (int i = 0; i < nr; ++i) u16.push_back(make_unic<16>(pf[i]));
    } else if (nphi >= 64 && nphi % 8 == 0 && pow2(nphi / 8) && nphi / 8 <= 32) {
        p->phi_M = 8;
        for (int i = 0; i < nr; ++i) u8.push_back(make_unic<8>(pf[i]));
    }
    std::vector<double> rfac;
    {
        int M = 0;
        if (nr >= 64 && nr % 8 == 0 && pow2(nr / 8) && nr / 8 <= 16) M = 8;
        else if (nr >= 64 && nr % 16 == 0 && pow2(nr / 16) && nr / 16 <= 16) M = 16;
        if (M) {
            const int MI = M - 1, nseg = nr / M;
            rfac.assign((size_t)nseg * RF_STRIDE, 0.0);
            std::vector<double> ra(nseg), rb(nseg), rc(nseg), aFv(nseg), cFv(nseg), aLv(nseg), cLv(nseg);
            for (int sgi = 0; sgi < nseg; ++sgi) {          // condense<M> of adi_core.hpp on the coefficients alone
                double *t = &rfac[(size_t)sgi * RF_STRIDE];
                const double *a = &ar[sgi * M], *b = &br[sgi * M], *c = &cr[sgi * M];
                double *wf = t, *wb = t + (RF_MAXM - 1), *ipv = t + 2 * (RF_MAXM - 1), *cv = t + 3 * (RF_MAXM - 1);
                double *sc = t + 4 * (RF_MAXM - 1);
                double e = 1.0;
                ipv[0] = 1.0 / b[0];
                for (int r = 1; r < MI; ++r) {
                    wf[r] = a[r] * ipv[r - 1];
                    ipv[r] = 1.0 / (b[r] - wf[r] * c[r - 1]);
                    e = -wf[r] * e;
                }
                for (int r = 0; r < MI; ++r) cv[r] = c[r];
                double jp = 1.0 / b[MI - 1], f2 = 1.0;
                for (int r = MI - 2; r >= 0; --r) {
                    wb[r] = c[r] * jp;
                    jp = 1.0 / (b[r] - wb[r] * a[r + 1]);
                    f2 = -wb[r] * f2;
                }
                sc[0] = ipv[MI - 1]; sc[1] = jp;
                aFv[sgi] = a[0] * jp; cFv[sgi] = c[MI - 1] * (f2 * jp);
                aLv[sgi] = a[0] * (e * ipv[MI - 1]); cLv[sgi] = c[MI - 1] * ipv[MI - 1];
                sc[2] = aFv[sgi]; sc[3] = cFv[sgi]; sc[4] = aLv[sgi]; sc[5] = cLv[sgi];
                sc[6] = a[0]; sc[7] = a[MI]; sc[8] = c[MI]; sc[9] = b[MI];
            }
            for (int sgi = 0; sgi < nseg; ++sgi) {          // reduced_row, matrix part
                const double *sc = &rfac[(size_t)sgi * RF_STRIDE + 4 * (RF_MAXM - 1)];
                const double aS = sc[7], cS = sc[8], bS = sc[9];
                const double aFn = sgi + 1 < nseg ? aFv[sgi + 1] : 0.0, cFn = sgi + 1 < nseg ? cFv[sgi + 1] : 0.0;
                ra[sgi] = -aS * aLv[sgi];
                rb[sgi] = bS - aS * cLv[sgi] - cS * aFn;
                rc[sgi] = -cS * cFn;
            }
            int lev = 0;
            for (int dl = 1; dl < nseg; dl <<= 1, ++lev) {  // pcr_solve, matrix part: the multipliers of every level
                std::vector<double> na(nseg), nb(nseg), nc(nseg);
                for (int i = 0; i < nseg; ++i) {
                    const bool hl = i - dl >= 0, hh = i + dl < nseg;
                    const double k1 = hl ? ra[i] / rb[i - dl] : 0.0, k2 = hh ? rc[i] / rb[i + dl] : 0.0;
                    double *pk = &rfac[(size_t)i * RF_STRIDE + 4 * (RF_MAXM - 1) + 10];
                    pk[lev] = k1; pk[6 + lev] = k2;
                    nb[i] = rb[i] - (hl ? k1 * rc[i - dl] : 0.0) - (hh ? k2 * ra[i + dl] : 0.0);
                    na[i] = hl ? -k1 * ra[i - dl] : 0.0;
                    nc[i] = hh ? -k2 * rc[i + dl] : 0.0;
                }
                ra = na; rb = nb; rc = nc;
            }
            for (int i = 0; i < nseg; ++i) rfac[(size_t)i * RF_STRIDE + 4 * (RF_MAXM - 1) + 10 + 12] = 1.0 / rb[i];
            p->r_M = M; p->r_nseg = nseg;
        }
    }

    auto up = [&](double **dst, const std::vector<double> &v) -> bool {
        if (hipMalloc((void **)dst, v.size() * sizeof(double)) != hipSuccess) return false;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&p->d_ar, ar) || !up(&p->d_br, br) || !up(&p->d_cr, cr) || !up(&p->d_fac, pf) || !up(&p->d_zt, zt) ||
        !up(&p->d_smden, smden) || (p->r_M && !up(&p->d_rfac, rfac))) {
        plan_free(p);
        return set_err(ADI_ERR_HIP, "adi_cyl_plan_create: device table upload failed");
    }
    if (p->phi_M) {
        const void *src = p->phi_M == 8 ? (const void *)u8.data() : (const void *)u16.data();
        const size_t bytes = p->phi_M == 8 ? u8.size() * sizeof(UniC<8>) : u16.size() * sizeof(UniC<16>);
        if (hipMalloc(&p->d_uphi, bytes) != hipSuccess || hipMemcpy(p->d_uphi, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            plan_free(p);
            return set_err(ADI_ERR_HIP, "adi_cyl_plan_create: device table upload failed");
        }
    }
    *out = p;
    return ADI_OK;
}

int adi_cyl_plan_destroy(adi_cyl_plan *plan)
{
    plan_free(plan);
    return ADI_OK;
}

static int cyl_sweep_r(const adi_cyl_plan *pl, const double *in, double *out, const double *d_S, const uint8_t *d_active,
                       double T_void, hipStream_t st)
{
    const int nr = pl->nr;
    const long plane = (long)pl->nphi * pl->nz;
    const double s_scale = pl->dt * (1.0 / (pl->rho * pl->cp));
    if (pl->r_M && plane % 64 == 0 && (long)nr * pl->sx < (1L << 31)) {
        const int Lp = pl->r_nseg;
        const long ntiles = plane / 64;
        const size_t lds = (size_t)3 * Lp * 64 * sizeof(double);
        if (pl->r_M == 8)
            hipLaunchKernelGGL((k_cyl_r_fast<8>), dim3((unsigned)ntiles), dim3(64 * Lp), lds, st, in, out, nr, pl->sx, (int)plane,
                               pl->r_nseg, Lp, ntiles, pl->d_rfac, pl->r_add_last, d_S, s_scale, d_active, T_void);
        else
            hipLaunchKernelGGL((k_cyl_r_fast<16>), dim3((unsigned)ntiles), dim3(64 * Lp), lds, st, in, out, nr, pl->sx, (int)plane,
                               pl->r_nseg, Lp, ntiles, pl->d_rfac, pl->r_add_last, d_S, s_scale, d_active, T_void);
    } else {
        dispatch_cyl_strided<0>(in, out, nr, pl->sx, (int)plane, 1, 0, pl, d_S, s_scale, d_active, T_void, st);
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

static int cyl_sweep_phi(const adi_cyl_plan *pl, const double *in, double *out, hipStream_t st)
{
    const int nr = pl->nr, nphi = pl->nphi, nz = pl->nz;
    if (pl->phi_M && nz % 32 == 0) {
        const int M = pl->phi_M, Lp = nphi / M;
        int lines = 32;                                   // 256-byte row pieces (64-line tiles, one wave per segment and the
        while (lines * Lp > 1024) lines >>= 1;            // Sherman-Morrison vector by scalar loads: 62.7 us against 55.6)
        const int tiles_inner = nz / lines;
        const long ntiles = (long)tiles_inner * nr;
        const size_t lds = (size_t)7 * lines * (Lp + 1) * sizeof(double);
        if (M == 8)
            hipLaunchKernelGGL((k_cyl_phi_fast<8>), dim3((unsigned)ntiles), dim3(lines * Lp), lds, st, in, out, nphi, (long)nz, nz,
                               pl->sx, Lp, lines, tiles_inner, ntiles, (const UniC<8> *)pl->d_uphi, pl->d_fac, pl->d_zt, pl->d_smden);
        else
            hipLaunchKernelGGL((k_cyl_phi_fast<16>), dim3((unsigned)ntiles), dim3(lines * Lp), lds, st, in, out, nphi, (long)nz, nz,
                               pl->sx, Lp, lines, tiles_inner, ntiles, (const UniC<16> *)pl->d_uphi, pl->d_fac, pl->d_zt, pl->d_smden);
    } else {
        dispatch_cyl_strided<1>(in, out, nphi, nz, nz, nr, pl->sx, pl, nullptr, 0.0, nullptr, 0.0, st);
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

static int cyl_sweep_z(const adi_cyl_plan *pl, const double *in, double *out, const uint8_t *d_active, double T_void,
                       double T_inner, hipStream_t st)
{
    const int nr = pl->nr, nphi = pl->nphi, nz = pl->nz;
    CylZ z;
    z.f = pl->zf; z.b0 = pl->zb0; z.bN = pl->zbN; z.aN = pl->za_N; z.c0 = pl->zc_0; z.add0 = pl->zadd0;
    z.addN = pl->zaddN; z.T0 = pl->zT0; z.TN = pl->zTN; z.dir0 = pl->zdir0; z.dirN = pl->zdirN;
    const long nlines = (long)nr * nphi;
    const int Lpf = nz / 16, lwf = Lpf > 0 ? 64 / Lpf : 0;
    const bool fast = !z.dir0 && !z.dirN && nz >= 128 && nz % 16 == 0 && Lpf <= 64 && (Lpf & (Lpf - 1)) == 0 &&
                      nphi % lwf == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && pl->sx % 2 == 0;
    if (fast) {
        const long waves = nlines / lwf;
        hipLaunchKernelGGL((k_cyl_z_fast<16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, in, out, nlines, nz, Lpf, z,
                           make_unic<16>(z.f), d_active, T_void, T_inner, (long)nphi, pl->sx);
    } else {
        switch (contig_rows(nz)) {
            case 2: launch_cyl_contig<2>(in, out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
            case 4: launch_cyl_contig<4>(in, out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
            case 8: launch_cyl_contig<8>(in, out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
            default: launch_cyl_contig<16>(in, out, nlines, nz, z, d_active, T_void, T_inner, nphi, pl->sx, st); break;
        }
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_cyl_sweep(const adi_cyl_plan *pl, int axis, const double *d_in, double *d_out, const double *d_S,
                  const uint8_t *d_active, double T_void, double T_inner, void *stream)
{
    // d_out == d_in is allowed: every thread of every sweep kernel in this file reads only the rows it later writes (and
    // reads them before the barriers that precede its stores), which is why `in` / `out` carry no __restrict__ here
    ADI_REQUIRE(pl && d_in && d_out, "adi_cyl_sweep: bad argument");
    ADI_REQUIRE(axis >= 0 && axis < 3, "adi_cyl_sweep: bad axis %d", axis);
    ADI_REQUIRE((long)pl->nphi * pl->nz <= 0x7fffffffL, "adi_cyl_sweep: (nphi, nz) plane too large");
    hipStream_t st = as_stream(stream);
    if (axis == 0) return cyl_sweep_r(pl, d_in, d_out, d_S, d_active, T_void, st);
    if (axis == 1) {
        if (pl->nphi == 1) {     // the reference returns a copy, adi3d_cyl_phi_v3.py:303-304
            const size_t cnt = (size_t)pl->nr * pl->sx;
            hipLaunchKernelGGL(k_copy, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_in, d_out, cnt);
            ADI_CHECK_LAUNCH();
            return ADI_OK;
        }
        return cyl_sweep_phi(pl, d_in, d_out, st);
    }
    return cyl_sweep_z(pl, d_in, d_out, d_active, T_void, T_inner, st);
}

int adi_cyl_step(const adi_cyl_plan *pl, const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b,
                 const double *d_S, const uint8_t *d_active, double T_void, double T_inner, void *stream)
{
    (void)d_tmp_a; (void)d_tmp_b;     // unused since ABI v13: the phi and z sweeps run in place on d_T_out
    ADI_REQUIRE(pl && d_T_in && d_T_out, "adi_cyl_step: null argument");
    ADI_REQUIRE(d_T_out != d_T_in, "adi_cyl_step: T_out aliases T_in (the step returns a new field, adi3d_cyl_phi_v3.py:350)");
    hipStream_t st = as_stream(stream);
    ADI_REQUIRE((long)pl->nphi * pl->nz <= 0x7fffffffL, "adi_cyl_step: (nphi, nz) plane too large");
    // r sweep: T_in -> T_out (source and void pre-clamp fused); then phi and z IN PLACE on T_out: the working set of the
    // two sweeps is one field (134 MB at 128 x 256 x 512, inside the 256 MB Infinity Cache) instead of two
    if (int rc = cyl_sweep_r(pl, d_T_in, d_T_out, d_S, d_active, T_void, st)) return rc;
    // (nphi == 1: the reference's phi solve returns a copy, :303-304 -- nothing to do)
    if (pl->nphi > 1)
        if (int rc = cyl_sweep_phi(pl, d_T_out, d_T_out, st)) return rc;
    // z sweep (void post-clamp fused)
    return cyl_sweep_z(pl, d_T_out, d_T_out, d_active, T_void, T_inner, st);
}

}  // extern "C"
