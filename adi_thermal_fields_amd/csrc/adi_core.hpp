// adi_core.hpp -- device-side tridiagonal machinery shared by every sweep kernel (gfx950, wave64).
//
// Algorithm (one tridiagonal system per grid line; replaces thomas_solve,
// adi3d_numba_coeff.py:121-130 / _thomas_batch_axis0, adi3d_gpu_coeff.py:140-152):
//
//   A line of n rows is cut into Lp segments of M consecutive rows; one thread owns one segment
//   and keeps its rows in registers.  Row M-1 of a segment is its SEPARATOR, rows 0..M-2 are
//   INTERIOR rows.
//     phase 1  condense():   two O(M) recurrences (top-down and bottom-up elimination of the
//                            interior block) give the first/last interior unknown as an affine
//                            function of the two neighbouring separators;
//     phase 2  pcr_solve():  the Lp separators of a line form a tridiagonal system that is solved
//                            by parallel cyclic reduction across Lp lanes of one wave (log2 Lp
//                            steps of cross-lane shuffles, no LDS);
//     phase 3  back_solve(): with both neighbouring separators known, the interior block is
//                            solved from the stored inverse pivots.
//   Every input row is read from HBM once and every output row written once; nothing is spilled.
//   The systems are strictly diagonally dominant (b >= 1 + |a| + |c|), for which both the block
//   elimination and PCR are unconditionally stable without pivoting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adi {

// 1/x for x >= 1 (pivots of a diagonally dominant system): v_rcp_f64 + two Newton steps.
// ~1 ulp; no denormal/inf handling needed in this range.
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// Affine description of a segment's interior block seen from its two separators:
//   x_first = gF - aF * xL - cF * xS        x_last = gL - aL * xL - cL * xS
// (xL = separator of the previous segment, xS = own separator).
struct Cond {
    double gF, aF, cF, gL, aL, cL;
};

// Off-diagonals of the assembled ADI rows take two values only -- the uniform -theta*gamma where the row couples to its
// neighbour, 0 where it does not (line ends, cells outside the mask, Dirichlet rows; adi3d_gpu_coeff.py:173-187) -- so a
// kernel that is short of registers keeps them as one bit per row and materialises the double where it is used: 2 VGPRs
// instead of 2*M per off-diagonal, the same numbers in the same operations.
struct MaskedCoef {
    unsigned m;     // bit r: row r couples
    double v;       // the coupling coefficient
    __device__ __forceinline__ double operator[](int r) const { return ((m >> r) & 1u) ? v : 0.0; }
};

// phase 1.  a,b,c,d: the M rows (row M-1 = separator, untouched here).  ip: inverse pivots of the
// top-down factorisation of the interior block, kept for phase 3.  AV / CV: double[M] or MaskedCoef.
template <int M, class AV, class CV>
__device__ __forceinline__ void condense(const AV &a, const double (&b)[M], const CV &c,
                                         const double (&d)[M], double (&ip)[M - 1], Cond &k)
{
    constexpr int MI = M - 1;
    double y = d[0], e = 1.0;
    ip[0] = frcp(b[0]);
#pragma unroll
    for (int r = 1; r < MI; ++r) {
        const double w = a[r] * ip[r - 1];
        ip[r] = frcp(__builtin_fma(-w, c[r - 1], b[r]));
        y = __builtin_fma(-w, y, d[r]);
        e = -w * e;
    }
    k.gL = y * ip[MI - 1];
    k.aL = a[0] * (e * ip[MI - 1]);
    k.cL = c[MI - 1] * ip[MI - 1];

    double jp = frcp(b[MI - 1]);
    double z = d[MI - 1], f = 1.0;
#pragma unroll
    for (int r = MI - 2; r >= 0; --r) {
        const double w = c[r] * jp;
        jp = frcp(__builtin_fma(-w, a[r + 1], b[r]));
        z = __builtin_fma(-w, z, d[r]);
        f = -w * f;
    }
    k.gF = z * jp;
    k.aF = a[0] * jp;
    k.cF = c[MI - 1] * (f * jp);
}

// Reduced (separator) row of a segment from its own condensation `k` and the first-row data
// (gFn, aFn, cFn) of the NEXT segment of the same line (ignored when cS == 0).
__device__ __forceinline__ void reduced_row(double aS, double bS, double cS, double dS, const Cond &k,
                                            double gFn, double aFn, double cFn,
                                            double &ra, double &rb, double &rc, double &rd)
{
    ra = -aS * k.aL;
    rb = __builtin_fma(-cS, aFn, __builtin_fma(-aS, k.cL, bS));
    rc = -cS * cFn;
    rd = __builtin_fma(-cS, gFn, __builtin_fma(-aS, k.gL, dS));
}

// phase 2.  Parallel cyclic reduction over the Lp separators of a line held by Lp consecutive lanes
// (Lp a power of two <= 64, li = lane index inside the line).  Rows outside [0, Lp) are identity.
__device__ __forceinline__ double pcr_solve(double ra, double rb, double rc, double rd, int li, int Lp)
{
    for (int dl = 1; dl < Lp; dl <<= 1) {
        const double inv = frcp(rb);
        const bool hl = (li - dl) >= 0, hh = (li + dl) < Lp;
        const double a_lo = __shfl_up(ra, dl, Lp), c_lo = __shfl_up(rc, dl, Lp);
        const double d_lo = __shfl_up(rd, dl, Lp), i_lo = __shfl_up(inv, dl, Lp);
        const double a_hi = __shfl_down(ra, dl, Lp), c_hi = __shfl_down(rc, dl, Lp);
        const double d_hi = __shfl_down(rd, dl, Lp), i_hi = __shfl_down(inv, dl, Lp);
        const double k1 = hl ? ra * i_lo : 0.0;
        const double k2 = hh ? rc * i_hi : 0.0;
        rb = __builtin_fma(-k2, a_hi, __builtin_fma(-k1, c_lo, rb));
        rd = __builtin_fma(-k2, d_hi, __builtin_fma(-k1, d_lo, rd));
        ra = -k1 * a_lo;
        rc = -k2 * c_hi;
    }
    return rd * frcp(rb);
}

// phase 3.  x[M-1] = xS; interior rows from the stored inverse pivots.
template <int M, class AV, class CV>
__device__ __forceinline__ void back_solve(const AV &a, const CV &c, const double (&d)[M],
                                           const double (&ip)[M - 1], double xL, double xS, double (&x)[M])
{
    constexpr int MI = M - 1;
    double y[MI];
    y[0] = __builtin_fma(-a[0], xL, d[0]);
#pragma unroll
    for (int r = 1; r < MI; ++r) {
        const double w = a[r] * ip[r - 1];
        y[r] = __builtin_fma(-w, y[r - 1], d[r]);
    }
    y[MI - 1] = __builtin_fma(-c[MI - 1], xS, y[MI - 1]);
    x[MI - 1] = y[MI - 1] * ip[MI - 1];
#pragma unroll
    for (int r = MI - 2; r >= 0; --r) x[r] = __builtin_fma(-c[r], x[r + 1], y[r]) * ip[r];
    x[M - 1] = xS;
}


// ---- uniform-interior fast path -----------------------------------------------------------------------
// Inside a solid region every row of a segment is a = c = s (= -theta*gamma), b = bu (= 1 + 2*theta*gamma):
// the M-1 interior rows form the constant matrix Uint = tridiag(s, bu, s), whose inverse column
// p = Uint^-1 e_0 and Thomas factors are computed ONCE on the host and passed as kernel arguments
// (scalar registers).  A segment whose first row differs only in its diagonal / left coupling (line start,
// exposed cell with a Robin coefficient) is Uint + delta e0 e0^T and is handled branch-free by
// Sherman-Morrison:  A^-1 r = P r - kappa (p.r) p,  kappa = delta / (1 + delta p_0).
// Replaces 2*(M-1) reciprocal chains by two dot products with constants.
// The helpers below take the constants as any type UC with members s, bu, p[], w[], ip[] (UniC<M> in scalar registers;
// the marching fused kernel passes a view of a copy in LDS for its rare surface lanes).
template <int M>
struct UniC {
    double s, bu;
    double p[M - 1];    // first column of Uint^-1 (the last column is p reversed: Uint is persymmetric)
    double w[M - 1];    // Thomas multipliers  w[r] = s * ip[r-1]   (w[0] unused)
    double ip[M - 1];   // inverse pivots of the uniform Thomas factorisation
};

template <int M, class UC>
__device__ __forceinline__ void condense_uniform(const UC &U, double a0, double b0, const double (&d)[M],
                                                 Cond &k, double &kappa)
{
    constexpr int MI = M - 1;
    double guF = 0.0, guL = 0.0;
#pragma unroll
    for (int r = 0; r < MI; ++r) {
        guF = __builtin_fma(U.p[r], d[r], guF);
        guL = __builtin_fma(U.p[MI - 1 - r], d[r], guL);
    }
    const double delta = b0 - U.bu;
    kappa = delta * frcp(__builtin_fma(delta, U.p[0], 1.0));
    const double f1 = __builtin_fma(-kappa, U.p[0], 1.0);
    const double pl = U.p[MI - 1];
    const double kpl = kappa * pl;
    k.gF = guF * f1;
    k.gL = __builtin_fma(-kpl, guF, guL);
    k.aF = a0 * (U.p[0] * f1);
    k.aL = a0 * (pl * f1);
    k.cF = U.s * __builtin_fma(-kpl, U.p[0], pl);
    k.cL = U.s * __builtin_fma(-kpl, pl, U.p[0]);
}

// in place: on entry d = right-hand sides of the M rows, on exit d = solution (row M-1 = xS)
template <int M, class UC>
__device__ __forceinline__ void back_solve_uniform(const UC &U, double a0, double kappa, double (&d)[M],
                                                   double xL, double xS)
{
    constexpr int MI = M - 1;
    d[0] = __builtin_fma(-a0, xL, d[0]);
#pragma unroll
    for (int r = 1; r < MI; ++r) d[r] = __builtin_fma(-U.w[r], d[r - 1], d[r]);
    d[MI - 1] = __builtin_fma(-U.s, xS, d[MI - 1]);
    d[MI - 1] = d[MI - 1] * U.ip[MI - 1];
#pragma unroll
    for (int r = MI - 2; r >= 0; --r) d[r] = __builtin_fma(-U.s, d[r + 1], d[r]) * U.ip[r];
    const double t = kappa * d[0];
#pragma unroll
    for (int r = 0; r < MI; ++r) d[r] = __builtin_fma(-t, U.p[r], d[r]);
    d[M - 1] = xS;
}

// ---- mixed segments: the surface of the solid crosses the segment once ----------------------------------------
// Inside the block (rows 0..M-2) an in-mask run of L rows is coupled to ONE outside unknown through a uniform row at
// one end and stops in a line-start / line-end row (only one neighbour in the mask, b = bmod) at the other end; the
// rest of the block is outside the mask (identity rows).
//   REV = true  ("tail"): run = rows [M-1-L, M-1), coupled to the separator row x_S below it, line start at row M-1-L
//   REV = false ("head"): run = rows [0, L),       coupled to x_prev above it (coefficient a_c),  line end at row L-1
// q = 0..L-1 counts rows from the coupled end; row(q) = REV ? M-2-q : q.  Lanes of this kind are rare (two per line
// crossing the solid), so they may afford what the uniform path avoids: one reciprocal chain along the run.
template <int M, bool REV>
__device__ __forceinline__ double &mixed_row(double (&d)[M], int q) { return d[REV ? (M - 2 - q) : q]; }

// x_(q=0) = G - A * x_out.  Row 0 of the inverse of the run's matrix, without a single reciprocal per row: the solution y of
// the homogeneous rows q = 1 .. L-1 started at the modified end (y_{L-1} = 1, y_{L-2} = -bmod/s, y_{q-1} = -(bu y_q + s
// y_{q+1})/s) grows towards the coupled end -- the stable direction of the three-term recurrence -- and by symmetry of the
// matrix  (A^-1)_{0,q} = y_q / D  with  D = bu y_0 + s y_1  (row 0).  One FMA for y and one for the dot product per row, one
// reciprocal at the end; the elimination form it replaces ran a reciprocal chain along the run (~9 dependent fp64
// operations per row, on a curved solid in two lanes of every line: +17 % on the contiguous sweep of a 512^3 ellipsoid).
// Growth: y_0 <= (2 + 1/tg)^(L-1) with L <= M-1 rows, and the FAST kernels hold up to 32 rows per lane (strided axis 1 at 512
// rows; 30 at 480): y_0 times data of order 1e3 must stay below DBL_MAX, i.e. (2 + 1/tg)^30 < 1e300 -> tg > 1e-10.  The
// callers send surface segments to the GENERAL kernels when tg < kMixedMinTg = 1e-9 (growth <= 1e270); round 3's 1e-12 was
// sized for 16-row segments and left tg in [1e-12, 7e-11] on 480 - 512-row lines to overflow into inf / inf.  Such steps
// (dt a billionth of the cell's diffusion time: residual sub-steps of an event loop) are rare; the GENERAL kernels are exact.
constexpr double kMixedMinTg = 1e-9;
template <int M, bool REV, class UC>
__device__ __forceinline__ void mixed_condense(const UC &U, double (&d)[M], int L, double bmod, double a_c,
                                               double &G, double &A)
{
    constexpr int MI = M - 1;
    const double nis = frcp(-U.s);                // -1/s > 0
    const double cq = U.bu * nis;                 // -bu/s
    double y1 = 0.0, y2 = 0.0, acc = 0.0;         // y_{q+1}, y_{q+2}
#pragma unroll
    for (int q = MI - 1; q >= 0; --q) {
        if (q <= L - 1) {
            const double dq = mixed_row<M, REV>(d, q);
            double yq;
            if (q == L - 1) yq = 1.0;
            else yq = __builtin_fma((q == L - 2) ? bmod * nis : cq, y1, -y2);
            acc = __builtin_fma(yq, dq, acc);
            y2 = y1; y1 = yq;
        }
    }
    // y1 = y_0, y2 = y_1 (0 when L == 1)
    const double D = (L == 1) ? bmod : __builtin_fma(U.bu, y1, U.s * y2);
    const double iD = frcp(D);
    G = acc * iD;
    A = a_c * (y1 * iD);
}

// in place: rows of the run <- solution, given the outside unknown; Thomas from the coupled end with the prefix
// factors of the uniform block (w, ip) and one modified pivot at the far end
template <int M, bool REV, class UC>
__device__ __forceinline__ void mixed_back_solve(const UC &U, double (&d)[M], int L, double bmod, double a_c,
                                                 double x_out)
{
    constexpr int MI = M - 1;
    double y = 0.0;
#pragma unroll
    for (int q = 0; q < MI; ++q) {
        if (q < L) {
            double &dq = mixed_row<M, REV>(d, q);
            y = (q == 0) ? __builtin_fma(-a_c, x_out, dq) : __builtin_fma(-U.w[q], y, dq);
            dq = y;
        }
    }
    double xn = 0.0;
#pragma unroll
    for (int q = MI - 1; q >= 0; --q) {
        if (q < L) {
            double &dq = mixed_row<M, REV>(d, q);
            if (q == L - 1) {
                const double delta = bmod - U.bu;
                xn = dq * (U.ip[q] * frcp(__builtin_fma(delta, U.ip[q], 1.0)));      // 1 / (1/ip + delta)
            } else {
                xn = __builtin_fma(-U.s, xn, dq) * U.ip[q];
            }
            dq = xn;
        }
    }
}

// ISLAND: an in-mask run [m, m+L) of at most 16 rows that starts and ends inside the block (thin walls): nothing couples
// it to the rest of the line, so it is solved on the spot -- Thomas with one reciprocal chain.  The inverse pivots of the
// even rows are kept in 8 registers addressed by (row >> 1) & 7, unique inside a run of at most 16 rows; the back
// substitution recomputes the pivot of an odd row from its even predecessor with the same operations (bit-identical).
// bS / bE: diagonals of the first / last row of the run (line start / line end; bS alone when L == 1).
template <int M, class UC>
__device__ __forceinline__ void island_solve(const UC &U, double (&d)[M], int m, int L, double bS, double bE)
{
    constexpr int MI = M - 1;
    const int e = m + L;
    double invp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) invp[q] = 0.0;
    double ipPrev = 0.0;
#pragma unroll
    for (int r = 0; r < MI; ++r) {
        if (r >= m && r < e) {
            double P = (r == m) ? bS : ((r == e - 1) ? bE : U.bu);
            if (r > 0) {
                if (r > m) {
                    const double w = U.s * ipPrev;
                    d[r] = __builtin_fma(-w, d[r > 0 ? r - 1 : 0], d[r]);
                    P = __builtin_fma(-w, U.s, P);
                }
            }
            ipPrev = frcp(P);
            if ((r & 1) == 0) invp[(r >> 1) & 7] = ipPrev;
        }
    }
    double xn = 0.0;
#pragma unroll
    for (int r = MI - 1; r >= 0; --r) {
        if (r >= m && r < e) {
            double ip;
            if ((r & 1) == 0) {
                ip = invp[(r >> 1) & 7];
            } else {
                double P = (r == m) ? bS : ((r == e - 1) ? bE : U.bu);
                if (r > m) {
                    const double w = U.s * invp[((r > 0 ? r - 1 : 0) >> 1) & 7];
                    P = __builtin_fma(-w, U.s, P);
                }
                ip = frcp(P);
            }
            const double v = (r == e - 1) ? d[r] : __builtin_fma(-U.s, xn, d[r]);
            xn = v * ip;
            d[r] = xn;
        }
    }
}

// host: constants for (s, bu)
template <int M>
inline UniC<M> make_unic(double tg)
{
    constexpr int MI = M - 1;
    UniC<M> U;
    const long double s = -(long double)tg, bu = 1.0L + 2.0L * (long double)tg;
    U.s = (double)s;
    U.bu = (double)bu;
    long double ip[MI], w[MI], y[MI], x[MI];
    ip[0] = 1.0L / bu;
    w[0] = 0.0L;
    for (int r = 1; r < MI; ++r) {
        w[r] = s * ip[r - 1];
        ip[r] = 1.0L / (bu - w[r] * s);
    }
    y[0] = 1.0L;   // Uint p = e_0
    for (int r = 1; r < MI; ++r) y[r] = -w[r] * y[r - 1];
    x[MI - 1] = y[MI - 1] * ip[MI - 1];
    for (int r = MI - 2; r >= 0; --r) x[r] = (y[r] - s * x[r + 1]) * ip[r];
    for (int r = 0; r < MI; ++r) {
        U.p[r] = (double)x[r];
        U.w[r] = (double)w[r];
        U.ip[r] = (double)ip[r];
    }
    return U;
}

// ---- whole-block condensation and merging (distributed / two-level solves) ----------------------
// condense_full: like condense(), but ALL M rows form the block (no separator): first/last unknown of
// the block as affine functions of the value left of row 0 (xl) and right of row M-1 (xr):
//   x_first = gF - aF * xl - cF * xr        x_last = gL - aL * xl - cL * xr
template <int M>
__device__ __forceinline__ void condense_full(const double (&a)[M], const double (&b)[M], const double (&c)[M],
                                              const double (&d)[M], Cond &k)
{
    double y = d[0], e = 1.0;
    double ip = frcp(b[0]);
#pragma unroll
    for (int r = 1; r < M; ++r) {
        const double w = a[r] * ip;
        ip = frcp(__builtin_fma(-w, c[r - 1], b[r]));
        y = __builtin_fma(-w, y, d[r]);
        e = -w * e;
    }
    k.gL = y * ip;
    k.aL = a[0] * (e * ip);
    k.cL = c[M - 1] * ip;
    double jp = frcp(b[M - 1]);
    double z = d[M - 1], f = 1.0;
#pragma unroll
    for (int r = M - 2; r >= 0; --r) {
        const double w = c[r] * jp;
        jp = frcp(__builtin_fma(-w, a[r + 1], b[r]));
        z = __builtin_fma(-w, z, d[r]);
        f = -w * f;
    }
    k.gF = z * jp;
    k.aF = a[0] * jp;
    k.cF = c[M - 1] * (f * jp);
}

// Condensation of the concatenation [A | B] of two adjacent blocks (A's last row couples to B's first).
__device__ __forceinline__ Cond merge_cond(const Cond &A, const Cond &B)
{
    const double idet = frcp(__builtin_fma(-A.cL, B.aF, 1.0));
    const double v0 = __builtin_fma(-B.aF, A.gL, B.gF) * idet;   // x_F^B at xl = xr = 0
    const double u0 = __builtin_fma(-A.cL, B.gF, A.gL) * idet;   // x_L^A at xl = xr = 0
    Cond R;
    R.gF = __builtin_fma(-A.cF, v0, A.gF);
    R.aF = __builtin_fma(A.cF * B.aF, A.aL * idet, A.aF);
    R.cF = -(A.cF * B.cF) * idet;
    R.gL = __builtin_fma(-B.aL, u0, B.gL);
    R.aL = -(B.aL * A.aL) * idet;
    R.cL = __builtin_fma(B.aL * A.cL, B.cF * idet, B.cL);
    return R;
}

__device__ __forceinline__ Cond shfl_down_cond(const Cond &k, int delta, int width)
{
    Cond r;
    r.gF = __shfl_down(k.gF, delta, width); r.aF = __shfl_down(k.aF, delta, width);
    r.cF = __shfl_down(k.cF, delta, width); r.gL = __shfl_down(k.gL, delta, width);
    r.aL = __shfl_down(k.aL, delta, width); r.cL = __shfl_down(k.cL, delta, width);
    return r;
}

// Ordered tree reduction over the Lp lanes of a line: lane 0 ends up with the condensation of all blocks
// li = 0..nblk-1 (lanes >= nblk hold nothing and are skipped).
__device__ __forceinline__ Cond reduce_cond(Cond k, int li, int Lp, int nblk)
{
    for (int dl = 1; dl < Lp; dl <<= 1) {
        const Cond o = shfl_down_cond(k, dl, Lp);
        // lane li (a multiple of 2*dl) holds blocks [li, li+dl), its partner holds [li+dl, li+2dl)
        if (((li & (2 * dl - 1)) == 0) && (li + dl) < nblk) k = merge_cond(k, o);
    }
    return k;
}

// Bijective XCD-aware block remap: blocks b, b+8, b+16, ... are observed to share an XCD (and its
// L2); give each XCD a contiguous chunk of the tile range so neighbouring tiles hit the same L2.
// Placement only changes speed, never results.
__device__ __forceinline__ long xcd_chunk_tile(long b, long ntiles)
{
    const long q = ntiles >> 3, rem = ntiles & 7;
    const long x = b & 7, idx = b >> 3;
    return x * q + (x < rem ? x : rem) + idx;
}

// smallest power of two >= v (host + device)
__host__ __device__ inline int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace adi
