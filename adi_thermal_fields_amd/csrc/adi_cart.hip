// adi_cart.hip -- Cartesian masked-voxel ADI step for MI355X (gfx950): hand-written HIP kernels.
//
//   K0 k_build_coeffs   one pass: Robin coefficient + Neumann flux fields for the three axes
//                       (precompute_coeff_packs_unified, adi3d_numba_coeff.py:57-118)
//   K1 k_explicit       fused masked 7-point explicit stage -> R0
//                       (lap1D_x/y/z + R0, adi3d_numba_coeff.py:240-288, :298)
//   K2 k_sweep_strided  batched tridiagonal sweep along a strided axis (memory axes 0 and 1)
//   K3 k_sweep_contig   batched tridiagonal sweep along the contiguous axis (memory axis 2)
//                       (sweep_axis0/1/2, adi3d_numba_coeff.py:133-237, in the full-length
//                        identity-row form of adi3d_gpu_coeff.py:154-191)
//   K4 k_sweep_generic  thread-per-line Thomas with HBM scratch for lines longer than 1024 rows
//
// Data layout: C-order (n0, n1, n2) fp64 fields and 1-byte masks, exactly the reference's
// (adi3d_numba_coeff.py:18, :31-36); axis 2 is contiguous.  All kernels are HBM-bandwidth bound
// (< 1 flop/byte); no MFMA.
#include <stdlib.h>

#include "adi_common.hpp"
#include "adi_core.hpp"

namespace adi {

struct SweepScal {
    double tg;    // theta * gamma
    double dt;
    double Tinf;
};

// Assemble one row of the full-length system (adi3d_gpu_coeff.py:173-187; numba form :147-162).
//   m / mL / mR : cell, previous and next cell of the line are in the mask
//   off-mask    : identity row keeping the incoming value
//   Dirichlet   : identity row with the prescribed value
template <bool HAS_DIR, bool HAS_Q>
__device__ __forceinline__ void assemble_row(bool m, bool mL, bool mR, bool dir, double in, double co,
                                             double dv, double q, const SweepScal &s,
                                             double &a, double &b, double &c, double &d)
{
    const bool fr = HAS_DIR ? (m && !dir) : m;
    const bool L = m && mL, R = m && mR;
    const double dc = s.dt * co;
    const double nnb = (double)((int)L + (int)R);
    a = (fr && L) ? -s.tg : 0.0;
    c = (fr && R) ? -s.tg : 0.0;
    b = fr ? (1.0 + s.tg * nnb + dc) : 1.0;
    double rhs = in;
    if (HAS_Q) rhs = rhs + s.dt * q;
    rhs = rhs + dc * s.Tinf;
    d = fr ? rhs : ((HAS_DIR && m) ? dv : in);
}

// ------------------------------------------------------------------------------------------------
// K3: contiguous-axis sweep.  One wave solves 64/Lp lines; lane li of a line owns rows
// [li*M, li*M+M) in registers.  No LDS, no barriers: waves are fully independent, so a CU holds
// many lines in different phases and HBM requests never drain.
// VEC: n % M == 0 and M even -> every lane's chunk is whole and 16-byte aligned (dwordx4 accesses).
// ------------------------------------------------------------------------------------------------
template <int M, bool VEC>
__device__ __forceinline__ void load_rows_contig(const double *__restrict__ p, long base, int r0, int n,
                                                 bool active, double (&v)[M])
{
    if (VEC) {
        if (active && r0 < n) {
            const double2 *q = reinterpret_cast<const double2 *>(p + base);
#pragma unroll
            for (int i = 0; i < M / 2; ++i) {
                const double2 t = q[i];
                v[2 * i] = t.x;
                v[2 * i + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < M; ++r) v[r] = 0.0;
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) v[r] = (active && r0 + r < n) ? p[base + r] : 0.0;
    }
}

// M flag/mask bytes of a lane's chunk, one byte per row in `b[r]`
template <int M, bool VEC>
__device__ __forceinline__ void load_bytes_contig(const uint8_t *__restrict__ p, long base, int r0, int n,
                                                  bool active, unsigned (&b)[M])
{
#pragma unroll
    for (int r = 0; r < M; ++r) b[r] = 0;
    if (VEC) {
        if (active && r0 < n) {
            if (M == 2) {
                const unsigned w = *reinterpret_cast<const uint16_t *>(p + base);
                b[0] = w & 0xffu;
                b[1] = w >> 8;
            } else if (M == 4) {
                const unsigned w = *reinterpret_cast<const uint32_t *>(p + base);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = (w >> (8 * r)) & 0xffu;
            } else {
#pragma unroll
                for (int h = 0; h < M / 8; ++h) {
                    const uint64_t w = *reinterpret_cast<const uint64_t *>(p + base + 8 * h);
#pragma unroll
                    for (int r = 0; r < 8; ++r) b[8 * h + r] = (unsigned)((w >> (8 * r)) & 0xffull);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (active && r0 + r < n) b[r] = p[base + r];
    }
}

template <int M, bool VEC, bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_sweep_contig(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, long nlines, int n, int Lp, SweepScal s)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lw = 64 / Lp;  // lines per wave
    const int li = lane & (Lp - 1);
    const long line = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * lw + (lane / Lp);
    const bool active = line < nlines;
    const int r0 = li * M;
    const long base = line * (long)n + r0;

    double vin[M], vco[M], vdv[M], vq[M];
    load_rows_contig<M, VEC>(in, base, r0, n, active, vin);
    load_rows_contig<M, VEC>(coeff, base, r0, n, active, vco);
    if (HAS_DIR) load_rows_contig<M, VEC>(dval, base, r0, n, active, vdv);
    if (HAS_Q) load_rows_contig<M, VEC>(qf, base, r0, n, active, vq);
    unsigned fb[M], db[M];
    load_bytes_contig<M, VEC>(flags, base, r0, n, active, fb);
    if (HAS_DIR) load_bytes_contig<M, VEC>(dmask, base, r0, n, active, db);

    double a[M], b[M], c[M], d[M];
#pragma unroll
    for (int r = 0; r < M; ++r)   // flags: bit0 cell in mask, bit5 / bit6 the z- / z+ neighbour is in the mask
        assemble_row<HAS_DIR, HAS_Q>(fb[r] & 1u, (fb[r] >> 5) & 1u, (fb[r] >> 6) & 1u, HAS_DIR && db[r] != 0, vin[r],
                                     vco[r], HAS_DIR ? vdv[r] : 0.0, HAS_Q ? vq[r] : 0.0, s, a[r], b[r], c[r], d[r]);

    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);
    // first-row data of the next segment of the same line
    const double gFn = __shfl_down(k.gF, 1, Lp), aFn = __shfl_down(k.aF, 1, Lp), cFn = __shfl_down(k.cF, 1, Lp);
    double ra, rb, rc, rd;
    reduced_row(a[M - 1], b[M - 1], c[M - 1], d[M - 1], k, gFn, aFn, cFn, ra, rb, rc, rd);
    const double xS = pcr_solve(ra, rb, rc, rd, li, Lp);
    double xL = __shfl_up(xS, 1, Lp);
    if (li == 0) xL = 0.0;
    double x[M];
    back_solve<M>(a, c, d, ip, xL, xS, x);

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

// ------------------------------------------------------------------------------------------------
// K2: strided-axis sweep.  A workgroup owns a tile of LINES adjacent lines (LINES*8 B contiguous per
// row: full 128-byte lines for LINES = 16) and all Lp segments of each; thread (s, kk) keeps the M rows
// of segment s of line kk in registers (lanes run along the contiguous direction, so every access is
// coalesced without a transpose).  Only the 7 condensation numbers per segment travel through LDS to
// regroup the separator system line-major for the in-wave PCR, and the separator values travel back.
// ------------------------------------------------------------------------------------------------
template <int M, bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512) void k_sweep_strided(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, int n, long stride, int n_inner, long outer_stride, int Lp, int LINES,
    int tiles_inner, long ntiles, int lbit, SweepScal s)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const long to = tile / tiles_inner;
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid % LINES, sg = tid / LINES;
    const int kcol = ti * LINES + kk;
    const bool active = kcol < n_inner;
    const long base = to * outer_stride + kcol;
    const int r0 = sg * M;

    double a[M], b[M], c[M], d[M];
    {
        double vin[M], vco[M], vdv[M], vq[M];
        unsigned fb[M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const bool ok = active && (r0 + r) < n;
            const long p = base + (long)(r0 + r) * stride;
            fb[r] = ok ? flags[p] : 0u;
            vin[r] = ok ? in[p] : 0.0;
            vco[r] = ok ? coeff[p] : 0.0;
            if (HAS_DIR) vdv[r] = ok ? dval[p] : 0.0;
            if (HAS_Q) vq[r] = ok ? qf[p] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < M; ++r) {
            bool dir = false;
            if (HAS_DIR) dir = active && (r0 + r) < n && dmask[base + (long)(r0 + r) * stride] != 0;
            assemble_row<HAS_DIR, HAS_Q>(fb[r] & 1u, (fb[r] >> lbit) & 1u, (fb[r] >> (lbit + 1)) & 1u, dir, vin[r],
                                         vco[r], HAS_DIR ? vdv[r] : 0.0, HAS_Q ? vq[r] : 0.0, s, a[r], b[r], c[r], d[r]);
        }
    }

    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);

    // LDS: 8 arrays [LINES][Lp + 1] (one padding column: conflict-free for both access directions)
    const int ld = Lp + 1;
    const int plane = LINES * ld;
    double *sX1 = sm, *sX2 = sm + plane, *sCS = sm + 2 * plane, *sX4 = sm + 3 * plane;
    double *sGF = sm + 4 * plane, *sAF = sm + 5 * plane, *sCF = sm + 6 * plane, *sXS = sm + 7 * plane;
    {
        const int w = kk * ld + sg;
        const double aS = a[M - 1];
        sX1[w] = -aS * k.aL;                                  // ra
        sX2[w] = __builtin_fma(-aS, k.cL, b[M - 1]);          // rb without the next-segment term
        sCS[w] = c[M - 1];
        sX4[w] = __builtin_fma(-aS, k.gL, d[M - 1]);          // rd without the next-segment term
        sGF[w] = k.gF;
        sAF[w] = k.aF;
        sCF[w] = k.cF;
    }
    __syncthreads();
    {
        const int pl = tid / Lp, ps = tid - pl * Lp;  // line-major regrouping: Lp consecutive lanes = one line
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
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (active && (r0 + r) < n) out[base + (long)(r0 + r) * stride] = x[r];
}

// ------------------------------------------------------------------------------------------------
// K4: generic fallback, one thread per line, normalised Thomas (adi3d_gpu_coeff.py:140-152) with the
// forward-pass c', d' kept in an HBM workspace.  Used only for lines longer than kMaxFastLine rows.
// ------------------------------------------------------------------------------------------------
template <bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_sweep_generic(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, int n, long stride, long n_inner, long inner_stride, long n_outer,
    long outer_stride, int lbit, double *__restrict__ wc, double *__restrict__ wd, SweepScal s)
{
    const long lid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= n_inner * n_outer) return;
    const long o = lid / n_inner, kc = lid - o * n_inner;
    const long base = o * outer_stride + kc * inner_stride;
    double cp = 0.0, dp = 0.0;
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> lbit) & 1u, (f >> (lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0, in[p],
                                     coeff[p], HAS_DIR ? dval[p] : 0.0, HAS_Q ? qf[p] : 0.0, s, a, b, c, d);
        const double inv = 1.0 / (b - a * cp);
        cp = c * inv;
        dp = (d - a * dp) * inv;
        wc[p] = cp;
        wd[p] = dp;
    }
    double x = 0.0;
    for (int r = n - 1; r >= 0; --r) {
        const long p = base + (long)r * stride;
        x = wd[p] - wc[p] * x;
        out[p] = x;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: explicit stage.  Expression order is the reference's and FMA contraction is off, so R0 is
// bit-identical to the NumPy evaluation.  Reads the neighbour-flags byte (bit0 cell in mask, bits 1..6:
// the x-,x+,y-,y+,z-,z+ neighbour is in the mask) instead of seven mask bytes.
//
// k_explicit_v2: a thread owns two adjacent k-cells (16-byte accesses) and marches over JR consecutive
// j-rows with a three-row register window, so per cell pair it issues three dwordx4 loads (row j+1 and
// the i-1 / i+1 planes); the k-neighbours come from the adjacent lanes.  Tiles are ordered
// [j-slab][i][j-chunk][k-tile] and handed to XCDs in contiguous chunks: an XCD streams one j-slab plane by
// plane, so the i+-1 planes of a slab (3 x 256 KiB at 512^2) stay in that XCD's 4 MiB L2 and HBM sees each
// T line once.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lap_axis(bool lo, bool hi, double tlo, double thi, double t, double invdx2)
{
#pragma clang fp contract(off)
    double sacc = 0.0, cnt = 0.0;   // s = 0; if lower in mask: s += T_lo; c += 1; ... (adi3d_numba_coeff.py:246-253)
    if (lo) { sacc += tlo; cnt += 1.0; }
    if (hi) { sacc += thi; cnt += 1.0; }
    return (sacc - cnt * t) * invdx2;
}

constexpr int kExplicitJR = 8;

__global__ __launch_bounds__(256) void k_explicit_v2(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                     double *__restrict__ R0, int nx, int ny, int nz,
                                                     double invdx2, double f, int jslab, int ktiles, long ntiles)
{
#pragma clang fp contract(off)
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int jc_per_slab = (jslab + kExplicitJR - 1) / kExplicitJR;
    const long per_plane = (long)jc_per_slab * ktiles;
    const long per_slab = per_plane * nx;
    const int slab = (int)(tile / per_slab);
    long rem = tile - (long)slab * per_slab;
    const int i = (int)(rem / per_plane);
    rem -= (long)i * per_plane;
    const int jc = (int)(rem / ktiles), kt = (int)(rem - (long)jc * ktiles);
    const int jbeg = slab * jslab + jc * kExplicitJR;
    int jend = jbeg + kExplicitJR;
    if (jend > (slab + 1) * jslab) jend = (slab + 1) * jslab;
    if (jend > ny) jend = ny;
    const int k0 = kt * 512 + 2 * (int)threadIdx.x;
    const bool kin = k0 < nz;            // nz is even: both cells of the pair are inside
    const int lane = threadIdx.x & 63;
    const long sx = (long)ny * nz, sy = nz;
    if (jbeg >= jend) return;
    long p = (long)i * sx + (long)jbeg * sy + k0;
    const double2 zero2 = make_double2(0.0, 0.0);
    double2 tm = zero2, tc = zero2, tp = zero2;
    if (kin) {
        tc = *reinterpret_cast<const double2 *>(T + p);
        if (jbeg > 0) tm = *reinterpret_cast<const double2 *>(T + p - sy);
    }
    for (int j = jbeg; j < jend; ++j, p += sy) {
        unsigned fl = 0;
        double2 ti0 = zero2, ti1 = zero2;
        if (kin) {
            fl = *reinterpret_cast<const uint16_t *>(flags + p);
            if (j + 1 < ny) tp = *reinterpret_cast<const double2 *>(T + p + sy);
            if (i > 0) ti0 = *reinterpret_cast<const double2 *>(T + p - sx);
            if (i + 1 < nx) ti1 = *reinterpret_cast<const double2 *>(T + p + sx);
        }
        // k-neighbours of the pair: adjacent lanes, wave edges from memory
        double kl = __shfl_up(tc.y, 1), kr = __shfl_down(tc.x, 1);
        if (lane == 0) kl = (kin && k0 > 0) ? T[p - 1] : 0.0;
        if (lane == 63) kr = (kin && k0 + 2 < nz) ? T[p + 2] : 0.0;
        const unsigned f0 = fl & 0xffu, f1 = fl >> 8;
        double r0v, r1v;
        {
            double L0 = 0.0, L1 = 0.0, L2 = 0.0;
            if (f0 & 1u) {
                L0 = lap_axis(f0 & 2u, f0 & 4u, ti0.x, ti1.x, tc.x, invdx2);
                L1 = lap_axis(f0 & 8u, f0 & 16u, tm.x, tp.x, tc.x, invdx2);
                L2 = lap_axis(f0 & 32u, f0 & 64u, kl, tc.y, tc.x, invdx2);
            }
            r0v = tc.x + f * ((L0 + L1) + L2);
        }
        {
            double L0 = 0.0, L1 = 0.0, L2 = 0.0;
            if (f1 & 1u) {
                L0 = lap_axis(f1 & 2u, f1 & 4u, ti0.y, ti1.y, tc.y, invdx2);
                L1 = lap_axis(f1 & 8u, f1 & 16u, tm.y, tp.y, tc.y, invdx2);
                L2 = lap_axis(f1 & 32u, f1 & 64u, tc.x, kr, tc.y, invdx2);
            }
            r1v = tc.y + f * ((L0 + L1) + L2);
        }
        if (kin) *reinterpret_cast<double2 *>(R0 + p) = make_double2(r0v, r1v);
        tm = tc;
        tc = tp;
    }
}

// generic form (odd nz or unaligned views): one cell per thread
__global__ __launch_bounds__(256) void k_explicit(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                  double *__restrict__ R0, int nx, int ny, int nz,
                                                  double invdx2, double f)
{
#pragma clang fp contract(off)
    const long N = (long)nx * ny * nz;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const long sx = (long)ny * nz, sy = nz;
    const double t = T[p];
    const unsigned fl = flags[p];
    double L0 = 0.0, L1 = 0.0, L2 = 0.0;
    if (fl & 1u) {
        L0 = lap_axis(fl & 2u, fl & 4u, (fl & 2u) ? T[p - sx] : 0.0, (fl & 4u) ? T[p + sx] : 0.0, t, invdx2);
        L1 = lap_axis(fl & 8u, fl & 16u, (fl & 8u) ? T[p - sy] : 0.0, (fl & 16u) ? T[p + sy] : 0.0, t, invdx2);
        L2 = lap_axis(fl & 32u, fl & 64u, (fl & 32u) ? T[p - 1] : 0.0, (fl & 64u) ? T[p + 1] : 0.0, t, invdx2);
    }
    R0[p] = t + f * ((L0 + L1) + L2);
}

// neighbour flags: bit0 = cell in mask, bit(1 + 2*axis) / bit(2 + 2*axis) = the minus / plus neighbour along
// `axis` exists and is in the mask.  Derived from the mask whenever it changes (the mask "folds into the
// coefficient build on device"); halo planes of a slab decomposition are simply part of the mask array.
__global__ __launch_bounds__(256) void k_build_flags(const uint8_t *__restrict__ mask, int nx, int ny, int nz,
                                                     uint8_t *__restrict__ flags)
{
    const long N = (long)nx * ny * nz;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int k = (int)(p % nz);
    const long ij = p / nz;
    const int j = (int)(ij % ny), i = (int)(ij / ny);
    const long sx = (long)ny * nz, sy = nz;
    unsigned f = 0;
    if (mask[p]) {
        f = 1u;
        if (i > 0 && mask[p - sx]) f |= 2u;
        if (i + 1 < nx && mask[p + sx]) f |= 4u;
        if (j > 0 && mask[p - sy]) f |= 8u;
        if (j + 1 < ny && mask[p + sy]) f |= 16u;
        if (k > 0 && mask[p - 1]) f |= 32u;
        if (k + 1 < nz && mask[p + 1]) f |= 64u;
    }
    flags[p] = (uint8_t)f;
}

// ------------------------------------------------------------------------------------------------
// K0: coefficient build.  Same accumulation order as the reference ('-' face then '+' face per axis,
// (h * A) / Ccell with IEEE division), contraction off -> bit-identical packs.
// ------------------------------------------------------------------------------------------------
struct FaceSpec {
    int mode[6];
    double scalar[6];
    const double *field[6];
};

__global__ __launch_bounds__(256) void k_build_coeffs(const uint8_t *__restrict__ mask, int nx, int ny, int nz,
                                                      double A, double Ccell, FaceSpec h, FaceSpec q,
                                                      double *__restrict__ c0, double *__restrict__ c1,
                                                      double *__restrict__ c2, double *__restrict__ q0,
                                                      double *__restrict__ q1, double *__restrict__ q2)
{
#pragma clang fp contract(off)
    const long N = (long)nx * ny * nz;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int k = (int)(p % nz);
    const long ij = p / nz;
    const int j = (int)(ij % ny), i = (int)(ij / ny);
    const long st[3] = {(long)ny * nz, (long)nz, 1};
    const int pos[3] = {i, j, k}, nn[3] = {nx, ny, nz};
    const bool m = mask[p] != 0;
    double co[3] = {0.0, 0.0, 0.0}, qq[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int ax = f >> 1;
        const int nbp = pos[ax] + ((f & 1) ? 1 : -1);
        bool exposed = m;
        if (m && nbp >= 0 && nbp < nn[ax]) exposed = mask[p + ((f & 1) ? st[ax] : -st[ax])] == 0;
        if (exposed) {
            if (h.mode[f] != ADI_FACE_NONE) {
                const double hv = (h.mode[f] == ADI_FACE_SCALAR) ? h.scalar[f] : h.field[f][p];
                co[ax] += (hv * A / Ccell);
            }
            if (q.mode[f] != ADI_FACE_NONE) {
                const double qv = (q.mode[f] == ADI_FACE_SCALAR) ? q.scalar[f] : q.field[f][p];
                qq[ax] += (qv * A / Ccell);
            }
        }
    }
    c0[p] = co[0]; c1[p] = co[1]; c2[p] = co[2];
    q0[p] = qq[0]; q1[p] = qq[1]; q2[p] = qq[2];
}

__global__ __launch_bounds__(256) void k_exposed(const uint8_t *__restrict__ mask, int nx, int ny, int nz,
                                                 int face, uint8_t *__restrict__ out)
{
    const long N = (long)nx * ny * nz;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int k = (int)(p % nz);
    const long ij = p / nz;
    const int j = (int)(ij % ny), i = (int)(ij / ny);
    const long st[3] = {(long)ny * nz, (long)nz, 1};
    const int pos[3] = {i, j, k}, nn[3] = {nx, ny, nz};
    const int ax = face >> 1;
    const bool m = mask[p] != 0;
    const int nbp = pos[ax] + ((face & 1) ? 1 : -1);
    bool e = m;
    if (m && nbp >= 0 && nbp < nn[ax]) e = mask[p + ((face & 1) ? st[ax] : -st[ax])] == 0;
    out[p] = e ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_masked_fill(double *__restrict__ T, const uint8_t *__restrict__ sel,
                                                     size_t n, double v)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && sel[p]) T[p] = v;
}

__global__ __launch_bounds__(256) void k_mask_or(uint8_t *__restrict__ dst, const uint8_t *__restrict__ a,
                                                 const uint8_t *__restrict__ b, size_t n)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) dst[p] = (a[p] || b[p]) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------------
// lines per strided tile (8 B * lines contiguous per row).  16 = whole 128-byte lines; ADI_STRIDED_LINES
// overrides for tuning runs.
static int strided_lines_pref()
{
    static int v = 0;
    if (v == 0) {
        const char *e = getenv("ADI_STRIDED_LINES");
        v = e ? atoi(e) : 8;
        if (v != 8 && v != 16) v = 8;
    }
    return v;
}

static int contig_rows_per_lane(int n) { return n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : 16)); }
static int strided_rows_per_thread(int n) { return n <= 16 ? 2 : (n <= 32 ? 4 : (n <= 512 ? 8 : 16)); }

template <int M, bool HAS_DIR, bool HAS_Q>
static void launch_contig(const double *in, const uint8_t *mask, const double *coeff, const uint8_t *dmask,
                          const double *dval, const double *qf, double *out, long nlines, int n, SweepScal s,
                          hipStream_t st)
{
    const int Lp = next_pow2((n + M - 1) / M);
    const int lw = 64 / Lp;
    const long waves = (nlines + lw - 1) / lw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    const bool aligned = (((uintptr_t)in | (uintptr_t)coeff | (uintptr_t)out | (uintptr_t)dval | (uintptr_t)qf) & 15) == 0 &&
                         (((uintptr_t)mask | (uintptr_t)dmask) & 7) == 0;
    const bool vec = aligned && (n % M == 0);
    if (vec)
        hipLaunchKernelGGL((k_sweep_contig<M, true, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, mask, coeff,
                           dmask, dval, qf, out, nlines, n, Lp, s);
    else
        hipLaunchKernelGGL((k_sweep_contig<M, false, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, mask, coeff,
                           dmask, dval, qf, out, nlines, n, Lp, s);
}

template <int M, bool HAS_DIR, bool HAS_Q>
static void launch_strided(const double *in, const uint8_t *mask, const double *coeff, const uint8_t *dmask,
                           const double *dval, const double *qf, double *out, int n, long stride, int n_inner,
                           long n_outer, long outer_stride, int lbit, SweepScal s, hipStream_t st)
{
    const int Lp = next_pow2((n + M - 1) / M);
    int lines = (M <= 8) ? strided_lines_pref() : 8;  // M = 16 keeps 512-thread workgroups (register budget)
    while (lines * Lp < 256) lines <<= 1;
    const int tiles_inner = (n_inner + lines - 1) / lines;
    const long ntiles = (long)tiles_inner * n_outer;
    const size_t lds = (size_t)8 * lines * (Lp + 1) * sizeof(double);
    hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q>), dim3((unsigned)ntiles), dim3(lines * Lp), lds, st, in,
                       mask, coeff, dmask, dval, qf, out, n, stride, n_inner, outer_stride, Lp, lines, tiles_inner,
                       ntiles, lbit, s);
}

template <bool HAS_DIR, bool HAS_Q>
static int sweep_dispatch(int axis, const double *in, const uint8_t *mask, const double *coeff, const uint8_t *dmask,
                          const double *dval, const double *qf, int nx, int ny, int nz, SweepScal s, double *out,
                          void *work, size_t work_bytes, hipStream_t st)
{
    const int nn[3] = {nx, ny, nz};
    const int n = nn[axis];
    const long N = (long)nx * ny * nz;
    if (n > kMaxFastLine) {
        if (work == nullptr || work_bytes < (size_t)2 * N * sizeof(double))
            return set_err(ADI_ERR_ARG, "adi_sweep: line length %d > %d needs a workspace of %zu bytes", n,
                           kMaxFastLine, (size_t)2 * N * sizeof(double));
        double *wc = (double *)work, *wd = wc + N;
        long n_inner, inner_stride, n_outer, outer_stride, stride;
        if (axis == 0) { n_inner = (long)ny * nz; inner_stride = 1; n_outer = 1; outer_stride = 0; stride = (long)ny * nz; }
        else if (axis == 1) { n_inner = nz; inner_stride = 1; n_outer = nx; outer_stride = (long)ny * nz; stride = nz; }
        else { n_inner = (long)nx * ny; inner_stride = nz; n_outer = 1; outer_stride = 0; stride = 1; }
        const long nl = n_inner * n_outer;
        hipLaunchKernelGGL((k_sweep_generic<HAS_DIR, HAS_Q>), dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, in,
                           mask, coeff, dmask, dval, qf, out, n, stride, n_inner, inner_stride, n_outer, outer_stride,
                           1 + 2 * axis, wc, wd, s);
        return ADI_OK;
    }
    if (axis == 2) {
        const long nlines = (long)nx * ny;
        switch (contig_rows_per_lane(n)) {
            case 2: launch_contig<2, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, nlines, n, s, st); break;
            case 4: launch_contig<4, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, nlines, n, s, st); break;
            case 8: launch_contig<8, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, nlines, n, s, st); break;
            default: launch_contig<16, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, nlines, n, s, st); break;
        }
    } else {
        const long stride = (axis == 0) ? (long)ny * nz : nz;
        const long n_inner_l = (axis == 0) ? (long)ny * nz : nz;
        const long n_outer = (axis == 0) ? 1 : nx;
        const long outer_stride = (axis == 0) ? 0 : (long)ny * nz;
        if (n_inner_l > 0x7fffffffL) return set_err(ADI_ERR_UNSUPPORTED, "adi_sweep: plane too large");
        const int n_inner = (int)n_inner_l;
        switch (strided_rows_per_thread(n)) {
            case 2: launch_strided<2, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, n, stride, n_inner, n_outer, outer_stride, 1 + 2 * axis, s, st); break;
            case 4: launch_strided<4, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, n, stride, n_inner, n_outer, outer_stride, 1 + 2 * axis, s, st); break;
            case 8: launch_strided<8, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, n, stride, n_inner, n_outer, outer_stride, 1 + 2 * axis, s, st); break;
            default: launch_strided<16, HAS_DIR, HAS_Q>(in, mask, coeff, dmask, dval, qf, out, n, stride, n_inner, n_outer, outer_stride, 1 + 2 * axis, s, st); break;
        }
    }
    return ADI_OK;
}

}  // namespace adi

using namespace adi;

extern "C" {

int adi_exposed_mask(const uint8_t *d_mask, int nx, int ny, int nz, int face, uint8_t *d_exposed, void *stream)
{
    ADI_REQUIRE(face >= 0 && face < 6, "bad face");  // ValueError("bad face"), adi3d_numba_coeff.py:54
    ADI_REQUIRE(d_mask && d_exposed && nx > 0 && ny > 0 && nz > 0, "adi_exposed_mask: bad argument");
    const long N = (long)nx * ny * nz;
    hipLaunchKernelGGL(k_exposed, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), d_mask, nx, ny,
                       nz, face, d_exposed);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_build_coeffs(const uint8_t *d_mask, int nx, int ny, int nz, double dx, double rho, double cp,
                     const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                     const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                     double *const *d_coeff, double *const *d_qflux, void *stream)
{
    ADI_REQUIRE(d_mask && h_mode && h_scalar && q_mode && q_scalar && d_coeff && d_qflux, "adi_build_coeffs: null argument");
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "adi_build_coeffs: bad shape");
    FaceSpec h, q;
    for (int f = 0; f < 6; ++f) {
        h.mode[f] = h_mode[f]; h.scalar[f] = h_scalar[f]; h.field[f] = d_h_field ? d_h_field[f] : nullptr;
        q.mode[f] = q_mode[f]; q.scalar[f] = q_scalar[f]; q.field[f] = d_q_field ? d_q_field[f] : nullptr;
        ADI_REQUIRE(h.mode[f] >= 0 && h.mode[f] <= 2 && q.mode[f] >= 0 && q.mode[f] <= 2, "adi_build_coeffs: bad face mode");
        ADI_REQUIRE(h.mode[f] != ADI_FACE_FIELD || h.field[f], "adi_build_coeffs: missing h field for face %d", f);
        ADI_REQUIRE(q.mode[f] != ADI_FACE_FIELD || q.field[f], "adi_build_coeffs: missing q field for face %d", f);
    }
    for (int a = 0; a < 3; ++a) ADI_REQUIRE(d_coeff[a] && d_qflux[a], "adi_build_coeffs: null output");
    // A = dx*dx, V = dx**3 (CPython float_pow -> libm pow), Ccell = rho*cp*V: adi3d_numba_coeff.py:66-68
    const double A = dx * dx, V = pow(dx, 3.0), Ccell = rho * cp * V;
    const long N = (long)nx * ny * nz;
    hipLaunchKernelGGL(k_build_coeffs, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), d_mask, nx,
                       ny, nz, A, Ccell, h, q, d_coeff[0], d_coeff[1], d_coeff[2], d_qflux[0], d_qflux[1], d_qflux[2]);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_build_nbr_flags(const uint8_t *d_mask, int nx, int ny, int nz, uint8_t *d_flags, void *stream)
{
    ADI_REQUIRE(d_mask && d_flags && nx > 0 && ny > 0 && nz > 0, "adi_build_nbr_flags: bad argument");
    const long N = (long)nx * ny * nz;
    hipLaunchKernelGGL(k_build_flags, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), d_mask, nx,
                       ny, nz, d_flags);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_explicit_rhs(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, double dx, double dt,
                     double kappa, double theta, double *d_R0, void *stream)
{
    ADI_REQUIRE(d_T && d_flags && d_R0 && nx > 0 && ny > 0 && nz > 0, "adi_explicit_rhs: bad argument");
    ADI_REQUIRE(d_T != d_R0, "adi_explicit_rhs: output aliases input");
    const double invdx2 = 1.0 / (dx * dx);
    const double f = dt * kappa * (1.0 - theta);
    const bool fast = (nz % 2 == 0) && ((((uintptr_t)d_T | (uintptr_t)d_R0) & 15) == 0) && (((uintptr_t)d_flags & 1) == 0);
    if (fast) {
        const int jslab = (ny + 7) / 8;
        const int nslab = (ny + jslab - 1) / jslab;
        const int ktiles = (nz + 511) / 512;
        const long ntiles = (long)nslab * nx * ((jslab + kExplicitJR - 1) / kExplicitJR) * ktiles;
        hipLaunchKernelGGL(k_explicit_v2, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                           nx, ny, nz, invdx2, f, jslab, ktiles, ntiles);
    } else {
        const long N = (long)nx * ny * nz;
        hipLaunchKernelGGL(k_explicit, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), d_T,
                           d_flags, d_R0, nx, ny, nz, invdx2, f);
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_sweep_workspace_bytes(int axis, int nx, int ny, int nz, size_t *bytes)
{
    ADI_REQUIRE(axis >= 0 && axis < 3 && bytes && nx > 0 && ny > 0 && nz > 0, "adi_sweep_workspace_bytes: bad argument");
    const int nn[3] = {nx, ny, nz};
    *bytes = nn[axis] > kMaxFastLine ? (size_t)2 * nx * ny * nz * sizeof(double) : 0;
    return ADI_OK;
}

int adi_sweep(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
              const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
              double theta, double gam, double dt, double Tinf, double *d_out, void *d_work, size_t work_bytes,
              void *stream)
{
    ADI_REQUIRE(axis >= 0 && axis < 3, "adi_sweep: bad axis %d", axis);
    ADI_REQUIRE(variant >= 0 && variant <= 3, "adi_sweep: bad variant %d", variant);
    ADI_REQUIRE(d_in && d_flags && d_coeff && d_out && nx > 0 && ny > 0 && nz > 0, "adi_sweep: bad argument");
    ADI_REQUIRE(d_in != d_out, "adi_sweep: output aliases input");
    const bool has_dir = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_Q);
    const bool has_q = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_DIR);
    ADI_REQUIRE(!has_dir || (d_dir_mask && d_dir_val), "adi_sweep: variant needs Dirichlet arrays");
    ADI_REQUIRE(!has_q || d_qflux, "adi_sweep: variant needs the flux array");
    SweepScal s;
    s.tg = theta * gam;
    s.dt = dt;
    s.Tinf = Tinf;
    hipStream_t st = as_stream(stream);
    int rc;
    if (has_dir && has_q) rc = sweep_dispatch<true, true>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, s, d_out, d_work, work_bytes, st);
    else if (has_q) rc = sweep_dispatch<false, true>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, d_qflux, nx, ny, nz, s, d_out, d_work, work_bytes, st);
    else if (has_dir) rc = sweep_dispatch<true, false>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, nullptr, nx, ny, nz, s, d_out, d_work, work_bytes, st);
    else rc = sweep_dispatch<false, false>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, nullptr, nx, ny, nz, s, d_out, d_work, work_bytes, st);
    if (rc != ADI_OK) return rc;
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_step(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b, const uint8_t *d_flags,
             const double *const *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
             const double *const *d_qflux, int variant, int nx, int ny, int nz, double dx, double rho, double cp,
             double k, double dt, double theta, double Tinf, void *d_work, size_t work_bytes, void *stream)
{
    ADI_REQUIRE(d_T_in && d_T_out && d_tmp_a && d_tmp_b && d_coeff, "adi_step: null argument");
    ADI_REQUIRE(d_tmp_a != d_tmp_b && d_tmp_a != d_T_in && d_tmp_b != d_T_in && d_T_out != d_tmp_a && d_T_out != d_T_in,
                "adi_step: buffers must be distinct (T_out may equal tmp_b only)");
    // kappa, gam: adi3d_numba_coeff.py:292
    const double kappa = k / (rho * cp);
    const double gam = kappa * dt / (dx * dx);
    const double *q0 = d_qflux ? d_qflux[0] : nullptr, *q1 = d_qflux ? d_qflux[1] : nullptr, *q2 = d_qflux ? d_qflux[2] : nullptr;
    int rc = adi_explicit_rhs(d_T_in, d_flags, nx, ny, nz, dx, dt, kappa, theta, d_tmp_a, stream);
    if (rc) return rc;
    rc = adi_sweep(0, variant, d_tmp_a, d_flags, d_coeff[0], d_dir_mask, d_dir_val, q0, nx, ny, nz, theta, gam, dt, Tinf, d_tmp_b, d_work, work_bytes, stream);
    if (rc) return rc;
    rc = adi_sweep(1, variant, d_tmp_b, d_flags, d_coeff[1], d_dir_mask, d_dir_val, q1, nx, ny, nz, theta, gam, dt, Tinf, d_tmp_a, d_work, work_bytes, stream);
    if (rc) return rc;
    return adi_sweep(2, variant, d_tmp_a, d_flags, d_coeff[2], d_dir_mask, d_dir_val, q2, nx, ny, nz, theta, gam, dt, Tinf, d_T_out, d_work, work_bytes, stream);
}

int adi_masked_fill(double *d_T, const uint8_t *d_sel, size_t n, double value, void *stream)
{
    ADI_REQUIRE(d_T && d_sel, "adi_masked_fill: null argument");
    if (n == 0) return ADI_OK;
    hipLaunchKernelGGL(k_masked_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), d_T, d_sel, n, value);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_mask_or(uint8_t *d_dst, const uint8_t *d_a, const uint8_t *d_b, size_t n, void *stream)
{
    ADI_REQUIRE(d_dst && d_a && d_b, "adi_mask_or: null argument");
    if (n == 0) return ADI_OK;
    hipLaunchKernelGGL(k_mask_or, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), d_dst, d_a, d_b, n);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

}  // extern "C"
