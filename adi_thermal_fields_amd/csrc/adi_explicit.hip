// adi_explicit.hip -- K0 / K1 and the byte kernels around them (hand-written HIP for gfx950, HBM-bound):
//   k_explicit_v5     the masked 7-point explicit stage -> R0 (lap1D_x/y/z + R0, adi3d_numba_coeff.py:240-288, :298),
//                     2.5-D marching form; <.., DOTS>: pass A of the slab decomposition folded in
//   k_explicit_cell   the same, one thread per cell (odd nz, unaligned views)
//   k_build_flags     the neighbour-flags digest of the mask every step kernel reads
//   k_build_coeffs    Robin coefficient + Neumann flux fields of the three axes in one pass
//                     (precompute_coeff_packs_unified, adi3d_numba_coeff.py:57-118)
//   k_exposed         exposed_mask (adi3d_numba_coeff.py:38-55)
#include <vector>

#include "adi_cart_host.hpp"

namespace adi {

template <int JT, bool DOTS = false>
__global__ __launch_bounds__(256) void k_explicit_v5(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                     double *__restrict__ R0, Lay L, double invdx2, double f,
                                                     int jslab, int ktiles, int ichunk, long ntiles, int i_begin,
                                                     int i_end, const double *__restrict__ wu = nullptr,
                                                     double *__restrict__ part = nullptr, int i_org = 0, int n_line = 0,
                                                     int nt = 1)
{
#pragma clang fp contract(off)
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    // tile order: [slab][i-chunk][j-tile in slab][k-tile]
    const int jt_per_slab = (jslab + JT - 1) / JT;
    const int nchunk = (i_end - i_begin + ichunk - 1) / ichunk;
    const long per_chunk = (long)jt_per_slab * ktiles;
    const long per_slab = per_chunk * nchunk;
    const unsigned t32 = (unsigned)tile, pch = (unsigned)per_chunk, psl = (unsigned)per_slab;
    const int slab = (int)(t32 / psl);
    unsigned rem = t32 - (unsigned)slab * psl;
    const int ic = (int)(rem / pch);
    rem -= (unsigned)ic * pch;
    const int jt = (int)(rem / (unsigned)ktiles), kt = (int)(rem - (unsigned)jt * (unsigned)ktiles);
    const int j0 = slab * jslab + jt * JT;
    int jend = j0 + JT;
    if (jend > (slab + 1) * jslab) jend = (slab + 1) * jslab;
    if (jend > ny) jend = ny;
    if (j0 >= jend) return;
    const int i0 = i_begin + ic * ichunk;
    const int i1 = (i0 + ichunk < i_end) ? i0 + ichunk : i_end;
    const int k0 = kt * 512 + 2 * (int)threadIdx.x;
    const bool kin = k0 < nz;
    const int lane = threadIdx.x & 63;
    const long sx = L.sx, sy = nz;
    const double2 zero2 = make_double2(0.0, 0.0);
    const long pbase = (long)j0 * sy + k0;
    // lane 0 / lane 63 fetch the value just outside the wave's k range (one load per row)
    const bool edge = kin && ((lane == 0 && k0 > 0) || (lane == 63 && k0 + 2 < nz));
    const long eoff = (lane == 0) ? -1 : 2;
    const bool up = kin && j0 > 0, dn = kin && jend < ny;

    double2 tm[JT], tc[JT], tp[JT], tq[JT];
    unsigned fl[JT], fln[JT];
    double ke[JT], ken[JT];
    double2 hm = zero2, hp = zero2, hmn = zero2, hpn = zero2;
    auto load_plane = [&](int i, double2 (&dst)[JT]) {
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            dst[r] = zero2;
            if (kin && j0 + r < jend && i >= 0 && i < nx)
                dst[r] = *reinterpret_cast<const double2 *>(T + (long)i * sx + pbase + (long)r * sy);
        }
    };
    auto load_meta = [&](int i, unsigned (&F)[JT], double (&E)[JT], double2 &HM, double2 &HP) {
        const long p = (long)i * sx + pbase;
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            F[r] = 0; E[r] = 0.0;
            if (kin && j0 + r < jend) F[r] = *reinterpret_cast<const uint16_t *>(flags + p + (long)r * sy);
            if (edge && j0 + r < jend) E[r] = T[p + (long)r * sy + eoff];
        }
        HM = zero2; HP = zero2;
        if (up) HM = *reinterpret_cast<const double2 *>(T + p - sy);
        if (dn) HP = *reinterpret_cast<const double2 *>(T + p + (long)(jend - j0) * sy);
    };
    load_plane(i0 - 1, tm);
    load_plane(i0, tc);
    load_plane(i0 + 1, tp);
    load_meta(i0, fl, ke, hm, hp);
    double2 su[DOTS ? JT : 1], sv[DOTS ? JT : 1];
    if (DOTS) {
#pragma unroll
        for (int r = 0; r < JT; ++r) { su[r] = zero2; sv[r] = zero2; }
    }
    // DOTS: lines start at plane i_org and have n_line rows; this launch covers whole chunks of them
    for (int i = i0; i < i1; ++i) {
        const long p = (long)i * sx + pbase;
        const bool more = i + 1 < i1;
        double wa = 0.0, wb = 0.0;
        if (DOTS) { wa = wu[i - i_org]; wb = wu[n_line - 1 - (i - i_org)]; }          // block-uniform: scalar loads
        if (more) {
            load_plane(i + 2, tq);
            load_meta(i + 1, fln, ken, hmn, hpn);
        }
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            const bool rin = j0 + r < jend;
            const long q = p + (long)r * sy;
            double kl = __shfl_up(tc[r].y, 1), kr = __shfl_down(tc[r].x, 1);
            if (lane == 0) kl = ke[r];
            if (lane == 63) kr = ke[r];
            const double2 jm = (r == 0) ? hm : tc[r > 0 ? r - 1 : 0];
            const double2 jp = (j0 + r + 1 == jend) ? hp : tc[r + 1 < JT ? r + 1 : JT - 1];
            const unsigned f0 = fl[r] & 0xffu, f1 = fl[r] >> 8;
            double r0v, r1v;
            {
                double L0 = 0.0, L1 = 0.0, L2 = 0.0;
                if (f0 & 1u) {
                    L0 = lap_axis(f0 & 2u, f0 & 4u, tm[r].x, tp[r].x, tc[r].x, invdx2);
                    L1 = lap_axis(f0 & 8u, f0 & 16u, jm.x, jp.x, tc[r].x, invdx2);
                    L2 = lap_axis(f0 & 32u, f0 & 64u, kl, tc[r].y, tc[r].x, invdx2);
                }
                r0v = tc[r].x + f * ((L0 + L1) + L2);
            }
            {
                double L0 = 0.0, L1 = 0.0, L2 = 0.0;
                if (f1 & 1u) {
                    L0 = lap_axis(f1 & 2u, f1 & 4u, tm[r].y, tp[r].y, tc[r].y, invdx2);
                    L1 = lap_axis(f1 & 8u, f1 & 16u, jm.y, jp.y, tc[r].y, invdx2);
                    L2 = lap_axis(f1 & 32u, f1 & 64u, tc[r].x, kr, tc[r].y, invdx2);
                }
                r1v = tc[r].y + f * ((L0 + L1) + L2);
            }
            if (kin && rin) st_stream2(reinterpret_cast<double2 *>(R0 + q), make_double2(r0v, r1v), nt != 0);
            if (DOTS) {
                su[r].x = __builtin_fma(wa, r0v, su[r].x); su[r].y = __builtin_fma(wa, r1v, su[r].y);
                sv[r].x = __builtin_fma(wb, r0v, sv[r].x); sv[r].y = __builtin_fma(wb, r1v, sv[r].y);
            }
        }
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            tm[r] = tc[r]; tc[r] = tp[r]; tp[r] = tq[r];
            fl[r] = fln[r]; ke[r] = ken[r];
        }
        hm = hmn; hp = hpn;
    }
    if (DOTS) {
        const long nlines = (long)ny * nz;
        double *pu = part + (long)((i_begin - i_org) / ichunk + ic) * 2 * nlines, *pv = pu + nlines;   // global chunk id
#pragma unroll
        for (int r = 0; r < JT; ++r)
            if (kin && j0 + r < jend) {
                const long line = (long)(j0 + r) * nz + k0;
                *reinterpret_cast<double2 *>(pu + line) = su[r];
                *reinterpret_cast<double2 *>(pv + line) = sv[r];
            }
    }
}

// ---- pass A from the dot products (slab decomposition) -----------------------------------------------------------------
// A line is "uniform" for the axis-0 sweep when its rows 1..n-2 are in the mask with both axis neighbours and are not
// Dirichlet, its end rows are in the mask (not Dirichlet) with their inward neighbour, and at most one end row differs
// from the interior row (line start/end, Robin coefficient).  cls[line] = 1 for those; the others are appended to
// list[1..] (list[0] = count) and condensed by k_condense_generic from the stored R0.
__global__ __launch_bounds__(256) void k_classify_lines0(const uint8_t *__restrict__ flags,
                                                         const uint8_t *__restrict__ dmask, Lay L,
                                                         uint8_t *__restrict__ cls, unsigned *__restrict__ list)
{
    const long nlines = (long)L.ny * L.nz;
    const long lid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= nlines) return;
    const int n = L.nx;
    bool ok = n >= 2;
    unsigned f0 = 0, fn = 0;
    for (int r = 0; r < n; ++r) {
        const long p = (long)r * L.sx + lid;
        const unsigned f = flags[p];
        if (dmask != nullptr && dmask[p] != 0) ok = false;
        if (r == 0) { f0 = f; ok = ok && ((f & 5u) == 5u); }                 // in mask, next row in mask
        else if (r == n - 1) { fn = f; ok = ok && ((f & 3u) == 3u); }        // in mask, previous row in mask
        else ok = ok && ((f & 7u) == 7u);
    }
    // an end row without its outward neighbour is a modified row (b = 1 + tg + dt*coeff): at most one per line
    if (ok && !(f0 & 2u) && !(fn & 4u)) ok = false;
    cls[lid] = ok ? 1 : 0;
    if (!ok) list[1 + atomicAdd(&list[0], 1u)] = (unsigned)lid;
}

// cond[6][nsel] of the lines [lb, le) from the partial dot products (uniform lines only; the others keep what
// k_condense_generic wrote).  Formulas: condense_uniform (adi_core.hpp) applied to the whole line.
template <bool HAS_Q>
__global__ __launch_bounds__(256) void k_dots_finish(const double *__restrict__ part, int nchunk,
                                                     const double *__restrict__ wu, const uint8_t *__restrict__ cls,
                                                     const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
                                                     const double *__restrict__ qf, Lay L, SweepScal s, long lb, long le,
                                                     double *__restrict__ cond)
{
    const long nlines = (long)L.ny * L.nz, nsel = le - lb;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nsel) return;
    const long lid = lb + id;
    if (!cls[lid]) return;
    const int n = L.nx;
    double gu = 0.0, gv = 0.0;
    for (int c = 0; c < nchunk; ++c) {                      // fixed order: deterministic
        gu += part[(long)c * 2 * nlines + lid];
        gv += part[((long)c * 2 + 1) * nlines + lid];
    }
    const long pF = lid, pL = (long)(n - 1) * L.sx + lid;
    const unsigned fF = flags[pF], fL = flags[pL];
    const bool loF = (fF & 2u) != 0, hiL = (fL & 4u) != 0;   // the line continues below / above the slab
    const double p0 = wu[0], pn = wu[n - 1];
    const double bu = 1.0 + 2.0 * s.tg;
    // right-hand side terms of the end rows beyond R0 (assemble_row): dt*q + dt*coeff*Tinf on axis-exposed cells
    const double coF = loF ? 0.0 : coeff[pF], coL = hiL ? 0.0 : coeff[pL];
    double xF = s.dt * coF * s.Tinf, xL = s.dt * coL * s.Tinf;
    if (HAS_Q) { if (!loF) xF += s.dt * qf[pF]; if (!hiL) xL += s.dt * qf[pL]; }
    gu += p0 * xF + pn * xL;                                 // (U^-1 d)_0
    gv += pn * xF + p0 * xL;                                 // (U^-1 d)_{n-1}
    const double a0 = loF ? -s.tg : 0.0, cn = hiL ? -s.tg : 0.0;
    double gF, aF, cF, gL, aL, cL;
    if (!loF) {            // row 0 modified: b0 = 1 + tg + dt*coF
        const double delta = (1.0 + s.tg + s.dt * coF) - bu;
        const double kappa = delta / (1.0 + delta * p0);
        const double f1 = 1.0 - kappa * p0, kpl = kappa * pn;
        gF = gu * f1;            gL = gv - kpl * gu;
        aF = 0.0;                aL = 0.0;
        cF = cn * (pn - kpl * p0); cL = cn * (p0 - kpl * pn);
    } else if (!hiL) {     // row n-1 modified
        const double delta = (1.0 + s.tg + s.dt * coL) - bu;
        const double kappa = delta / (1.0 + delta * p0);
        const double f1 = 1.0 - kappa * p0, kpl = kappa * pn;
        gL = gv * f1;            gF = gu - kpl * gv;
        cL = 0.0;                cF = 0.0;
        aL = a0 * (pn - kpl * p0); aF = a0 * (p0 - kpl * pn);
    } else {
        gF = gu; gL = gv;
        aF = a0 * p0; cF = cn * pn; aL = a0 * pn; cL = cn * p0;
    }
    cond[id] = gF; cond[nsel + id] = aF; cond[2 * nsel + id] = cF;
    cond[3 * nsel + id] = gL; cond[4 * nsel + id] = aL; cond[5 * nsel + id] = cL;
}

// cell index -> (i, j, k, memory offset) for elementwise kernels over a padded-plane layout
__device__ __forceinline__ bool cell_of(long q, const Lay &L, int &i, int &j, int &k, long &p)
{
    const long plane = (long)L.ny * L.nz;
    if (q >= plane * L.nx) return false;
    i = (int)(q / plane);
    const long r = q - (long)i * plane;
    j = (int)(r / L.nz);
    k = (int)(r - (long)j * L.nz);
    p = (long)i * L.sx + r;
    return true;
}

// the same restricted to the planes [k0, k1) of axis 2 (the planes a layer birth touches)
__device__ __forceinline__ bool cell_of_k(long q, const Lay &L, int k0, int k1, int &i, int &j, int &k, long &p)
{
    const int nk = k1 - k0;
    if (q >= (long)L.nx * L.ny * nk) return false;
    const long row = q / nk;
    k = k0 + (int)(q - row * nk);
    i = (int)(row / L.ny);
    j = (int)(row - (long)i * L.ny);
    p = (long)i * L.sx + (long)j * L.nz + k;
    return true;
}

// generic form (odd nz or unaligned views): one cell per thread
__global__ __launch_bounds__(256) void k_explicit_cell(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                  double *__restrict__ R0, Lay L, double invdx2, double f, int i_begin,
                                                  int i_end)
{
#pragma clang fp contract(off)
    int i, j, k;
    long p;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x + (long)i_begin * L.ny * L.nz;
    if (q >= (long)i_end * L.ny * L.nz) return;
    if (!cell_of(q, L, i, j, k, p)) return;
    const long sx = L.sx, sy = L.nz;
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
__global__ __launch_bounds__(256) void k_build_flags(const uint8_t *__restrict__ mask, Lay L, int k0, int k1,
                                                     uint8_t *__restrict__ flags)
{
    int i, j, k;
    long p;
    if (!cell_of_k((long)blockIdx.x * blockDim.x + threadIdx.x, L, k0, k1, i, j, k, p)) return;
    const long sx = L.sx, sy = L.nz;
    unsigned f = 0;
    if (mask[p]) {
        f = 1u;
        if (i > 0 && mask[p - sx]) f |= 2u;
        if (i + 1 < L.nx && mask[p + sx]) f |= 4u;
        if (j > 0 && mask[p - sy]) f |= 8u;
        if (j + 1 < L.ny && mask[p + sy]) f |= 16u;
        if (k > 0 && mask[p - 1]) f |= 32u;
        if (k + 1 < L.nz && mask[p + 1]) f |= 64u;
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

__global__ __launch_bounds__(256) void k_build_coeffs(const uint8_t *__restrict__ mask, Lay L, int k0, int k1, double A,
                                                      double Ccell, FaceSpec h, FaceSpec q, double *__restrict__ c0,
                                                      double *__restrict__ c1, double *__restrict__ c2,
                                                      double *__restrict__ q0, double *__restrict__ q1,
                                                      double *__restrict__ q2)
{
#pragma clang fp contract(off)
    int i, j, k;
    long p;
    if (!cell_of_k((long)blockIdx.x * blockDim.x + threadIdx.x, L, k0, k1, i, j, k, p)) return;
    const long st[3] = {L.sx, (long)L.nz, 1};
    const int pos[3] = {i, j, k}, nn[3] = {L.nx, L.ny, L.nz};
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

__global__ __launch_bounds__(256) void k_exposed(const uint8_t *__restrict__ mask, Lay L, int face,
                                                 uint8_t *__restrict__ out)
{
    int i, j, k;
    long p;
    if (!cell_of((long)blockIdx.x * blockDim.x + threadIdx.x, L, i, j, k, p)) return;
    const long st[3] = {L.sx, (long)L.nz, 1};
    const int pos[3] = {i, j, k}, nn[3] = {L.nx, L.ny, L.nz};
    const int ax = face >> 1;
    const bool m = mask[p] != 0;
    const int nbp = pos[ax] + ((face & 1) ? 1 : -1);
    bool e = m;
    if (m && nbp >= 0 && nbp < nn[ax]) e = mask[p + ((face & 1) ? st[ax] : -st[ax])] == 0;
    out[p] = e ? 1 : 0;
}

// Exposed lateral faces per plane k of axis 2, from the neighbour-flags digest: an in-mask cell exposes one face for every
// neighbour bit in `bits` that is clear (bits 1..6 = x-, x+, y-, y+, z-, z+; the domain boundary counts as exposed, as in
// exposed_mask).  This is the count loop of quick_compare_layer_birth_robin_v3.py:97-108 for every layer at once.
// Thread t of a block owns plane k = blockIdx.x*256 + t and walks a chunk of (i, j) rows: coalesced along k, one
// 64-bit atomic per thread at the end.
__global__ __launch_bounds__(256) void k_count_exposed(const uint8_t *__restrict__ flags, Lay L, unsigned bits, int rows_per_block,
                                                       unsigned long long *__restrict__ counts)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= L.nz) return;
    const long nrows = (long)L.nx * L.ny;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < nrows ? r0 + rows_per_block : nrows;
    unsigned long long c = 0;
    for (long r = r0; r < r1; ++r) {
        const unsigned i = (unsigned)r / (unsigned)L.ny, j = (unsigned)r - i * (unsigned)L.ny;
        const unsigned f = flags[(long)i * L.sx + (long)j * L.nz + k];
        if (f & 1u) c += __popc(~f & bits & 0x7eu);
    }
    if (c) atomicAdd(&counts[k], c);
}

// A birth in one pass (activate_layer, waam_from_stl_v7_mm.py:487-495): on the planes [k0, k1) of axis 2
//     newborn = full & ~active;   T[newborn] = Ts;   active |= full          and the number of newborn cells is counted.
// Cell q of the (nx*ny) x (k1-k0) box: consecutive threads run along k.
__global__ __launch_bounds__(256) void k_birth(double *__restrict__ T, uint8_t *__restrict__ active,
                                               const uint8_t *__restrict__ full, Lay L, int k0, int k1, double Ts,
                                               unsigned long long *__restrict__ n_new)
{
    const int nk = k1 - k0;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)L.nx * L.ny * nk;
    unsigned born = 0;
    if (q < total) {
        const long row = q / nk;
        const int k = k0 + (int)(q - row * nk);
        const unsigned i = (unsigned)row / (unsigned)L.ny, j = (unsigned)row - i * (unsigned)L.ny;
        const long p = (long)i * L.sx + (long)j * L.nz + k;
        if (full[p] != 0 && active[p] == 0) { T[p] = Ts; active[p] = 1; born = 1; }
    }
    const unsigned long long b = __ballot(born);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_new, (unsigned long long)__popcll(b));
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


}  // namespace adi

using namespace adi;

extern "C" {

int adi_exposed_mask(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, int face, uint8_t *d_exposed,
                     void *stream)
{
    ADI_REQUIRE(face >= 0 && face < 6, "bad face");  // ValueError("bad face"), adi3d_numba_coeff.py:54
    ADI_REQUIRE(d_mask && d_exposed, "adi_exposed_mask: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    hipLaunchKernelGGL(k_exposed, dim3(cell_blocks(L)), dim3(256), 0, as_stream(stream), d_mask, L, face, d_exposed);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_count_exposed_faces(const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride, int face_bits,
                            unsigned long long *d_counts, void *stream)
{
    ADI_REQUIRE(d_flags && d_counts, "adi_count_exposed_faces: null argument");
    ADI_REQUIRE(face_bits > 0 && face_bits < 64, "adi_count_exposed_faces: face_bits selects none of the six faces");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    hipStream_t st = as_stream(stream);
    ADI_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nz * sizeof(unsigned long long), st));
    const long nrows = (long)nx * ny;
    const int rows_per_block = nrows > 4096 ? 256 : 16;
    const dim3 grid((unsigned)((nz + 255) / 256), (unsigned)((nrows + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(k_count_exposed, grid, dim3(256), 0, st, d_flags, L, (unsigned)face_bits << 1, rows_per_block, d_counts);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_birth_planes(double *d_T, uint8_t *d_active, const uint8_t *d_full, int nx, int ny, int nz, long plane_stride,
                     int k_begin, int k_end, double Ts, unsigned long long *d_newborn, void *stream)
{
    ADI_REQUIRE(d_T && d_active && d_full && d_newborn, "adi_birth_planes: null argument");
    ADI_REQUIRE(d_active != d_full, "adi_birth_planes: the active mask aliases the full mask");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    ADI_REQUIRE(k_begin >= 0 && k_end <= nz && k_begin <= k_end, "adi_birth_planes: bad plane range [%d, %d)", k_begin, k_end);
    hipStream_t st = as_stream(stream);
    ADI_HIP_TRY(hipMemsetAsync(d_newborn, 0, sizeof(unsigned long long), st));
    if (k_begin == k_end) return ADI_OK;
    const long total = (long)nx * ny * (k_end - k_begin);
    hipLaunchKernelGGL(k_birth, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_T, d_active, d_full, L, k_begin, k_end,
                       Ts, d_newborn);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

static unsigned range_blocks(const Lay &L, int k0, int k1) { return (unsigned)(((long)L.nx * L.ny * (k1 - k0) + 255) / 256); }

int adi_build_coeffs_planes(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, double dx, double rho,
                            double cp, const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                            const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                            double *const *d_coeff, double *const *d_qflux, int k_begin, int k_end, void *stream)
{
    ADI_REQUIRE(d_mask && h_mode && h_scalar && q_mode && q_scalar && d_coeff && d_qflux, "adi_build_coeffs: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    ADI_REQUIRE(k_begin >= 0 && k_end <= nz && k_begin <= k_end, "adi_build_coeffs_planes: bad plane range [%d, %d)", k_begin, k_end);
    FaceSpec h, q;
    for (int f = 0; f < 6; ++f) {
        h.mode[f] = h_mode[f]; h.scalar[f] = h_scalar[f]; h.field[f] = d_h_field ? d_h_field[f] : nullptr;
        q.mode[f] = q_mode[f]; q.scalar[f] = q_scalar[f]; q.field[f] = d_q_field ? d_q_field[f] : nullptr;
        ADI_REQUIRE(h.mode[f] >= 0 && h.mode[f] <= 2 && q.mode[f] >= 0 && q.mode[f] <= 2, "adi_build_coeffs: bad face mode");
        ADI_REQUIRE(h.mode[f] != ADI_FACE_FIELD || h.field[f], "adi_build_coeffs: missing h field for face %d", f);
        ADI_REQUIRE(q.mode[f] != ADI_FACE_FIELD || q.field[f], "adi_build_coeffs: missing q field for face %d", f);
    }
    for (int a = 0; a < 3; ++a) ADI_REQUIRE(d_coeff[a] && d_qflux[a], "adi_build_coeffs: null output");
    if (k_begin == k_end) return ADI_OK;
    // A = dx*dx, V = dx**3 (CPython float_pow -> libm pow), Ccell = rho*cp*V: adi3d_numba_coeff.py:66-68
    const double A = dx * dx, V = pow(dx, 3.0), Ccell = rho * cp * V;
    hipLaunchKernelGGL(k_build_coeffs, dim3(range_blocks(L, k_begin, k_end)), dim3(256), 0, as_stream(stream), d_mask, L,
                       k_begin, k_end, A, Ccell, h, q, d_coeff[0], d_coeff[1], d_coeff[2], d_qflux[0], d_qflux[1], d_qflux[2]);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_face_constants(double dx, double rho, double cp, const int *h_mode, const double *h_scalar, const int *q_mode,
                       const double *q_scalar, double *h_consts, int *h_valid)
{
#pragma clang fp contract(off)
    ADI_REQUIRE(h_mode && h_scalar && q_mode && q_scalar && h_consts && h_valid, "adi_face_constants: null argument");
    // the very expressions of adi_build_coeffs_planes / k_build_coeffs: (h * A) / Ccell, IEEE, no contraction
    const double A = dx * dx, V = pow(dx, 3.0), Ccell = rho * cp * V;
    for (int a = 0; a < 3; ++a) {
        bool ok = true;
        for (int s = 0; s < 2; ++s) {
            const int f = 2 * a + s;
            ADI_REQUIRE(h_mode[f] >= 0 && h_mode[f] <= 2 && q_mode[f] >= 0 && q_mode[f] <= 2, "adi_face_constants: bad face mode");
            ok = ok && h_mode[f] != ADI_FACE_FIELD && q_mode[f] != ADI_FACE_FIELD;
            h_consts[4 * a + s] = (h_mode[f] == ADI_FACE_SCALAR) ? (h_scalar[f] * A / Ccell) : 0.0;
            h_consts[4 * a + 2 + s] = (q_mode[f] == ADI_FACE_SCALAR) ? (q_scalar[f] * A / Ccell) : 0.0;
        }
        h_valid[a] = ok ? 1 : 0;
    }
    return ADI_OK;
}

int adi_build_coeffs(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, double dx, double rho,
                     double cp, const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                     const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                     double *const *d_coeff, double *const *d_qflux, void *stream)
{
    return adi_build_coeffs_planes(d_mask, nx, ny, nz, plane_stride, dx, rho, cp, h_mode, h_scalar, d_h_field, q_mode,
                                   q_scalar, d_q_field, d_coeff, d_qflux, 0, nz, stream);
}

int adi_build_nbr_flags_planes(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, uint8_t *d_flags,
                               int k_begin, int k_end, void *stream)
{
    ADI_REQUIRE(d_mask && d_flags, "adi_build_nbr_flags: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    ADI_REQUIRE(k_begin >= 0 && k_end <= nz && k_begin <= k_end, "adi_build_nbr_flags_planes: bad plane range [%d, %d)", k_begin, k_end);
    if (k_begin == k_end) return ADI_OK;
    hipLaunchKernelGGL(k_build_flags, dim3(range_blocks(L, k_begin, k_end)), dim3(256), 0, as_stream(stream), d_mask, L, k_begin,
                       k_end, d_flags);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_build_nbr_flags(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, uint8_t *d_flags,
                        void *stream)
{
    return adi_build_nbr_flags_planes(d_mask, nx, ny, nz, plane_stride, d_flags, 0, nz, stream);
}

int adi_explicit_rhs_planes(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride,
                            double dx, double dt, double kappa, double theta, double *d_R0, int i_begin, int i_end,
                            void *stream)
{
    ADI_REQUIRE(d_T && d_flags && d_R0, "adi_explicit_rhs: null argument");
    ADI_REQUIRE(d_T != d_R0, "adi_explicit_rhs: output aliases input");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    ADI_REQUIRE(i_begin >= 0 && i_end <= nx && i_begin <= i_end, "adi_explicit_rhs_planes: bad plane range [%d, %d)",
                i_begin, i_end);
    if (i_begin == i_end) return ADI_OK;
    const double invdx2 = 1.0 / (dx * dx);
    const double f = dt * kappa * (1.0 - theta);
    const int np = i_end - i_begin;
    const bool fast = (nz % 2 == 0) && (L.sx % 2 == 0) && ((((uintptr_t)d_T | (uintptr_t)d_R0) & 15) == 0) &&
                      (((uintptr_t)d_flags & 1) == 0);
    if (fast) {
        const int jslab = (ny + 7) / 8;
        const int nslab = (ny + jslab - 1) / jslab;
        const int ktiles = (nz + 511) / 512;
        // planes marched per block: 32 amortises the leading halo plane; short plane ranges (the boundary windows of a
        // slab) get shorter chunks so that the launch still has ~16 chunks' worth of blocks
        const int ichunk = dots_ichunk(np);
        const int nchunk = (np + ichunk - 1) / ichunk;
        const long ntiles = (long)nslab * nchunk * ((jslab + 1) / 2) * ktiles;
        hipLaunchKernelGGL(k_explicit_v5<2>, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                           L, invdx2, f, jslab, ktiles, ichunk, ntiles, i_begin, i_end, (const double *)nullptr,
                           (double *)nullptr, 0, 0, store_policy_nt(L.nx, L.sx));
    } else {    // odd nz / unaligned views: one thread per cell
        const long cells = (long)np * ny * nz;
        hipLaunchKernelGGL(k_explicit_cell, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, as_stream(stream), d_T,
                           d_flags, d_R0, L, invdx2, f, i_begin, i_end);
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_explicit_rhs(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride, double dx,
                     double dt, double kappa, double theta, double *d_R0, void *stream)
{
    return adi_explicit_rhs_planes(d_T, d_flags, nx, ny, nz, plane_stride, dx, dt, kappa, theta, d_R0, 0, nx, stream);
}

// ---- pass A folded into the explicit stage (slab decomposition) ------------------------------------------------------
int adi_axis0_dots_supported(int nx, int ny, int nz, long plane_stride)
{
    Lay L;
    if (make_lay(nx, ny, nz, plane_stride, &L) != ADI_OK) return 0;
    return (nx >= 2 && nz % 2 == 0 && L.sx % 2 == 0) ? 1 : 0;      // the marching explicit kernel's own conditions
}

int adi_axis0_dots_workspace(int nx, int ny, int nz, size_t *part_bytes, size_t *list_bytes)
{
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && part_bytes && list_bytes, "adi_axis0_dots_workspace: bad argument");
    const int ich = dots_ichunk(nx);
    const long nchunk = (nx + ich - 1) / ich;
    *part_bytes = (size_t)nchunk * 2 * (size_t)ny * nz * sizeof(double);
    *list_bytes = ((size_t)ny * nz + 1) * sizeof(unsigned);
    return ADI_OK;
}

int adi_axis0_dots_setup(int n, double theta, double gam, double *d_weights, void *stream)
{
    ADI_REQUIRE(n >= 2 && d_weights, "adi_axis0_dots_setup: bad argument");
    // u = first column of tridiag(-tg, 1+2tg, -tg)^-1 (n x n): Thomas on e_0 in long double
    const long double tg = (long double)theta * (long double)gam, b = 1.0L + 2.0L * tg;
    std::vector<long double> cp(n), x(n);
    std::vector<double> u(n);
    long double piv = b;
    cp[0] = -tg / piv; x[0] = 1.0L / piv;
    for (int i = 1; i < n; ++i) {
        piv = b + tg * cp[i - 1];
        cp[i] = -tg / piv;
        x[i] = (tg * x[i - 1]) / piv;
    }
    for (int i = n - 2; i >= 0; --i) x[i] -= cp[i] * x[i + 1];
    for (int i = 0; i < n; ++i) u[i] = (double)x[i];
    ADI_HIP_TRY(hipMemcpyAsync(d_weights, u.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, as_stream(stream)));
    ADI_HIP_TRY(hipStreamSynchronize(as_stream(stream)));      // u lives on this stack frame
    return ADI_OK;
}

int adi_axis0_classify(const uint8_t *d_flags, const uint8_t *d_dir_mask, int nx, int ny, int nz, long plane_stride,
                       uint8_t *d_cls, unsigned *d_list, void *stream)
{
    ADI_REQUIRE(d_flags && d_cls && d_list, "adi_axis0_classify: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    hipStream_t st = as_stream(stream);
    ADI_HIP_TRY(hipMemsetAsync(d_list, 0, sizeof(unsigned), st));
    const long nlines = (long)ny * nz;
    hipLaunchKernelGGL(k_classify_lines0, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, st, d_flags, d_dir_mask, L,
                       d_cls, d_list);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_axis0_dots_ichunk(int n_line) { return dots_ichunk(n_line); }

int adi_explicit_rhs_dots(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride,
                          double dx, double dt, double kappa, double theta, double *d_R0, int i_begin, int i_end,
                          int i_org, int n_line, const double *d_weights, double *d_part, void *stream)
{
    ADI_REQUIRE(d_T && d_flags && d_R0 && d_weights && d_part, "adi_explicit_rhs_dots: null argument");
    ADI_REQUIRE(d_T != d_R0, "adi_explicit_rhs_dots: output aliases input");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    ADI_REQUIRE(i_begin >= 0 && i_end <= nx && i_begin < i_end, "adi_explicit_rhs_dots: bad plane range [%d, %d)", i_begin,
                i_end);
    ADI_REQUIRE(n_line >= 2 && i_org >= 0 && i_org + n_line <= nx && i_begin >= i_org && i_end <= i_org + n_line,
                "adi_explicit_rhs_dots: planes [%d, %d) outside the lines [%d, %d)", i_begin, i_end, i_org, i_org + n_line);
    ADI_REQUIRE((nz % 2 == 0) && (L.sx % 2 == 0) && ((((uintptr_t)d_T | (uintptr_t)d_R0 | (uintptr_t)d_part) & 15) == 0) &&
                    (((uintptr_t)d_flags & 1) == 0),
                "adi_explicit_rhs_dots: needs even nz / plane stride and 16-byte aligned fields");
    const int np = i_end - i_begin;
    const int jslab = (ny + 7) / 8, nslab = (ny + jslab - 1) / jslab, ktiles = (nz + 511) / 512;
    // chunks of planes are counted from the start of the lines: a launch on part of the planes (interior first, the
    // planes next to the halos once those have landed) covers whole chunks, except at the end of the lines
    const int ichunk = dots_ichunk(n_line), nchunk = (np + ichunk - 1) / ichunk;
    ADI_REQUIRE((i_begin - i_org) % ichunk == 0 && ((i_end - i_org) % ichunk == 0 || i_end == i_org + n_line),
                "adi_explicit_rhs_dots: plane range [%d, %d) does not cover whole chunks of %d planes", i_begin, i_end, ichunk);
    const long ntiles = (long)nslab * nchunk * ((jslab + 1) / 2) * ktiles;
    hipLaunchKernelGGL((k_explicit_v5<2, true>), dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                       L, 1.0 / (dx * dx), dt * kappa * (1.0 - theta), jslab, ktiles, ichunk, ntiles, i_begin, i_end,
                       d_weights, d_part, i_org, n_line, store_policy_nt(L.nx, L.sx));
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}
int adi_axis0_dots_finish(int variant, const double *d_part, const double *d_weights, const uint8_t *d_cls,
                          const unsigned *d_list, const double *d_R0, const uint8_t *d_flags, const double *d_coeff,
                          const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny,
                          int nz, long plane_stride, double theta, double gam, double dt, double Tinf, long line_begin,
                          long line_end, double *d_cond, void *stream)
{
    bool has_dir, has_q;
    if (int rc = variant_flags(variant, &has_dir, &has_q)) return rc;
    ADI_REQUIRE(d_part && d_weights && d_cls && d_list && d_R0 && d_flags && d_coeff && d_cond,
                "adi_axis0_dots_finish: null argument");
    ADI_REQUIRE(!has_dir || (d_dir_mask && d_dir_val), "adi_axis0_dots_finish: variant needs Dirichlet arrays");
    ADI_REQUIRE(!has_q || d_qflux, "adi_axis0_dots_finish: variant needs the flux array");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    const long nlines = (long)ny * nz;
    ADI_REQUIRE(line_begin >= 0 && line_end <= nlines && line_begin < line_end, "adi_axis0_dots_finish: bad line range");
    SweepScal s;
    s.tg = theta * gam; s.dt = dt; s.Tinf = Tinf; s.sparse = 0; s.box = 0;
    hipStream_t st = as_stream(stream);
    const long nsel = line_end - line_begin;
    const int nchunk = (nx + dots_ichunk(nx) - 1) / dots_ichunk(nx);
    const unsigned grid = (unsigned)((nsel + 255) / 256);
    if (has_q) hipLaunchKernelGGL((k_dots_finish<true>), dim3(grid), dim3(256), 0, st, d_part, nchunk, d_weights, d_cls, d_flags, d_coeff, d_qflux, L, s, line_begin, line_end, d_cond);
    else hipLaunchKernelGGL((k_dots_finish<false>), dim3(grid), dim3(256), 0, st, d_part, nchunk, d_weights, d_cls, d_flags, d_coeff, d_qflux, L, s, line_begin, line_end, d_cond);
    // the lines that are not uniform: the serial two-recurrence condensation from the stored R0
    SweepArgs a;
    a.in = d_R0; a.flags = d_flags; a.coeff = d_coeff; a.dmask = d_dir_mask; a.dval = d_dir_val; a.qf = d_qflux;
    condense_generic_lines(has_dir, has_q, a, L, s, d_cond, d_list, line_begin, nsel, st);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
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
