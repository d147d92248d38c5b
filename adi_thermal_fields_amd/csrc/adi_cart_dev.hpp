// adi_cart_dev.hpp -- device-side pieces shared by the Cartesian translation units of libadi_hip.so (gfx950):
// cache-policy knobs, the per-sweep scalars, segment classification, the row assembly of the tridiagonal systems
// (adi3d_gpu_coeff.py:173-187), the explicit-stage arithmetic (adi3d_numba_coeff.py:240-298) and the surface-segment
// ("mixed") lanes of the FAST kernels.
#pragma once
#include "adi_common.hpp"
#include "adi_core.hpp"

namespace adi {

#ifndef ADI_LOAD_AUX
#define ADI_LOAD_AUX 0      // strided kernels: cache policy of the once-read loads (`in` rows, flags).  nt (2) measured
#endif                      // SLOWER there: axis-1 sweep 0.41 -> 0.45 ms, axis 0 0.45 -> 0.50 ms
#ifndef ADI_LOAD_NT_CONTIG
#define ADI_LOAD_NT_CONTIG 1   // contiguous kernels: streaming (nt) loads of the coalesced rows: 0.42 -> 0.38 ms
#endif
#ifndef ADI_STORE_AUX
#define ADI_STORE_AUX 2     // cache policy of the output stores (2 = nt: streaming; 0 = default)
#endif
#ifndef ADI_BUF_STRIDED
#define ADI_BUF_STRIDED 1   // unfused strided FAST kernels: buffer addressing for whole tiles (0: flat loads)
#endif
constexpr bool kBufStrided = ADI_BUF_STRIDED != 0;
#ifndef ADI_LOAD_PRIO
#define ADI_LOAD_PRIO 3     // s_setprio level of a wave while it issues its loads, back to 0 before its solver phase: waves that are
#endif                      // loading go ahead of waves that are solving, so the memory pipe is fed earlier.  Round 4, alternating A/B
                            // on one box (profiles/r04_r_load_prio_ab.txt): unfused strided FAST sweep (axis 1) 0.433 -> 0.419 ms
                            // (-3 ... -5 %), cylindrical r / phi / z sweeps -5 / -3 / -1 % (step 0.141 -> 0.136 ms); NOT applied where
                            // it measured neutral or worse: the fused kernel (+1 %), the GENERAL strided kernels (+1 ... +2 %), the
                            // contiguous kernels (+-0)
#ifndef ADI_FUSE_D
#define ADI_FUSE_D 10    // rows of j-neighbour loads in flight per thread in the fused FAST kernels (512^3: 2 0.77 ms, 4 0.70, 8 0.68; with the results pinned in the loader 8 0.64-0.66, 10 0.63-0.65, 12 0.67, 16 0.90: spills)
#endif
#ifndef ADI_FUSE_D_MIXED
#define ADI_FUSE_D_MIXED 6   // the build that carries the surface-segment lanes (MIXED = true) has 12 fewer registers to give: with 10 rows in flight it spills (20 B/lane + 102 SGPRs) and runs 0.73 ms on the all-solid 512^3 box and 0.79 ms on the ellipsoid; 8: 0.67 / 0.72, 6: 0.665 / 0.71 (the build without them: 0.65)
#endif
#ifndef ADI_FUSE_OCC
#define ADI_FUSE_OCC 4   // waves per SIMD the fused FAST kernels are compiled for (4: two 512-thread workgroups per CU; 3 measures the same, 2 with deeper prefetch is slower)
#endif

// output fields are written once and not read again by the writing kernel: streaming (nt) stores keep them from
// displacing the lines other workgroups are about to re-read from L2 (measured on the fused kernel: -3.4 %)
__device__ __forceinline__ double2 ld_stream2(const double2 *p)
{
#if ADI_LOAD_NT_CONTIG
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v w = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return make_double2(w.x, w.y);
#else
    return *p;
#endif
}
// nt: block-uniform, decided by the host per launch (SweepScal::nt): streaming stores for fields that do not fit the 256 MB
// Infinity Cache, plain stores for fields that do -- the next kernel then reads what this one wrote from the cache
// (scripts/nt_probe.py: 256^3, 134 MB per field, 0.226 -> 0.204 ms per step with plain stores; 512^3 1.48 against 1.55 with nt)
__device__ __forceinline__ void st_stream2(double2 *p, double2 v, bool nt = (ADI_STORE_AUX == 2))
{
    if (nt) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p));
    } else {
        *p = v;
    }
}

struct SweepScal {
    double tg;    // theta * gamma
    double dt;
    double Tinf;
    int box;      // hint (bit 1 of the ABI's `sparse` argument): every cell of the box is in the mask, so no surface
                  // crosses a segment; the fused FAST kernel then runs its build without the TAIL / HEAD lanes
                  // (12 more registers fit the 128-VGPR budget: 0.69 instead of 0.80 ms at 512^3)
    int sparse;   // 1: coeff / qflux are non-zero only on cells exposed along the sweep axis (packs built by
                  //    adi_build_coeffs), so they are loaded only there; dir_val only where dir_mask is set
    int nofb = 0; // promise (bit 2 of `sparse`): the FAST kernel takes every unit of this sweep -- the caller has seen the
                  //    unit queue of the same (mask, packs, variant, shape) come back empty; the queue reset and the GENERAL
                  //    launch behind the FAST kernel are skipped (3 + 3 launches of ~4.5 us and their gaps per step)
    // Deferred interface correction of a slab decomposition (adi_sweep_corrected, strided axis 1 only): the value this
    // sweep reads at (plane i, row j, column k) is  in + c_w[i] * c_lo[j*nz + k] + c_w[c_n - 1 - i] * c_hi[j*nz + k]
    // -- the axis-0 sweep before it solved every line with zero boundary values, and by linearity the true solution
    // differs from that by the two interface values of the line times the decaying homogeneous solutions c_w.
    // c_lo / c_hi: dense (ny, nz) planes (null: no neighbour on that side); c_w: c_n weights, exactly 0 beyond their reach.
    const double *c_lo = nullptr, *c_hi = nullptr, *c_w = nullptr;
    int c_n = 0;
    unsigned c_bytes = 0;   // bytes of a correction plane (the range of the buffer descriptors over c_lo / c_hi)
    // (Lines that are not uniform: their own weights are applied in memory by adi_deferred_lines_apply before this sweep, and
    // c_lo / c_hi carry zeros on them -- ABI v18; v17 streamed per-cell weight planes through this kernel.)
    // Packs built from per-face SCALARS (h_face_consts of the sweep entry points; only with `sparse`): the Robin coefficient /
    // Neumann flux of a cell exposed along the sweep axis is  (0 + [minus neighbour missing] fc[0]) + [plus neighbour missing]
    // fc[1]  (flux: fc[2], fc[3]) -- the accumulation order of precompute_coeff_packs_unified (adi3d_numba_coeff.py:93-114),
    // so the value is the one stored in the pack array, bit for bit -- and follows from the flags byte alone.  The kernels
    // then do not load coeff / qflux at all: on a curved solid those loads can only be issued once the flags have arrived, a
    // second memory latency in every wave that holds a surface row (512^3 ellipsoid: 6 % of the step).
    int fconst = 0;
    double fc[4] = {0.0, 0.0, 0.0, 0.0};
    int nt = 1;   // streaming output stores (fields beyond the Infinity Cache); 0: plain stores (host: store_policy_nt)
};

// fields of at most this many bytes are written with plain stores (scripts/nt_probe.py: plain wins up to 134 - 168 MB per
// field, streaming from 190 MB)
constexpr size_t kPlainStoreMaxBytes = (size_t)160 << 20;
inline int store_policy_nt(long nx, long sx) { return ((size_t)nx * (size_t)sx * sizeof(double) > kPlainStoreMaxBytes) ? 1 : 0; }

// Coefficient / flux of an in-mask cell that is exposed along the sweep axis: from the flags (per-face scalars, see
// SweepScal::fconst) or from the pack array.  has_lo / has_hi: the minus / plus neighbour along the axis is in the mask.
// FCM: 0 = decided at run time (s.fconst), 1 = always from the flags, 2 = always from the array.  The strided FAST kernels
// are instantiated both ways (a template parameter, adi_sweep_strided_fc.hip): with the choice made at run time the mere
// presence of the load path cost the 512^3 ellipsoid 0.02 ms in each of the two strided sweeps.
template <int FCM = 0>
__device__ __forceinline__ double pack_co(const SweepScal &s, const double *p, bool has_lo, bool has_hi)
{
    if (FCM == 1 || (FCM == 0 && s.fconst)) {
        double co = 0.0;
        if (!has_lo) co += s.fc[0];
        if (!has_hi) co += s.fc[1];
        return co;
    }
    return *p;
}
template <bool HAS_Q, int FCM = 0>
__device__ __forceinline__ double pack_q(const SweepScal &s, const double *p, bool has_lo, bool has_hi)
{
    if (!HAS_Q) return 0.0;
    if (FCM == 1 || (FCM == 0 && s.fconst)) {
        double q = 0.0;
        if (!has_lo) q += s.fc[2];
        if (!has_hi) q += s.fc[3];
        return q;
    }
    return *p;
}

// cell is in the mask and lacks at least one in-mask neighbour along the sweep axis: the only cells where
// precompute_coeff_packs_unified writes a Robin coefficient or a Neumann flux for that axis (:93-99, :104-114)
__device__ __forceinline__ bool axis_exposed(unsigned f, int lbit)
{
    return (f & 1u) && (((f >> lbit) & 3u) != 3u);
}

// Segment classes of the FAST kernels (M rows: block rows 0..M-2 + separator row M-1), from the in-mask bits of its rows:
//   UNI   the uniform-interior segment of section 3.2        OFF  every row outside the mask (identity rows)
//   PAD   beyond the end of the line (no rows)
//   TAIL  block rows [M-1-L, M-1) in the mask down to the in-mask separator, the rows above them outside: the line
//         STARTS inside the segment        HEAD  block rows [0, L) in the mask below an in-mask previous row, the rest
//         of the segment (separator included) outside: the line ENDS inside the segment.   (adi_core.hpp, mixed_*)
//   ISLAND an in-mask run of at most 16 rows that starts and ends inside the block, separator outside (thin walls):
//         decoupled from the rest of the line, solved on the spot (island_solve); L | (first row << 8) is returned
//   GAP   a HEAD run [0, e) and a TAIL run [M-1-L, M-1] with rows outside the mask between them (a slot or channel cut
//         by the line): the two runs are independent, both paths in one lane; e | (L << 8) is returned
enum { SEG_NONE = 0, SEG_UNI = 1, SEG_OFF = 2, SEG_PAD = 3, SEG_TAIL = 4, SEG_HEAD = 5, SEG_ISLAND = 6, SEG_GAP = 7 };

template <int M>
__device__ __forceinline__ int classify_mixed(unsigned inm, unsigned f0, int lbit, int &L)
{
    constexpr int MI = M - 1;
    const unsigned ALL = (M >= 32) ? 0xffffffffu : ((1u << (M & 31)) - 1u);
    L = 0;
    if (inm == 0u) return SEG_OFF;
    if ((inm >> MI) & 1u) {                        // separator in the mask: rows [m, M) in, [0, m) out, 1 <= m <= MI
        const int m = __ffs(inm) - 1;
        if (m >= 1 && inm == (ALL & ~((1u << m) - 1u))) { L = MI - m; return SEG_TAIL; }
        if (m == 0 && ((f0 >> lbit) & 1u)) {       // ... or [0, e) in, a gap, [M-1-L2, M) in; previous row in the mask
            const int e = __ffs(~inm) - 1;         // (inm != ALL here: a full segment is UNI or queued before this)
            const unsigned hi = inm >> e;
            const int z = __ffs(hi) - 1, n2 = __popc(hi);
            if (e >= 1 && e < MI && z >= 1 && (hi >> z) == ((1u << n2) - 1u)) { L = e | ((n2 - 1) << 8); return SEG_GAP; }
        }
    } else {                                       // separator outside: rows [0, e) in, previous row in the mask
        const int e = __popc(inm);
        if (inm == ((1u << e) - 1u) && ((f0 >> lbit) & 1u)) { L = e; return SEG_HEAD; }
        // ... or one short run [m, m + e) with nothing in the mask before it (thin wall)
        const int m = __ffs(inm) - 1;
        if (e <= 16 && (inm >> m) == ((1u << e) - 1u) && (m >= 1 || !((f0 >> lbit) & 1u))) { L = e | (m << 8); return SEG_ISLAND; }
    }
    return SEG_NONE;
}

// One axis of lap1D_x/y/z (adi3d_numba_coeff.py:240-288) in the reference's evaluation order.
__device__ __forceinline__ double lap_axis(bool lo, bool hi, double tlo, double thi, double t, double invdx2)
{
#pragma clang fp contract(off)
    double sacc = 0.0, cnt = 0.0;   // s = 0; if lower in mask: s += T_lo; c += 1; ... (adi3d_numba_coeff.py:246-253)
    if (lo) { sacc += tlo; cnt += 1.0; }
    if (hi) { sacc += thi; cnt += 1.0; }
    return (sacc - cnt * t) * invdx2;
}

// Explicit stage folded into the loads of the axis-0 sweep (FUSE kernels): `in` is the state T, and the value a
// row feeds into its right-hand side is R0 = T + f*(Lx+Ly+Lz) (adi3d_numba_coeff.py:292-298) computed on the fly
// from the six neighbours -- i-neighbours are the thread's own adjacent rows, k-neighbours sit in the adjacent
// lanes, j-neighbours are re-read (L2 / Infinity Cache serves them: the tile that owns them runs next door).
// [vlo, vhi): element offsets relative to `in` that may be read (the whole buffer the view lives in, halo planes
// of a slab included); the FAST kernel loads neighbours without waiting for the flags and needs the bound, the
// GENERAL kernel loads a neighbour only where the flags byte says it exists.
struct Fuse {
    double invdx2, f;
    long sy;
    long vlo, vhi;
    int kt, ny, kg;     // FAST kernel tile order: kt tiles per j-row, groups of kg k-tiles walked j-fastest (kg = 0: off)
    long wlo;           // FAST kernel: the buffer descriptor of the state covers [wlo, wlo + wbytes/8) relative to `in`
    unsigned wbytes;    // (0: the window would not fit 32-bit offsets, GENERAL kernel only)
    double *r0_out;     // pass A only (may be null): R0 is also stored here (box layout of `in`), so that pass B can be
                        // the plain sweep instead of evaluating the explicit stage a second time
};

// FUSE tile order: the j-neighbour rows a tile re-reads belong to the tiles of the adjacent j-rows; walking groups of kg
// k-tiles j-fastest puts those tiles on the same XCD at the same time, so the re-reads are L2 hits.
// kt need not be a multiple of kg: the k-tiles left over form a last, narrower group (a padded nz of 272 has 17 k-tiles; with
// whole groups only the group width fell back to 1 there, i.e. no j-fast order at all: fused kernel 132 against 197 Gcell/s).
__device__ __forceinline__ long tile_jfast(long t, const Fuse &z)
{
    const unsigned kg = (unsigned)z.kg, kt = (unsigned)z.kt;
    const unsigned per = (unsigned)z.ny * kg;
    const unsigned full = kt / kg;                                  // whole groups
    unsigned hi = (unsigned)t / per;
    unsigned w = kg;                                                // width of this tile's group
    if (hi >= full) { hi = full; w = kt - full * kg; }
    const unsigned r = (unsigned)t - hi * per;
    const unsigned j = r / w, lo = r - j * w;
    return (long)j * kt + (long)hi * kg + lo;
}

__device__ __forceinline__ double explicit_cell(unsigned fl, double t, double im, double ip, double jm, double jp,
                                                double km, double kp, const Fuse &z)
{
#pragma clang fp contract(off)
    double L0 = 0.0, L1 = 0.0, L2 = 0.0;
    if (fl & 1u) {
        L0 = lap_axis(fl & 2u, fl & 4u, im, ip, t, z.invdx2);
        L1 = lap_axis(fl & 8u, fl & 16u, jm, jp, t, z.invdx2);
        L2 = lap_axis(fl & 32u, fl & 64u, km, kp, t, z.invdx2);
    }
    return t + z.f * ((L0 + L1) + L2);
}

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

// What a mixed lane does instead of condense_uniform (shared by the FAST kernels).  row0p: pointer to this thread's row
// 0 in coeff / qflux (rows `rstride` elements apart).  On entry d = the incoming values with rows 0 and M-1 already
// assembled (a0, b0 belong to row 0); on exit d[modified row] = its assembled right-hand side, bmod = its diagonal and
// k = the segment's condensation.
template <int M, bool HAS_Q, int FCM = 0, class UC>
__device__ __forceinline__ void mixed_lane_condense(int kind, int L, const UC &U, const SweepScal &s,
                                                    const double *__restrict__ coeff0, const double *__restrict__ qf0,
                                                    long rstride, double a0, double b0, double (&d)[M], double2 &bm,
                                                    Cond &k)
{
    constexpr int MI = M - 1;
    double &bmod = bm.x;
    if (kind == SEG_GAP) {                                  // HEAD run [0, e1) and TAIL run [MI - L2, MI]: independent
        const int e1 = L & 255, L2 = L >> 8;
        double G = 0.0, A = 0.0;
        bm.x = b0;
        if (e1 > 1) {
            const int rm = e1 - 1;
            const double co = pack_co<FCM>(s, coeff0 + (long)rm * rstride, true, false), q = pack_q<HAS_Q, FCM>(s, qf0 + (long)rm * rstride, true, false);
            double din = 0.0, am, cm, dm;
#pragma unroll
            for (int r = 1; r < MI; ++r) din = (r == rm) ? d[r] : din;
            assemble_row<false, HAS_Q>(true, true, false, false, din, co, 0.0, q, s, am, bm.x, cm, dm);
#pragma unroll
            for (int r = 1; r < MI; ++r) d[r] = (r == rm) ? dm : d[r];
        }
        mixed_condense<M, false>(U, d, e1, bm.x, a0, G, A);
        k.gF = G; k.aF = A; k.cF = 0.0;
        bm.y = 1.0;
        if (L2 >= 1) {
            const int rm = MI - L2;
            const double co = pack_co<FCM>(s, coeff0 + (long)rm * rstride, false, true), q = pack_q<HAS_Q, FCM>(s, qf0 + (long)rm * rstride, false, true);
            double din = 0.0, am, cm, dm;
#pragma unroll
            for (int r = 1; r < MI; ++r) din = (r == rm) ? d[r] : din;
            assemble_row<false, HAS_Q>(true, false, true, false, din, co, 0.0, q, s, am, bm.y, cm, dm);
#pragma unroll
            for (int r = 1; r < MI; ++r) d[r] = (r == rm) ? dm : d[r];
            mixed_condense<M, true>(U, d, L2, bm.y, U.s, G, A);
        }
        k.gL = (L2 >= 1) ? G : d[MI - 1]; k.aL = 0.0; k.cL = (L2 >= 1) ? A : 0.0;
        return;
    }
    if (kind == SEG_ISLAND) {
        const int m = L >> 8, len = L & 255, e = m + len;
        double bS = b0, bE = U.bu;                          // a run that starts at row 0: fast_segment_ends assembled it
        if (m >= 1) {
            const double co = pack_co<FCM>(s, coeff0 + (long)m * rstride, false, len > 1), q = pack_q<HAS_Q, FCM>(s, qf0 + (long)m * rstride, false, len > 1);
            double din = 0.0, am, cm, dm;
#pragma unroll
            for (int r = 1; r < MI; ++r) din = (r == m) ? d[r] : din;
            assemble_row<false, HAS_Q>(true, false, len > 1, false, din, co, 0.0, q, s, am, bS, cm, dm);
#pragma unroll
            for (int r = 1; r < MI; ++r) d[r] = (r == m) ? dm : d[r];
        }
        if (len > 1) {
            const double co = pack_co<FCM>(s, coeff0 + (long)(e - 1) * rstride, true, false), q = pack_q<HAS_Q, FCM>(s, qf0 + (long)(e - 1) * rstride, true, false);
            double din = 0.0, am, cm, dm;
#pragma unroll
            for (int r = 1; r < MI; ++r) din = (r == e - 1) ? d[r] : din;
            assemble_row<false, HAS_Q>(true, true, false, false, din, co, 0.0, q, s, am, bE, cm, dm);
#pragma unroll
            for (int r = 1; r < MI; ++r) d[r] = (r == e - 1) ? dm : d[r];
        }
        island_solve<M>(U, d, m, len, bS, bE);
        k.gF = d[0]; k.gL = d[MI - 1];
        k.aF = k.cF = k.aL = k.cL = 0.0;
        bmod = 1.0;
        return;
    }
    const bool tail = kind == SEG_TAIL;
    const int rmod = tail ? MI - L : L - 1;                 // the line-start / line-end row of the run
    bmod = b0;                                              // head run of one row: row 0 is that row, already assembled
    if (L >= 1 && (tail || rmod > 0)) {
        const double co = pack_co<FCM>(s, coeff0 + (long)rmod * rstride, !tail, tail);   // exposed along the axis: carries the Robin coefficient
        const double q = pack_q<HAS_Q, FCM>(s, qf0 + (long)rmod * rstride, !tail, tail);
        double din = 0.0;
#pragma unroll
        for (int r = 1; r < MI; ++r) din = (r == rmod) ? d[r] : din;
        double am, cm, dm;
        assemble_row<false, HAS_Q>(true, !tail, tail, false, din, co, 0.0, q, s, am, bmod, cm, dm);
#pragma unroll
        for (int r = 1; r < MI; ++r) d[r] = (r == rmod) ? dm : d[r];
    }
    double G = 0.0, A = 0.0;
    if (tail) {
        if (L >= 1) mixed_condense<M, true>(U, d, L, bmod, U.s, G, A);
        k.gF = d[0]; k.aF = 0.0; k.cF = 0.0;                // row 0 is outside the mask (m >= 1)
        k.gL = (L >= 1) ? G : d[MI - 1]; k.aL = 0.0; k.cL = (L >= 1) ? A : 0.0;
    } else {
        mixed_condense<M, false>(U, d, L, bmod, a0, G, A);
        k.gF = G; k.aF = A; k.cF = 0.0;
        k.gL = 0.0; k.aL = 0.0; k.cL = 0.0;                 // the separator row is outside the mask: a_S = 0
    }
}

template <int M, class UC>
__device__ __forceinline__ void mixed_lane_back_solve(int kind, int L, const UC &U, double2 bm, double a0,
                                                      double (&d)[M], double xL, double xS)
{
    const double bmod = bm.x;
    if (kind == SEG_GAP) {
        const int e1 = L & 255, L2 = L >> 8;
        mixed_back_solve<M, false>(U, d, e1, bm.x, a0, xL);
        if (L2 >= 1) mixed_back_solve<M, true>(U, d, L2, bm.y, U.s, xS);
        d[M - 1] = xS;
    } else if (kind == SEG_TAIL) {
        if (L >= 1) mixed_back_solve<M, true>(U, d, L, bmod, U.s, xS);
        d[M - 1] = xS;
    } else if (kind == SEG_HEAD) {
        mixed_back_solve<M, false>(U, d, L, bmod, a0, xL);
    }                                                       // (an ISLAND was solved when it was condensed)
}

// LDS hand-off between the lanes of ONE wave (wave-private strips): no workgroup barrier needed
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Lay {
    int nx, ny, nz;
    long sx;   // plane stride in elements (>= ny*nz); row stride is nz
};

// Work queue shared by the FAST and the GENERAL kernel of one sweep: q[0] = number of queued units,
// q[1..] = unit ids.  A FAST kernel that meets a unit it cannot take (a wave / tile touching the surface of
// the solid in a way the uniform model does not cover) appends the unit and leaves it to the GENERAL kernel
// launched right behind it on the same stream, which reads every array of the pack.
__device__ __forceinline__ void enqueue_unit(unsigned *queue, unsigned unit)
{
    if (queue == nullptr) return;          // (no-fallback promise: nobody would drain the queue)
    const unsigned idx = atomicAdd(&queue[0], 1u);
    queue[1 + idx] = unit;
}

struct LineGeom {
    int n;              // rows per line
    long stride;        // elements between consecutive rows
    int n_inner;        // lines that are contiguous in memory (stride 1)
    long n_outer;       // groups of n_inner lines
    long outer_stride;  // elements between groups
    int lbit;           // flags bit of the "previous row in mask" test (next row: lbit + 1)
};

// separator system of a tile through LDS: line-major regrouping, in-wave PCR, separator values back
__device__ __forceinline__ void tile_separators(double *sm, int tid, int kk, int sg, int Lp, int LINES, double aS,
                                                double bS, double cS, double dS, const Cond &k, double &xL,
                                                double &xS)
{
    // LDS: 7 arrays [LINES][Lp + 1] (one padding column: conflict-free for both access directions); the
    // separator values are written over the first array (every thread overwrites the entry it has just read)
    const int ld = Lp + 1;
    const int plane = LINES * ld;
    double *sX1 = sm, *sX2 = sm + plane, *sCS = sm + 2 * plane, *sX4 = sm + 3 * plane;
    double *sGF = sm + 4 * plane, *sAF = sm + 5 * plane, *sCF = sm + 6 * plane, *sXS = sm;
    {
        const int w = kk * ld + sg;
        sX1[w] = -aS * k.aL;                          // ra
        sX2[w] = __builtin_fma(-aS, k.cL, bS);        // rb without the next-segment term
        sCS[w] = cS;
        sX4[w] = __builtin_fma(-aS, k.gL, dS);        // rd without the next-segment term
        sGF[w] = k.gF;
        sAF[w] = k.aF;
        sCF[w] = k.cF;
    }
    __syncthreads();
    {
        const int pl = tid >> (__ffs(Lp) - 1), ps = tid & (Lp - 1);   // Lp is a power of two  // line-major regrouping: Lp consecutive lanes = one line
        const int w = pl * ld + ps;
        const double c2 = sCS[w];
        const bool hasn = ps < Lp - 1;
        const double gFn = hasn ? sGF[w + 1] : 0.0, aFn = hasn ? sAF[w + 1] : 0.0, cFn = hasn ? sCF[w + 1] : 0.0;
        const double ra = sX1[w];
        const double rb = __builtin_fma(-c2, aFn, sX2[w]);
        const double rc = -c2 * cFn;
        const double rd = __builtin_fma(-c2, gFn, sX4[w]);
        sXS[w] = pcr_solve(ra, rb, rc, rd, ps, Lp);
    }
    __syncthreads();
    xS = sXS[kk * ld + sg];
    xL = (sg > 0) ? sXS[kk * ld + sg - 1] : 0.0;
}

// Buffer addressing (raw_buffer_load/store: 128-bit descriptor + per-thread 32-bit byte offset + scalar byte offset):
// a strided tile touches M rows x several arrays, and with flat global loads every one of them costs 64-bit address
// arithmetic in the VALU (measured: half of the fused kernel's VALU instructions); here the row offsets live in
// scalar registers and one per-thread offset serves every load.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ double buf_load_f64_once(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, ADI_LOAD_AUX);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ u32x2 as_u32x2(double x)
{
    u32x2 v;
    v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
    return v;
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x,
                                              bool nt = (ADI_STORE_AUX == 2))
{
    u32x2 v;
    v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
    if (nt) __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// Deferred interface correction (SweepScal::c_*): weights of the plane `to` this tile lies in (block-uniform, scalar
// loads) and the correction of M rows of one line.  off8: byte offset of this thread's row 0 inside a correction plane,
// st8: bytes between rows.  The planes are read through range-checked buffer descriptors, so the rows of a padding
// segment (beyond the end of the line) read 0 -- with the WHOLE offset in the per-thread register: the hardware checks
// the register offset against the descriptor's range, a scalar row offset is added unchecked (with per-plane corrections
// the rows of a padding segment of the last plane lay beyond the allocation: a memory fault, found by the first GPU run of
// the per-line form).  The loads are L2 hits: two planes re-read by every plane of the slab.
__device__ __forceinline__ double2 corr_weights(const SweepScal &s, long to)
{
    double2 w = make_double2(0.0, 0.0);
    if (s.c_w != nullptr) {
        if (s.c_lo != nullptr) w.x = s.c_w[to];
        if (s.c_hi != nullptr) w.y = s.c_w[s.c_n - 1 - to];
    }
    return w;
}
// (Round 4 tried fetching both planes with ONE 16-byte load per row from an interleaved copy -- the strided kernels are bound by
// the number of vector-memory instructions -- on the planes within reach of both interfaces: slower, not faster.  32 rows of
// 16-byte results in flight took the axis-1 FAST kernel from 177 to 212 VGPRs and the corrected sweep from 0.51 to 0.58 ms.)
template <int M>
__device__ __forceinline__ void corr_apply(const SweepScal &s, double2 w, long to, unsigned off8, unsigned st8, double (&d)[M],
                                           bool inside)
{
    (void)to;
    if (w.x != 0.0) {
        const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc((void *)s.c_lo, 0, (int)s.c_bytes, 0x00020000);
        if (inside) {                  // every row of the tile lies inside the plane: scalar row offsets (0.02 ms at 512^3)
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = __builtin_fma(w.x, buf_load_f64(rL, off8, (unsigned)r * st8), d[r]);
        } else {
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = __builtin_fma(w.x, buf_load_f64(rL, off8 + (unsigned)r * st8, 0u), d[r]);
        }
    }
    if (w.y != 0.0) {
        const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void *)s.c_hi, 0, (int)s.c_bytes, 0x00020000);
        if (inside) {
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = __builtin_fma(w.y, buf_load_f64(rH, off8, (unsigned)r * st8), d[r]);
        } else {
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = __builtin_fma(w.y, buf_load_f64(rH, off8 + (unsigned)r * st8, 0u), d[r]);
        }
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double old, double src)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// value of lane r of every 16-lane row, in all lanes of that row (DPP row_newbcast:r)
__device__ __forceinline__ double row_bcast(double v, int r)
{
    switch (r & 15) {
#define ADI_BC(n) case n: return dpp_mov<0x150 + n>(0.0, v);
        ADI_BC(0) ADI_BC(1) ADI_BC(2) ADI_BC(3) ADI_BC(4) ADI_BC(5) ADI_BC(6) ADI_BC(7)
        ADI_BC(8) ADI_BC(9) ADI_BC(10) ADI_BC(11) ADI_BC(12) ADI_BC(13) ADI_BC(14)
#undef ADI_BC
        default: return dpp_mov<0x15f>(0.0, v);
    }
}

// part 2: the two general rows of a uniform segment (row 0 and the separator)
template <int M, bool HAS_DIR, bool HAS_Q, int FCM = 0>
__device__ __forceinline__ void fast_segment_ends(const double *__restrict__ coeff, const double *__restrict__ dval,
                                                  const double *__restrict__ qf, const LineGeom &g, long base, int r0,
                                                  unsigned f0, unsigned fS, bool dirS, const SweepScal &s,
                                                  double (&d)[M], double &a0, double &b0, double &aS, double &bS,
                                                  double &cS)
{
    const long p0 = base + (long)r0 * g.stride, pS = base + (long)(r0 + M - 1) * g.stride;
    const bool e0 = axis_exposed(f0, g.lbit), eS = axis_exposed(fS, g.lbit);
    const bool l0 = (f0 >> g.lbit) & 1u, h0 = (f0 >> (g.lbit + 1)) & 1u, lS = (fS >> g.lbit) & 1u, hS = (fS >> (g.lbit + 1)) & 1u;
    const double co0 = e0 ? pack_co<FCM>(s, coeff + p0, l0, h0) : 0.0, coS = eS ? pack_co<FCM>(s, coeff + pS, lS, hS) : 0.0;
    double q0 = 0.0, qS = 0.0, dvS = 0.0;
    if (HAS_Q) { q0 = e0 ? pack_q<HAS_Q, FCM>(s, qf + p0, l0, h0) : 0.0; qS = eS ? pack_q<HAS_Q, FCM>(s, qf + pS, lS, hS) : 0.0; }
    if (HAS_DIR) dvS = dirS ? dval[pS] : 0.0;
    double c0;
    assemble_row<HAS_DIR, HAS_Q>(f0 & 1u, (f0 >> g.lbit) & 1u, (f0 >> (g.lbit + 1)) & 1u, false, d[0], co0, 0.0, q0, s,
                                 a0, b0, c0, d[0]);
    assemble_row<HAS_DIR, HAS_Q>(fS & 1u, (fS >> g.lbit) & 1u, (fS >> (g.lbit + 1)) & 1u, dirS, d[M - 1], coS, dvS, qS,
                                 s, aS, bS, cS, d[M - 1]);
}

// ---- coalesced global access for the contiguous FAST kernel ------------------------------------------------
// A lane that loads its own M consecutive rows touches 64 different 128-byte lines per wave instruction; measured on
// this part, pure streaming with 128-byte lane chunks tops out at 4.6-4.8 TB/s against 6.1 TB/s for fully
// coalesced 16-byte-per-lane accesses.  So the wave reads its 64*M contiguous doubles coalesced (lane l: elements
// 2l, 2l+1 of each 128-element piece), transposes through a wave-private LDS strip (chunk of M doubles + 16 bytes of
// padding: conflict-free for the 128-bit reads), in two halves of 32 lanes to keep the strip at 4.5 KiB, and writes
// the result back the same way.  No block barrier: the strip is private to the wave.
template <int M>
__device__ __forceinline__ void coal_load(const double *__restrict__ gsrc /* wave base */, double *strip, int lane,
                                          double (&d)[M])
{
    constexpr int CH = M + 2;              // chunk pitch in doubles (M*8 + 16 bytes)
    constexpr int NJ = (32 * M) / 128 > 0 ? (32 * M) / 128 : 1;     // double2 loads per lane and half (M >= 4)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double2 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            v[j] = ld_stream2(reinterpret_cast<const double2 *>(gsrc + h * 32 * M + 128 * j + 2 * lane));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = 128 * j + 2 * lane;           // element inside the half
            *reinterpret_cast<double2 *>(strip + (e / M) * CH + (e % M)) = v[j];
        }
        wave_lds_fence();
        if ((lane >> 5) == h) {
            const double *c = strip + (lane & 31) * CH;
#pragma unroll
            for (int i = 0; i < M / 2; ++i) {
                const double2 t = *reinterpret_cast<const double2 *>(c + 2 * i);
                d[2 * i] = t.x;
                d[2 * i + 1] = t.y;
            }
        }
        wave_lds_fence();
    }
}

template <int M>
__device__ __forceinline__ void coal_store(double *__restrict__ gdst, double *strip, int lane, const double (&d)[M],
                                           bool nt = (ADI_STORE_AUX == 2))
{
    constexpr int CH = M + 2;
    constexpr int NJ = (32 * M) / 128 > 0 ? (32 * M) / 128 : 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            double *c = strip + (lane & 31) * CH;
#pragma unroll
            for (int i = 0; i < M / 2; ++i) *reinterpret_cast<double2 *>(c + 2 * i) = make_double2(d[2 * i], d[2 * i + 1]);
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = 128 * j + 2 * lane;
            const double2 t = *reinterpret_cast<const double2 *>(strip + (e / M) * CH + (e % M));
            st_stream2(reinterpret_cast<double2 *>(gdst + h * 32 * M + e), t, nt);
        }
        wave_lds_fence();
    }
}

// The same for lines whose segment count S = n / M is not a power of two (Lp = next_pow2(S) lanes per line, the lanes beyond S
// are padding): the lines of a wave are still consecutive in memory, n doubles each, so every half of the wave reads its
// lines' contiguous range coalesced and scatters the pairs into the chunk slots of the lanes that own them.  (Until late round
// 3 such lines took lane-owned 128-byte chunks: 496-row lines 259 Gcell/s against 365 at 512 rows.)
template <int M>
__device__ __forceinline__ void coal_half_range(int h, int n, int Lp, int &start, int &len, int &lph)
{
    lph = Lp <= 32 ? 32 / Lp : 0;                 // whole lines per half (0: one line spans both halves)
    start = lph ? h * lph * n : h * 32 * M;       // first element of the half, from the wave base
    len = lph ? lph * n : n - h * 32 * M;         // (Lp == 64 means n > 32 M: both halves own rows)
    if (len > 32 * M) len = 32 * M;
}

template <int M>
__device__ __forceinline__ int coal_slot(int e, int n, int Lp, int lph)
{
    constexpr int CH = M + 2;
    const int ll = lph ? (int)(e >= n) + (int)(e >= 2 * n) + (int)(e >= 3 * n) : 0;     // line of the half (at most 4)
    const int w = e - ll * n;
    return (ll * Lp + w / M) * CH + w % M;
}

template <int M>
__device__ __forceinline__ void coal_load_r(const double *__restrict__ gsrc /* wave base */, double *strip, int lane,
                                            double (&d)[M], int n, int Lp)
{
    constexpr int CH = M + 2;
    constexpr int NJ = (32 * M) / 128 > 0 ? (32 * M) / 128 : 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int start, len, lph;
        coal_half_range<M>(h, n, Lp, start, len, lph);
        double2 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = 128 * j + 2 * lane;
            v[j] = make_double2(0.0, 0.0);
            if (e < len) v[j] = ld_stream2(reinterpret_cast<const double2 *>(gsrc + start + e));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = 128 * j + 2 * lane;
            if (e < len) *reinterpret_cast<double2 *>(strip + coal_slot<M>(e, n, Lp, lph)) = v[j];
        }
        wave_lds_fence();
        if ((lane >> 5) == h) {
            const double *c = strip + (lane & 31) * CH;
#pragma unroll
            for (int i = 0; i < M / 2; ++i) {
                const double2 t = *reinterpret_cast<const double2 *>(c + 2 * i);
                d[2 * i] = t.x;
                d[2 * i + 1] = t.y;
            }
        }
        wave_lds_fence();
    }
}

template <int M>
__device__ __forceinline__ void coal_store_r(double *__restrict__ gdst, double *strip, int lane, const double (&d)[M], int n,
                                             int Lp, bool owns, bool nt)
{
    constexpr int CH = M + 2;
    constexpr int NJ = (32 * M) / 128 > 0 ? (32 * M) / 128 : 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int start, len, lph;
        coal_half_range<M>(h, n, Lp, start, len, lph);
        if ((lane >> 5) == h && owns) {            // (padding lanes own no rows: their chunk slots are never read below)
            double *c = strip + (lane & 31) * CH;
#pragma unroll
            for (int i = 0; i < M / 2; ++i) *reinterpret_cast<double2 *>(c + 2 * i) = make_double2(d[2 * i], d[2 * i + 1]);
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = 128 * j + 2 * lane;
            if (e < len) {
                const double2 t = *reinterpret_cast<const double2 *>(strip + coal_slot<M>(e, n, Lp, lph));
                st_stream2(reinterpret_cast<double2 *>(gdst + start + e), t, nt);
            }
        }
        wave_lds_fence();
    }
}

}  // namespace adi
