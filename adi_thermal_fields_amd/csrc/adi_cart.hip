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
//   K5 k_condense_*     slab condensation + k_interface: the reduced interface system of a sweep whose
//                       lines span several GPUs (slab decomposition along memory axis 0)
//
// Device layout: planes (fixed first index) may be padded: `sx` = plane stride in elements >= ny*nz.
// A power-of-two plane stride (512^2 doubles = 2 MiB) puts all rows of an axis-0 tile on the same HBM
// channels (measured: 2.7 -> 4.1 TB/s on the axis-0 sweep when the stride is padded by one row).
//
// Data layout: C-order (n0, n1, n2) fp64 fields and 1-byte masks, exactly the reference's
// (adi3d_numba_coeff.py:18, :31-36); axis 2 is contiguous.  All kernels are HBM-bandwidth bound
// (< 1 flop/byte); no MFMA.
#include <stdlib.h>

#include <vector>

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
#ifndef ADI_FUSE_D
#define ADI_FUSE_D 8     // rows of j-neighbour loads in flight per thread in the fused FAST kernels (2: 0.77 ms, 4: 0.70, 8: 0.68, 16: 1.02 at 512^3)
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
__device__ __forceinline__ void st_stream2(double2 *p, double2 v)
{
#if ADI_STORE_AUX == 2
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p));
#else
    *p = v;
#endif
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
};

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
__device__ __forceinline__ long tile_jfast(long t, const Fuse &z)
{
    const unsigned per = (unsigned)z.ny * (unsigned)z.kg;
    const unsigned hi = (unsigned)t / per, r = (unsigned)t - hi * per;
    const unsigned j = r / (unsigned)z.kg, lo = r - j * (unsigned)z.kg;
    return (long)j * z.kt + (long)hi * z.kg + lo;
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
template <int M, bool HAS_Q>
__device__ __forceinline__ void mixed_lane_condense(int kind, int L, const UniC<M> &U, const SweepScal &s,
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
            const double co = coeff0[(long)rm * rstride], q = HAS_Q ? qf0[(long)rm * rstride] : 0.0;
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
            const double co = coeff0[(long)rm * rstride], q = HAS_Q ? qf0[(long)rm * rstride] : 0.0;
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
            const double co = coeff0[(long)m * rstride], q = HAS_Q ? qf0[(long)m * rstride] : 0.0;
            double din = 0.0, am, cm, dm;
#pragma unroll
            for (int r = 1; r < MI; ++r) din = (r == m) ? d[r] : din;
            assemble_row<false, HAS_Q>(true, false, len > 1, false, din, co, 0.0, q, s, am, bS, cm, dm);
#pragma unroll
            for (int r = 1; r < MI; ++r) d[r] = (r == m) ? dm : d[r];
        }
        if (len > 1) {
            const double co = coeff0[(long)(e - 1) * rstride], q = HAS_Q ? qf0[(long)(e - 1) * rstride] : 0.0;
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
        const double co = coeff0[(long)rmod * rstride];     // exposed along the axis: carries the Robin coefficient
        const double q = HAS_Q ? qf0[(long)rmod * rstride] : 0.0;
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

template <int M>
__device__ __forceinline__ void mixed_lane_back_solve(int kind, int L, const UniC<M> &U, double2 bm, double a0,
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
                const double2 t = q[i];      // lane-owned chunks: several instructions share a 128-byte line -> default policy (nt: 1.01 -> 1.60 ms)
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
    const unsigned idx = atomicAdd(&queue[0], 1u);
    queue[1 + idx] = unit;
}

// ---- coalesced global access for the contiguous FAST kernel ------------------------------------------------
// A lane that loads its own M consecutive rows touches 64 different 128-byte lines per wave instruction; measured on
// this part, pure streaming with 128-byte lane chunks tops out at 4.6-4.8 TB/s against 6.1 TB/s for fully
// coalesced 16-byte-per-lane accesses.  So the wave reads its 64*M contiguous doubles coalesced (lane l: elements
// 2l, 2l+1 of each 128-element piece), transposes through a wave-private LDS strip (chunk of M doubles + 16 bytes of
// padding: conflict-free for the 128-bit reads), in two halves of 32 lanes to keep the strip at 4.5 KiB, and writes
// the result back the same way.  No block barrier: the strip is private to the wave.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
__device__ __forceinline__ void coal_store(double *__restrict__ gdst, double *strip, int lane, const double (&d)[M])
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
            st_stream2(reinterpret_cast<double2 *>(gdst + h * 32 * M + e), t);
        }
        wave_lds_fence();
    }
}

// GENERAL body for one wave-unit (unit = index of a group of 64/Lp consecutive lines)
template <int M, bool VEC, bool HAS_DIR, bool HAS_Q, bool COAL>
__device__ __forceinline__ void contig_unit_general(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, const Lay &L, int Lp, const SweepScal &s, long unit, double *strip)
{
    const int n = L.nz;
    const long nlines = (long)L.nx * L.ny;
    const int lane = threadIdx.x & 63;
    const int lw = 64 >> (__ffs(Lp) - 1);  // lines per wave
    const int li = lane & (Lp - 1);
    const unsigned line = (unsigned)unit * (unsigned)lw + ((unsigned)lane >> (__ffs(Lp) - 1));   // < 2^31 lines
    const bool active = line < (unsigned long)nlines;
    const int r0 = li * M;
    const unsigned pi = line / (unsigned)L.ny;
    const long base = (long)pi * L.sx + (long)(line - pi * (unsigned)L.ny) * n + r0;

    double vin[M], vco[M], vdv[M], vq[M];
    unsigned fb[M], db[M];
    load_bytes_contig<M, VEC>(flags, base, r0, n, active, fb);
    if (HAS_DIR) load_bytes_contig<M, VEC>(dmask, base, r0, n, active, db);
    const long wbase = __shfl(base, 0);    // COAL: the unit's lines are contiguous from lane 0's base
    if constexpr (COAL) coal_load<M>(in + wbase, strip, lane, vin);
    else load_rows_contig<M, VEC>(in, base, r0, n, active, vin);
    if (COAL && !s.sparse) {               // dense packs: every lane needs every array -> cooperative loads
        if constexpr (COAL) {
            coal_load<M>(coeff + wbase, strip, lane, vco);
            if (HAS_Q) coal_load<M>(qf + wbase, strip, lane, vq);
            if (HAS_DIR) coal_load<M>(dval + wbase, strip, lane, vdv);
        }
    } else {
        bool need = !s.sparse, needd = !s.sparse;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            need = need || axis_exposed(fb[r], 5);
            if (HAS_DIR) needd = needd || (db[r] != 0);
        }
        load_rows_contig<M, VEC>(coeff, base, r0, n, active && need, vco);
        if (HAS_Q) load_rows_contig<M, VEC>(qf, base, r0, n, active && need, vq);
        if (HAS_DIR) load_rows_contig<M, VEC>(dval, base, r0, n, active && needd, vdv);
    }
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
    if constexpr (COAL) {
        coal_store<M>(out + wbase, strip, lane, x);
    } else if (VEC) {
        if (active && r0 < n) {
            double2 *q = reinterpret_cast<double2 *>(out + base);
#pragma unroll
            for (int i = 0; i < M / 2; ++i) q[i] = make_double2(x[2 * i], x[2 * i + 1]);   // lane-owned chunks: default policy
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (active && r0 + r < n) out[base + r] = x[r];
    }
}

// GENERAL kernel: every unit (queue == nullptr) or the units a FAST kernel queued.
// MODE: 0 scalar loads, 1 lane-chunk vector loads, 2 coalesced loads transposed through a wave-private LDS strip.
template <int M, int MODE, bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_sweep_contig(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, Lay L, int Lp, SweepScal s, long nunits, const unsigned *__restrict__ queue,
    int ratio)
{
    // ratio: units of this kernel per queued unit (the FAST kernel may group more lines per wave)
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    __shared__ __align__(16) double strips[MODE == 2 ? 4 * 32 * (M + 2) : 2];
    double *strip = strips + (MODE == 2 ? wave * 32 * (M + 2) : 0);
    if (queue == nullptr) {
        const long unit = (long)blockIdx.x * wpb + wave;
        if (unit < nunits)
            contig_unit_general<M, MODE != 0, HAS_DIR, HAS_Q, MODE == 2>(in, flags, coeff, dmask, dval, qf, out, L, Lp, s,
                                                                         unit, strip);
    } else {
        const long cnt = (long)queue[0] * ratio;
        for (long i = (long)blockIdx.x * wpb + wave; i < cnt; i += (long)gridDim.x * wpb) {
            const long unit = (long)queue[1 + i / ratio] * ratio + ((unsigned)i % (unsigned)ratio);
            if (unit < nunits)
                contig_unit_general<M, MODE != 0, HAS_DIR, HAS_Q, MODE == 2>(in, flags, coeff, dmask, dval, qf, out, L, Lp,
                                                                             s, unit, strip);
        }
    }
}

// FAST kernel (sparse packs only): waves whose lanes all hold uniform-interior segments (rows 1..M-2 have both
// z-neighbours in the mask and are not Dirichlet; row 0 may start a line / carry a Robin coefficient; the
// separator row is general).  Such a wave needs no reciprocal chains (condense_uniform) and ~50 VGPRs, so
// 8 waves per SIMD keep HBM busy.  Other waves are queued for the GENERAL kernel.
template <int M, int MODE, bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_sweep_contig_fast(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, Lay L, int Lp, SweepScal s, long nunits, unsigned *__restrict__ queue, UniC<M> U)
{
    const int n = L.nz;
    const long nlines = (long)L.nx * L.ny;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wave));
    if (unit >= nunits) return;
    const int lw = 64 >> (__ffs(Lp) - 1);
    const int li = lane & (Lp - 1);
    const unsigned line = (unsigned)unit * (unsigned)lw + ((unsigned)lane >> (__ffs(Lp) - 1));   // < 2^31 lines
    const bool active = line < (unsigned long)nlines;
    const int r0 = li * M;
    const unsigned pi = line / (unsigned)L.ny;
    const long base = (long)pi * L.sx + (long)(line - pi * (unsigned)L.ny) * n + r0;

    constexpr bool VEC = MODE != 0;
    __shared__ __align__(16) double strips[MODE == 2 ? 4 * 32 * (M + 2) : 2];
    double *strip = strips + (MODE == 2 ? wave * 32 * (M + 2) : 0);
    // MODE 2: the 64/Lp lines of a unit are consecutive in memory (host checks ny % lw == 0), so the wave's 64*M
    // doubles start at the base of lane 0
    const long wbase = __shfl(base, 0);
    double d[M];
    unsigned fb[M], db[M];
    load_bytes_contig<M, VEC>(flags, base, r0, n, active, fb);
    if (HAS_DIR) load_bytes_contig<M, VEC>(dmask, base, r0, n, active, db);
    if constexpr (MODE == 2) coal_load<M>(in + wbase, strip, lane, d);
    else load_rows_contig<M, VEC>(in, base, r0, n, active, d);
    // the two ends of a line are always exposed: fetch their coefficient / flux with the first batch of loads
    const bool sp0 = active && li == 0, spS = active && (r0 + M == n);
    const double co0s = sp0 ? coeff[base] : 0.0, coSs = spS ? coeff[base + M - 1] : 0.0;
    double q0s = 0.0, qSs = 0.0;
    if (HAS_Q) { q0s = sp0 ? qf[base] : 0.0; qSs = spS ? qf[base + M - 1] : 0.0; }
    // padding lanes (beyond the end of a line whose segment count is not a power of two, or beyond the last line)
    // own no rows: they never force the unit to the GENERAL kernel, export an identity block and store nothing
    const bool pad = !active || r0 >= n;
    int kind = SEG_NONE, Lm = 0;                    // segment class of this lane (classify_mixed) and length of a mixed run
    {
        const unsigned FULL = 1u | (3u << 5), ROW0 = 1u | (1u << 6);
        bool uni = ((fb[0] & ROW0) == ROW0) && !(HAS_DIR && db[0] != 0), nodir = !(HAS_DIR && db[0] != 0);
        unsigned inm = fb[0] & 1u;
#pragma unroll
        for (int r = 1; r < M - 1; ++r) {
            uni = uni && ((fb[r] & FULL) == FULL) && !(HAS_DIR && db[r] != 0);
            nodir = nodir && !(HAS_DIR && db[r] != 0);
        }
#pragma unroll
        for (int r = 1; r < M; ++r) inm |= (fb[r] & 1u) << r;
        if (pad) kind = SEG_PAD;
        else if (uni) kind = SEG_UNI;
        else {
            kind = classify_mixed<M>(inm, fb[0], 5, Lm);
            if (kind >= SEG_TAIL && !nodir) kind = SEG_NONE;
        }
    }
    const bool off = kind == SEG_OFF;
    if (!__all(kind != SEG_NONE)) {
        if (lane == 0) enqueue_unit(queue, (unsigned)unit);
        return;
    }
    // row 0 and the separator row are general rows; only they can carry a coefficient / flux / Dirichlet value
    const bool e0 = axis_exposed(fb[0], 5), eS = axis_exposed(fb[M - 1], 5);
    const double co0 = e0 ? (sp0 ? co0s : coeff[base]) : 0.0, coS = eS ? (spS ? coSs : coeff[base + M - 1]) : 0.0;
    double q0 = 0.0, qS = 0.0, dvS = 0.0;
    if (HAS_Q) { q0 = e0 ? (sp0 ? q0s : qf[base]) : 0.0; qS = eS ? (spS ? qSs : qf[base + M - 1]) : 0.0; }
    const bool dirS = HAS_DIR && db[M - 1] != 0;
    if (HAS_DIR) dvS = dirS ? dval[base + M - 1] : 0.0;
    double a0, b0, c0, aS, bS, cS;
    assemble_row<HAS_DIR, HAS_Q>(fb[0] & 1u, (fb[0] >> 5) & 1u, (fb[0] >> 6) & 1u, false, d[0], co0, 0.0, q0, s, a0,
                                 b0, c0, d[0]);
    assemble_row<HAS_DIR, HAS_Q>(fb[M - 1] & 1u, (fb[M - 1] >> 5) & 1u, (fb[M - 1] >> 6) & 1u, dirS, d[M - 1], coS,
                                 dvS, qS, s, aS, bS, cS, d[M - 1]);
    Cond k;
    double kappa;
    condense_uniform<M>(U, a0, b0, d, k, kappa);
    if (pad || off) {                               // identity block; an off segment keeps d = in, which it stores back
        k.gF = k.aF = k.cF = k.gL = k.aL = k.cL = 0.0;
        kappa = 0.0; aS = 0.0; bS = 1.0; cS = 0.0;
        if (pad) d[M - 1] = 0.0;
    }
    double2 bmod = make_double2(1.0, 1.0);
    if (kind >= SEG_TAIL) mixed_lane_condense<M, HAS_Q>(kind, Lm, U, s, coeff + base, qf + base, 1L, a0, b0, d, bmod, k);
    const double gFn = __shfl_down(k.gF, 1, Lp), aFn = __shfl_down(k.aF, 1, Lp), cFn = __shfl_down(k.cF, 1, Lp);
    double ra, rb, rc, rd;
    reduced_row(aS, bS, cS, d[M - 1], k, gFn, aFn, cFn, ra, rb, rc, rd);
    const double xS = pcr_solve(ra, rb, rc, rd, li, Lp);
    double xL = __shfl_up(xS, 1, Lp);
    if (li == 0) xL = 0.0;
    if (kind == SEG_UNI || kind == SEG_PAD) back_solve_uniform<M>(U, a0, kappa, d, xL, xS);
    else if (kind >= SEG_TAIL) mixed_lane_back_solve<M>(kind, Lm, U, bmod, a0, d, xL, xS);
    if constexpr (MODE == 2) {
        coal_store<M>(out + wbase, strip, lane, d);      // (the host takes this mode only when no lane is padding)
    } else if (pad) {
        // nothing to store
    } else if (VEC) {
        double2 *q = reinterpret_cast<double2 *>(out + base);
#pragma unroll
        for (int i = 0; i < M / 2; ++i) q[i] = make_double2(d[2 * i], d[2 * i + 1]);   // lane-owned chunks: default policy
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) out[base + r] = d[r];
    }
}

// ------------------------------------------------------------------------------------------------
// K2: strided-axis sweep.  A workgroup owns a tile of LINES adjacent lines and all Lp segments of each;
// thread (sg, kk) keeps the M rows of segment sg of line kk in registers (lanes run along the contiguous
// direction, so every access is coalesced without a transpose).  Only the 7 condensation numbers per
// segment travel through LDS to regroup the separator system line-major for the in-wave PCR, and the
// separator values travel back.  LINES = 8 (64-byte row pieces, 512-thread workgroups, two per CU so one
// loads while the other solves); the XCD-chunked tile order puts the tile holding the other half of each
// 128-byte line on the same XCD right behind it, so the half-line is served by that XCD's L2.
//
// Lines geometry: element (row r, line (to, kcol)) lives at to*outer_stride + r*stride + kcol.
// xlo/xhi (optional, dense per line): values of the unknown just before row 0 / after row n-1 when the line
// continues on a neighbouring GPU; the coupling itself comes from the halo bits of `flags`.
// ------------------------------------------------------------------------------------------------
struct LineGeom {
    int n;              // rows per line
    long stride;        // elements between consecutive rows
    int n_inner;        // lines that are contiguous in memory (stride 1)
    long n_outer;       // groups of n_inner lines
    long outer_stride;  // elements between groups
    int lbit;           // flags bit of the "previous row in mask" test (next row: lbit + 1)
};

template <int M>
struct SegRaw {
    double vin[M], vco[M], vdv[M], vq[M];
    unsigned fb[M];
    bool dirb[M];
};

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__device__ __forceinline__ void load_segment_raw(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    const LineGeom &g, long base, int r0, bool active, const SweepScal &s, SegRaw<M> &R, const Fuse &fz = Fuse())
{
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const bool ok = active && (r0 + r) < g.n;
        const long p = base + (long)(r0 + r) * g.stride;
        R.fb[r] = ok ? flags[p] : 0u;
        R.vin[r] = ok ? in[p] : 0.0;
    }
    if (FUSE) {
        // vin <- R0 of the explicit stage; every neighbour is loaded only where the flags byte says it exists
        // (rows beyond the line / inactive lanes have flags 0 and stay 0)
        double prev = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const long p = base + (long)(r0 + r) * g.stride;
            const unsigned f = R.fb[r];
            const double cur = R.vin[r];
            double im, ip;
            if (r > 0) im = prev; else im = (f & 2u) ? in[p - g.stride] : 0.0;
            if (r < M - 1 && r0 + r + 1 < g.n) ip = R.vin[r + 1];   // still the state: rows are overwritten in order
            else ip = (f & 4u) ? in[p + g.stride] : 0.0;             // next segment / halo plane of a slab
            const double jm = (f & 8u) ? in[p - fz.sy] : 0.0, jp = (f & 16u) ? in[p + fz.sy] : 0.0;
            const double km = (f & 32u) ? in[p - 1] : 0.0, kp = (f & 64u) ? in[p + 1] : 0.0;
            R.vin[r] = explicit_cell(f, cur, im, ip, jm, jp, km, kp, fz);
            prev = cur;
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const bool ok = active && (r0 + r) < g.n;
        const long p = base + (long)(r0 + r) * g.stride;
        const bool need = ok && (!s.sparse || axis_exposed(R.fb[r], g.lbit));
        R.dirb[r] = false;
        if (HAS_DIR) R.dirb[r] = ok && dmask[p] != 0;
        R.vco[r] = need ? coeff[p] : 0.0;
        R.vq[r] = (HAS_Q && need) ? qf[p] : 0.0;
        R.vdv[r] = (HAS_DIR && ok && (!s.sparse || R.dirb[r])) ? dval[p] : 0.0;
    }
}

template <int M, bool HAS_DIR, bool HAS_Q>
__device__ __forceinline__ void assemble_one(const SegRaw<M> &R, int r, int lbit, const SweepScal &s, double &a,
                                             double &b, double &c, double &d)
{
    assemble_row<HAS_DIR, HAS_Q>(R.fb[r] & 1u, (R.fb[r] >> lbit) & 1u, (R.fb[r] >> (lbit + 1)) & 1u, R.dirb[r],
                                 R.vin[r], R.vco[r], R.vdv[r], R.vq[r], s, a, b, c, d);
}

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

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__device__ __forceinline__ void strided_tile_general(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, const LineGeom &g, int Lp, int LINES, int tiles_inner, long tile,
    const double *__restrict__ xlo, const double *__restrict__ xhi, const SweepScal &s, double *sm,
    const Fuse &fz = Fuse())
{
    const int tid = threadIdx.x;
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < g.n_inner;
    const long base = to * g.outer_stride + kcol;
    const int r0 = sg * M;
    const long line_id = to * (long)g.n_inner + kcol;

    double a[M], b[M], c[M], d[M];
    {
        SegRaw<M> R;
        load_segment_raw<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, g, base, r0, active, s, R, fz);
#pragma unroll
        for (int r = 0; r < M; ++r) assemble_one<M, HAS_DIR, HAS_Q>(R, r, g.lbit, s, a[r], b[r], c[r], d[r]);
    }
    // line ends: fold the coupling to the neighbouring GPU's row into the right-hand side
    if (r0 == 0) {
        if (xlo != nullptr && active) d[0] = __builtin_fma(-a[0], xlo[line_id], d[0]);
        a[0] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (r0 + r == g.n - 1) {
            if (xhi != nullptr && active) d[r] = __builtin_fma(-c[r], xhi[line_id], d[r]);
            c[r] = 0.0;
        }
    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);
    double xL, xS;
    tile_separators(sm, tid, kk, sg, Lp, LINES, a[M - 1], b[M - 1], c[M - 1], d[M - 1], k, xL, xS);
    double x[M];
    back_solve<M>(a, c, d, ip, xL, xS, x);
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (active && (r0 + r) < g.n) out[base + (long)(r0 + r) * g.stride] = x[r];
}

// GENERAL kernel: every tile (queue == nullptr) or the tiles a FAST kernel queued
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512) void k_sweep_strided(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ xlo, const double *__restrict__ xhi, SweepScal s, const unsigned *__restrict__ queue,
    int ratio, int tiles_inner_f, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    if (queue == nullptr) {
        strided_tile_general<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES, tiles_inner,
                                                      xcd_chunk_tile(blockIdx.x, ntiles), xlo, xhi, s, sm, fz);
    } else {
        // a queued unit is a tile of the FAST kernel = `ratio` adjacent tiles of this kernel
        const long cnt = (long)queue[0] * ratio;
        for (long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const long u = queue[1 + (unsigned)i / (unsigned)ratio];
            const long to = (long)((unsigned)u / (unsigned)tiles_inner_f);
            const long tig = (u - to * tiles_inner_f) * ratio + ((unsigned)i % (unsigned)ratio);
            if (tig < tiles_inner)
                strided_tile_general<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES,
                                                              tiles_inner, to * tiles_inner + tig, xlo, xhi, s, sm, fz);
            __syncthreads();   // the LDS arrays are reused by the next tile
        }
    }
}

// FAST strided kernels, part 1: load the segment's `in` rows and classify it.  Only the flags of row 0 and of the
// separator row are kept (the interior rows just have to be uniform).
// Addressing: element (row sg*M + r, column kcol) = [tile base + r*stride] (block-uniform -> scalar registers)
//             + voff, voff = sg*M*stride + kk a per-thread 32-bit offset that is the same for every row and array.
template <int M, bool HAS_DIR>
__device__ __forceinline__ bool fast_segment_load(const double *__restrict__ in_t, const uint8_t *__restrict__ flags_t,
                                                  const uint8_t *__restrict__ dmask_t, const LineGeom &g, unsigned voff,
                                                  int r0, bool active, double (&d)[M], unsigned &f0, unsigned &fS,
                                                  bool &dirS, int &kind, int &Lm)
{
    // kind: the segment class (SEG_*): a padding segment (r0 >= n: the line has fewer than Lp segments) owns no rows, a
    // segment whose rows are all outside the mask is M identity rows, TAIL / HEAD are crossed by the surface once
    const bool pad = active && r0 >= g.n;
    bool uni = active && (r0 + M <= g.n);
    const unsigned FULL = 1u | (3u << g.lbit), ROW0 = 1u | (2u << g.lbit);
    f0 = 0; fS = 0;
    unsigned inm = 0;
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const bool ok = active && (r0 + r) < g.n;
        const unsigned f = ok ? (flags_t + (size_t)r * g.stride)[voff] : 0u;
        d[r] = ok ? (in_t + (size_t)r * g.stride)[voff] : 0.0;
        inm |= (f & 1u) << r;
        if (r == 0) { f0 = f; uni = uni && ((f & ROW0) == ROW0); }
        else if (r == M - 1) fS = f;
        else uni = uni && ((f & FULL) == FULL);
    }
    bool nodir = true;
    dirS = false;
    if (HAS_DIR) {
#pragma unroll
        for (int r = 0; r < M - 1; ++r)
            nodir = nodir && (pad || inm == 0u || (dmask_t + (size_t)r * g.stride)[voff] == 0);
        dirS = active && inm != 0u && (r0 + M - 1) < g.n && (dmask_t + (size_t)(M - 1) * g.stride)[voff] != 0;
    }
    Lm = 0;
    if (pad) kind = SEG_PAD;
    else if (!active || r0 + M > g.n) kind = SEG_NONE;
    else if (uni) kind = nodir ? SEG_UNI : SEG_NONE;
    else {
        kind = classify_mixed<M>(inm, f0, g.lbit, Lm);
        if (kind >= SEG_TAIL && !nodir) kind = SEG_NONE;
    }
    return kind != SEG_NONE;
}

// Buffer addressing (raw_buffer_load/store: 128-bit descriptor + per-thread 32-bit byte offset + scalar byte offset):
// a strided tile touches M rows x several arrays, and with flat global loads every one of them costs 64-bit address
// arithmetic in the VALU (measured: half of the fused kernel's VALU instructions); here the row offsets live in
// scalar registers and one per-thread offset serves every load.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x)
{
    u32x2 v;
    v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, ADI_STORE_AUX);
}

// The same for whole tiles (block-uniform precondition: every lane active, every thread owns M rows), buffer
// addressing: scalar row offsets, one per-thread offset, no per-row predicates and no 64-bit address arithmetic.
template <int M, bool HAS_DIR>
__device__ __forceinline__ bool fast_segment_load_buf(const double *__restrict__ in_t, const uint8_t *__restrict__ flags_t,
                                                      const uint8_t *__restrict__ dmask_t, const LineGeom &g, unsigned voff,
                                                      double (&d)[M], unsigned &f0, unsigned &fS, bool &dirS, int &kind,
                                                      int &Lm)
{
    const unsigned FULL = 1u | (3u << g.lbit), ROW0 = 1u | (2u << g.lbit);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void *)in_t, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void *)flags_t, 0, 0x7fffffff, 0x00020000);
    const unsigned st = (unsigned)g.stride;
    bool uni = true;
    unsigned inm = 0;
    f0 = 0; fS = 0;
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const unsigned f = __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * st, ADI_LOAD_AUX);
        d[r] = buf_load_f64_once(rT, voff * 8u, (unsigned)r * st * 8u);
        inm |= (f & 1u) << r;
        if (r == 0) { f0 = f; uni = uni && ((f & ROW0) == ROW0); }
        else if (r == M - 1) fS = f;
        else uni = uni && ((f & FULL) == FULL);
    }
    bool nodir = true;
    dirS = false;
    if (HAS_DIR) {
        const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void *)dmask_t, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < M - 1; ++r)
            nodir = nodir && (inm == 0u || __builtin_amdgcn_raw_buffer_load_b8(rD, voff, (unsigned)r * st, 0) == 0);
        dirS = inm != 0u && __builtin_amdgcn_raw_buffer_load_b8(rD, voff, (unsigned)(M - 1) * st, 0) != 0;
    }
    Lm = 0;
    if (uni) kind = nodir ? SEG_UNI : SEG_NONE;
    else {
        kind = classify_mixed<M>(inm, f0, g.lbit, Lm);
        if (kind >= SEG_TAIL && !nodir) kind = SEG_NONE;
    }
    return kind != SEG_NONE;
}

// The same with the explicit stage folded in (FUSE kernels): d <- R0 = T + f*(Lx+Ly+Lz) of this thread's M rows.
// Preconditions (block-uniform, checked by the caller): the tile is whole -- LINES == 16 active lines, every thread owns
// M rows of the line (Lp*M == n) -- so no load needs a per-row predicate.  Neighbour loads do not wait for the flags:
// an address is read whenever it lies inside [vlo, vhi) and the value is used only where the flags byte says the
// neighbour exists; only the first row of a line can fall below vlo and only the last row above vhi.
// A 16-lane DPP row = the 16 lines of one segment: k-neighbours come from the adjacent lanes (row_shr/row_shl), the two
// outside the tile are loaded transposed (lane kk fetches the pair of row kk) and handed to lanes 0 / 15 with
// row_newbcast as the `old` operand of the shift, which is what the out-of-row lane keeps.
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

template <int M, bool HAS_DIR, bool MIXED = true>
__device__ __forceinline__ bool fast_segment_load_fused(const double *__restrict__ in, const uint8_t *__restrict__ flags_t,
                                                        const uint8_t *__restrict__ dmask_t, const LineGeom &g,
                                                        unsigned voff, int r0, int kk, long tbase, const Fuse &fz,
                                                        double (&d)[M], unsigned &f0, unsigned &fS, bool &dirS, int &kind,
                                                        int &Lm)
{
#pragma clang fp contract(off)
    constexpr int LINES = 16;
    const unsigned FULL = 1u | (3u << g.lbit), ROW0 = 1u | (2u << g.lbit);
    // the state through a descriptor over the window [wlo, wlo + wbytes/8) of `in` (host: covers every neighbour of
    // the box that exists in memory, < 4 GiB); flags through one based at the tile
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void *)(in + fz.wlo), 0, (int)fz.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void *)flags_t, 0, 0x7fffffff, 0x00020000);
    const unsigned vb = voff * 8u;                                   // this thread's row 0, bytes from the tile base
    const unsigned R0 = (unsigned)((tbase - fz.wlo) * 8);            // tile base, bytes from the window start (scalar)
    const unsigned st8 = (unsigned)(g.stride * 8), sy8 = (unsigned)(fz.sy * 8);
    // Every load below is unconditional.  Rows whose neighbour always lies inside the window take the scalar row offset;
    // the few that can fall outside it (first row: i-1, j-1, k0-1; last row: i+1, j+1, k0+16) carry the whole offset in
    // the per-thread register, where the descriptor's range check turns an address before or after the window into a
    // load of 0 -- such a neighbour does not exist and the flags byte says so.
    unsigned fb[M];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        fb[r] = __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * (unsigned)g.stride, ADI_LOAD_AUX);
        d[r] = buf_load_f64(rT, vb, R0 + (unsigned)r * st8);
    }
    const unsigned vw = vb + R0;                                     // this thread's row 0, bytes from the window start
    const unsigned vl = vw + (unsigned)(M - 1) * st8;                // its last row
    const double tim = buf_load_f64(rT, vw - st8, 0u);
    const double tip = buf_load_f64(rT, vl + st8, 0u);
    // k-neighbours outside the tile, loaded transposed: lane kk fetches the pair of row kk (columns k0-1, k0+16)
    const unsigned ve = R0 + (unsigned)(r0 + (int)(threadIdx.x & 15u)) * st8;
    const double eL = buf_load_f64(rT, ve - 8u, 0u);
    const double eR = buf_load_f64(rT, ve + LINES * 8u, 0u);
    bool uni = true, full = true;
    unsigned inm = 0;                               // MIXED: bit r = row r in the mask; otherwise just "any row in the mask"
#pragma unroll
    for (int r = 0; r < M; ++r) {
        full = full && (fb[r] == 0x7fu);
        if (MIXED) inm |= (fb[r] & 1u) << r;
        else inm |= fb[r];
        if (r == 0) uni = uni && ((fb[r] & ROW0) == ROW0);
        else if (r < M - 1) uni = uni && ((fb[r] & FULL) == FULL);
    }
    if (!MIXED) inm &= 1u;
    f0 = fb[0]; fS = fb[M - 1];
    const bool wave_full = __all(full);                     // every cell of this wave has its six neighbours
    // j-neighbour rows: a software pipeline D rows deep (they are L2 hits -- the tiles of the adjacent j-rows run next
    // door on the same XCD -- so a short pipeline covers their latency; a register pair per row in flight).  The
    // sched_barriers pin the order: without them the scheduler hoists every load to the top and spills.
    constexpr int D = (M >= 8) ? ADI_FUSE_D : M;
    auto load_jm = [&](int r) -> double {
        return (r == 0) ? buf_load_f64(rT, vw - sy8, 0u) : buf_load_f64(rT, vb, R0 + (unsigned)r * st8 - sy8);
    };
    auto load_jp = [&](int r) -> double {
        return (r == M - 1) ? buf_load_f64(rT, vl + sy8, 0u) : buf_load_f64(rT, vb, R0 + (unsigned)r * st8 + sy8);
    };
    double hm[D], hp[D];
#pragma unroll
    for (int q = 0; q < D; ++q) { hm[q] = load_jm(q); hp[q] = load_jp(q); }
    double prev = tim;
#pragma unroll
    for (int r = 0; r < M; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        const double cur = d[r];
        const double nxt = (r < M - 1) ? d[r + 1] : tip;     // still the state: rows are overwritten in order
        const double jm = hm[r % D], jp = hp[r % D];
        const double km = dpp_mov<0x111>(row_bcast(eL, r), cur), kp = dpp_mov<0x101>(row_bcast(eR, r), cur);
        if (wave_full) {
            // lap_axis with both neighbours present, same operation order: ((0 + lo) + hi - 2*t) * invdx2
            const double c2 = 2.0 * cur;
            const double L0 = (((0.0 + prev) + nxt) - c2) * fz.invdx2;
            const double L1 = (((0.0 + jm) + jp) - c2) * fz.invdx2;
            const double L2 = (((0.0 + km) + kp) - c2) * fz.invdx2;
            d[r] = cur + fz.f * ((L0 + L1) + L2);
        } else {
            d[r] = explicit_cell(fb[r], cur, prev, nxt, jm, jp, km, kp, fz);
        }
        prev = cur;
        if (r + D < M) {
            __builtin_amdgcn_sched_barrier(0);
            hm[r % D] = load_jm(r + D); hp[r % D] = load_jp(r + D);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    bool nodir = true;
    dirS = false;
    if (HAS_DIR) {
#pragma unroll
        for (int r = 0; r < M - 1; ++r)
            nodir = nodir && (inm == 0u || (dmask_t + (size_t)r * g.stride)[voff] == 0);
        dirS = inm != 0u && (dmask_t + (size_t)(M - 1) * g.stride)[voff] != 0;
    }
    Lm = 0;
    if (uni) kind = nodir ? SEG_UNI : SEG_NONE;
    else if (!MIXED) kind = (inm == 0u) ? SEG_OFF : SEG_NONE;
    else {
        kind = classify_mixed<M>(inm, f0, g.lbit, Lm);     // rows outside the mask have R0 = T: identity rows
        if (kind >= SEG_TAIL && !nodir) kind = SEG_NONE;
    }
    return kind != SEG_NONE;
}

// part 2: the two general rows of a uniform segment (row 0 and the separator)
template <int M, bool HAS_DIR, bool HAS_Q>
__device__ __forceinline__ void fast_segment_ends(const double *__restrict__ coeff, const double *__restrict__ dval,
                                                  const double *__restrict__ qf, const LineGeom &g, long base, int r0,
                                                  unsigned f0, unsigned fS, bool dirS, const SweepScal &s,
                                                  double (&d)[M], double &a0, double &b0, double &aS, double &bS,
                                                  double &cS)
{
    const long p0 = base + (long)r0 * g.stride, pS = base + (long)(r0 + M - 1) * g.stride;
    const bool e0 = axis_exposed(f0, g.lbit), eS = axis_exposed(fS, g.lbit);
    const double co0 = e0 ? coeff[p0] : 0.0, coS = eS ? coeff[pS] : 0.0;
    double q0 = 0.0, qS = 0.0, dvS = 0.0;
    if (HAS_Q) { q0 = e0 ? qf[p0] : 0.0; qS = eS ? qf[pS] : 0.0; }
    if (HAS_DIR) dvS = dirS ? dval[pS] : 0.0;
    double c0;
    assemble_row<HAS_DIR, HAS_Q>(f0 & 1u, (f0 >> g.lbit) & 1u, (f0 >> (g.lbit + 1)) & 1u, false, d[0], co0, 0.0, q0, s,
                                 a0, b0, c0, d[0]);
    assemble_row<HAS_DIR, HAS_Q>(fS & 1u, (fS >> g.lbit) & 1u, (fS >> (g.lbit + 1)) & 1u, dirS, d[M - 1], coS, dvS, qS,
                                 s, aS, bS, cS, d[M - 1]);
}

// FAST kernel (sparse packs): tiles whose every segment is uniform-interior (see k_sweep_contig_fast)
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool MIXED = true>
__global__ __launch_bounds__(512, FUSE ? ADI_FUSE_OCC : 1) void k_sweep_strided_fast(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ xlo, const double *__restrict__ xhi, SweepScal s, unsigned *__restrict__ queue,
    UniC<M> U, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    if (FUSE && fz.kg > 0) tile = tile_jfast(tile, fz);
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < g.n_inner;
    const long base = to * g.outer_stride + kcol;
    const int r0 = sg * M;
    const long line_id = to * (long)g.n_inner + kcol;

    double d[M];
    unsigned f0, fS;
    bool dirS;
    // block-uniform tile base (scalar) + one 32-bit per-thread offset for every row of every array
    const long tbase = to * g.outer_stride + (long)ti * LINES;
    const unsigned voff = (unsigned)((long)r0 * g.stride + kk);
    const bool pad = r0 >= g.n;                    // this thread's segment lies beyond the end of the line
    int kind = SEG_NONE, Lm = 0;                   // segment class (classify_mixed) and length of a mixed run
    bool lane_fast;
    if constexpr (FUSE) {
        // whole tiles only (block-uniform): anything else goes to the GENERAL kernel before a single load is issued
        if (LINES != 16 || (ti + 1) * LINES > g.n_inner || g.n % M != 0) {
            if (tid == 0) enqueue_unit(queue, (unsigned)tile);
            return;
        }
        // a padding segment (line with fewer than Lp segments) re-reads segment 0 -- valid addresses, values unused
        const int r0e = pad ? 0 : r0;
        lane_fast = fast_segment_load_fused<M, HAS_DIR, MIXED>(in, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g,
                                                               pad ? (unsigned)kk : voff, r0e, kk, tbase, fz, d, f0, fS, dirS,
                                                               kind, Lm) || pad;
        if (pad) kind = SEG_PAD;
    } else {
        // whole tiles whose rows fit 31-bit byte offsets take the buffer-addressed loader (block-uniform choice)
        const bool whole = kBufStrided && (ti + 1) * LINES <= g.n_inner && Lp * M == g.n &&
                           (long)g.n * g.stride * 8 < 0x7fffffffL;
        if (whole)
            lane_fast = fast_segment_load_buf<M, HAS_DIR>(in + tbase, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g, voff, d,
                                                          f0, fS, dirS, kind, Lm);
        else
            lane_fast = fast_segment_load<M, HAS_DIR>(in + tbase, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g, voff, r0,
                                                      active, d, f0, fS, dirS, kind, Lm);
    }
    if (pad) { f0 = 0; fS = 0; dirS = false; }       // (the fused loader showed a padding thread segment 0's flags)
    if (!MIXED && kind >= SEG_TAIL) lane_fast = false;
    if (!__syncthreads_and(lane_fast)) {
        if (tid == 0) enqueue_unit(queue, (unsigned)tile);
        return;
    }
    double a0, b0, aS, bS, cS;
    fast_segment_ends<M, HAS_DIR, HAS_Q>(coeff, dval, qf, g, base, r0, f0, fS, dirS, s, d, a0, b0, aS, bS, cS);
    if (r0 == 0) {
        if (xlo != nullptr) d[0] = __builtin_fma(-a0, xlo[line_id], d[0]);
        a0 = 0.0;
    }
    if (r0 + M == g.n) {
        if (xhi != nullptr) d[M - 1] = __builtin_fma(-cS, xhi[line_id], d[M - 1]);
        cS = 0.0;
    }
    Cond k;
    double kappa;
    condense_uniform<M>(U, a0, b0, d, k, kappa);
    const bool off = kind == SEG_OFF;              // a segment outside the mask: identity rows, x = in
    if (pad || off) {                              // identity block: nothing reaches the real segments
        k.gF = k.aF = k.cF = k.gL = k.aL = k.cL = 0.0;
        kappa = 0.0; aS = 0.0; bS = 1.0; cS = 0.0;
        if (pad) d[M - 1] = 0.0;
    }
    double2 bmod = make_double2(1.0, 1.0);
    if constexpr (MIXED) {
        if (kind >= SEG_TAIL)                      // the surface crosses the segment once (adi_core.hpp, mixed_*)
            mixed_lane_condense<M, HAS_Q>(kind, Lm, U, s, coeff + base + (long)r0 * g.stride,
                                          HAS_Q ? qf + base + (long)r0 * g.stride : qf, g.stride, a0, b0, d, bmod, k);
    }
    double xL, xS;
    tile_separators(sm, tid, kk, sg, Lp, LINES, aS, bS, cS, d[M - 1], k, xL, xS);
    if (kind == SEG_UNI || kind == SEG_PAD) back_solve_uniform<M>(U, a0, kappa, d, xL, xS);
    else if constexpr (MIXED) {
        if (kind >= SEG_TAIL) mixed_lane_back_solve<M>(kind, Lm, U, bmod, a0, d, xL, xS);
    }
    if (pad) return;                               // (after the last barrier)
    double *out_t = out + tbase;
    if (FUSE || (kBufStrided && (long)g.n * g.stride * 8 < 0x7fffffffL)) {
        const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)out_t, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < M; ++r) buf_store_f64(rO, voff * 8u, (unsigned)r * (unsigned)(g.stride * 8), d[r]);
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) (out_t + (size_t)r * g.stride)[voff] = d[r];
    }
}

// ------------------------------------------------------------------------------------------------
// K5a: slab condensation (pass A of a sweep whose lines continue on neighbouring GPUs).  Same loads and
// per-thread work as K2's phase 1, but every thread condenses ALL its M rows and the Lp blocks of a line
// are merged by an ordered tree reduction; lane 0 writes the six numbers that describe the slab's part of
// the line to its neighbours:  x_first = gF - aF*xl - cF*xr,  x_last = gL - aL*xl - cL*xr.
// Requires n % M == 0 (whole segments).  cond: [6][nlines] dense.
// ------------------------------------------------------------------------------------------------
// lane 0 of every line: ordered merge of the Lp block condensations -> six numbers per line
__device__ __forceinline__ void tile_reduce_store(double *sm, int tid, int kk, int sg, int Lp, int LINES, const Cond &k,
                                                  int nblk, long to, int ti, const LineGeom &g, long nlines,
                                                  double *__restrict__ cond)
{
    const int ld = Lp + 1;
    const int plane = LINES * ld;
    {
        const int w = kk * ld + sg;
        sm[w] = k.gF; sm[plane + w] = k.aF; sm[2 * plane + w] = k.cF;
        sm[3 * plane + w] = k.gL; sm[4 * plane + w] = k.aL; sm[5 * plane + w] = k.cL;
    }
    __syncthreads();
    const int pl = tid >> (__ffs(Lp) - 1), ps = tid & (Lp - 1);   // Lp is a power of two
    const int w = pl * ld + ps;
    Cond q;
    q.gF = sm[w]; q.aF = sm[plane + w]; q.cF = sm[2 * plane + w];
    q.gL = sm[3 * plane + w]; q.aL = sm[4 * plane + w]; q.cL = sm[5 * plane + w];
    q = reduce_cond(q, ps, Lp, nblk);
    const int kc2 = ti * LINES + pl;
    if (ps == 0 && kc2 < g.n_inner) {
        const long id = to * (long)g.n_inner + kc2;
        cond[id] = q.gF; cond[nlines + id] = q.aF; cond[2 * nlines + id] = q.cF;
        cond[3 * nlines + id] = q.gL; cond[4 * nlines + id] = q.aL; cond[5 * nlines + id] = q.cL;
    }
}

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__device__ __forceinline__ void condense_tile_general(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ cond, long nlines, const LineGeom &g, int Lp, int LINES, int tiles_inner, long tile,
    const SweepScal &s, double *sm, const Fuse &fz = Fuse())
{
    const int tid = threadIdx.x;
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < g.n_inner;
    const long base = to * g.outer_stride + kcol;
    const int r0 = sg * M;
    double a[M], b[M], c[M], d[M];
    {
        // same assembly as the solve pass, but the end couplings stay in a[0] / c[n-1] (they are the
        // aF, aL / cF, cL of the slab)
        SegRaw<M> R;
        load_segment_raw<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, g, base, r0, active, s, R, fz);
        if (FUSE && fz.r0_out != nullptr) {
#pragma unroll
            for (int r = 0; r < M; ++r)
                if (active && (r0 + r) < g.n) fz.r0_out[base + (long)(r0 + r) * g.stride] = R.vin[r];
        }
#pragma unroll
        for (int r = 0; r < M; ++r) assemble_one<M, HAS_DIR, HAS_Q>(R, r, g.lbit, s, a[r], b[r], c[r], d[r]);
    }
    Cond k;
    condense_full<M>(a, b, c, d, k);
    tile_reduce_store(sm, tid, kk, sg, Lp, LINES, k, g.n / M, to, ti, g, nlines, cond);
}

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512) void k_condense_strided(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ cond, long nlines, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    SweepScal s, const unsigned *__restrict__ queue, int ratio, int tiles_inner_f, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    if (queue == nullptr) {
        condense_tile_general<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, cond, nlines, g, Lp, LINES,
                                                       tiles_inner, xcd_chunk_tile(blockIdx.x, ntiles), s, sm, fz);
    } else {
        const long cnt = (long)queue[0] * ratio;
        for (long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const long u = queue[1 + (unsigned)i / (unsigned)ratio];
            const long to = (long)((unsigned)u / (unsigned)tiles_inner_f);
            const long tig = (u - to * tiles_inner_f) * ratio + ((unsigned)i % (unsigned)ratio);
            if (tig < tiles_inner)
                condense_tile_general<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, cond, nlines, g, Lp,
                                                               LINES, tiles_inner, to * tiles_inner + tig, s, sm, fz);
            __syncthreads();
        }
    }
}

// FAST pass A: uniform-interior segments (see k_sweep_strided_fast); the block of a thread = its M-1 uniform
// interior rows merged with its general separator row.
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false>
__global__ __launch_bounds__(512, FUSE ? ADI_FUSE_OCC : 1) void k_condense_strided_fast(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ cond, long nlines, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    SweepScal s, unsigned *__restrict__ queue, UniC<M> U, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    if (FUSE && fz.kg > 0) tile = tile_jfast(tile, fz);
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < g.n_inner;
    const long base = to * g.outer_stride + kcol;
    const int r0 = sg * M;

    double d[M];
    unsigned f0, fS;
    bool dirS;
    // block-uniform tile base (scalar) + one 32-bit per-thread offset for every row of every array
    const long tbase = to * g.outer_stride + (long)ti * LINES;
    const unsigned voff = (unsigned)((long)r0 * g.stride + kk);
    const bool pad = r0 >= g.n;                    // this thread's segment lies beyond the end of the line
    int kind = SEG_NONE, Lm = 0;                   // segment class (classify_mixed) and length of a mixed run
    bool lane_fast;
    if constexpr (FUSE) {
        // whole tiles only (block-uniform): anything else goes to the GENERAL kernel before a single load is issued
        if (LINES != 16 || (ti + 1) * LINES > g.n_inner || g.n % M != 0) {
            if (tid == 0) enqueue_unit(queue, (unsigned)tile);
            return;
        }
        // a padding segment (line with fewer than Lp segments) re-reads segment 0 -- valid addresses, values unused
        const int r0e = pad ? 0 : r0;
        lane_fast = fast_segment_load_fused<M, HAS_DIR>(in, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g,
                                                        pad ? (unsigned)kk : voff, r0e, kk, tbase, fz, d, f0, fS, dirS, kind,
                                                        Lm) || pad;
        if (pad) kind = SEG_PAD;
    } else {
        // whole tiles whose rows fit 31-bit byte offsets take the buffer-addressed loader (block-uniform choice)
        const bool whole = kBufStrided && (ti + 1) * LINES <= g.n_inner && Lp * M == g.n &&
                           (long)g.n * g.stride * 8 < 0x7fffffffL;
        if (whole)
            lane_fast = fast_segment_load_buf<M, HAS_DIR>(in + tbase, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g, voff, d,
                                                          f0, fS, dirS, kind, Lm);
        else
            lane_fast = fast_segment_load<M, HAS_DIR>(in + tbase, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g, voff, r0,
                                                      active, d, f0, fS, dirS, kind, Lm);
    }
    if (pad) { f0 = 0; fS = 0; dirS = false; }       // (the fused loader showed a padding thread segment 0's flags)
    if (!__syncthreads_and(lane_fast)) {
        if (tid == 0) enqueue_unit(queue, (unsigned)tile);
        return;
    }
    if constexpr (FUSE) {
        if (fz.r0_out != nullptr && !pad) {
            const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)(fz.r0_out + tbase), 0, 0x7fffffff,
                                                                                0x00020000);
#pragma unroll
            for (int r = 0; r < M; ++r) buf_store_f64(rO, voff * 8u, (unsigned)r * (unsigned)(g.stride * 8), d[r]);
        }
    }
    double a0, b0, aS, bS, cS;
    fast_segment_ends<M, HAS_DIR, HAS_Q>(coeff, dval, qf, g, base, r0, f0, fS, dirS, s, d, a0, b0, aS, bS, cS);
    Cond ki;
    double kappa;
    condense_uniform<M>(U, a0, b0, d, ki, kappa);
    if (pad) {                                     // finite values; tile_reduce_store ignores blocks >= n / M
        ki.gF = ki.aF = ki.cF = ki.gL = ki.aL = ki.cL = 0.0;
        aS = 0.0; bS = 1.0; cS = 0.0; d[M - 1] = 0.0;
    } else if (kind == SEG_OFF) {                  // segment outside the mask: M identity rows
        ki.gF = d[0]; ki.gL = d[M - 2];
        ki.aF = ki.cF = ki.aL = ki.cL = 0.0;
        aS = 0.0; bS = 1.0; cS = 0.0;
    } else if (kind >= SEG_TAIL) {                 // the surface crosses the segment once
        double2 bmod;
        mixed_lane_condense<M, HAS_Q>(kind, Lm, U, s, coeff + base + (long)r0 * g.stride,
                                      HAS_Q ? qf + base + (long)r0 * g.stride : qf, g.stride, a0, b0, d, bmod, ki);
    }
    const double ib = frcp(bS);
    Cond rowc;
    rowc.gF = rowc.gL = d[M - 1] * ib;
    rowc.aF = rowc.aL = aS * ib;
    rowc.cF = rowc.cL = cS * ib;
    const Cond k = merge_cond(ki, rowc);
    tile_reduce_store(sm, tid, kk, sg, Lp, LINES, k, g.n / M, to, ti, g, nlines, cond);
}

// K5b: generic slab condensation, one thread per line, two serial recurrences (any n; reads rows twice).
template <bool HAS_DIR, bool HAS_Q>
__global__ __launch_bounds__(256) void k_condense_generic(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ cond, long nlines, LineGeom g, long inner_stride, SweepScal s,
    const unsigned *__restrict__ list = nullptr, long lb = 0, long nsel = 0)
{
    // list != nullptr: only the listed lines that fall into [lb, lb + nsel), written to a [6][nsel] block
    long lid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long oid = lid, ostr = nlines;
    if (list != nullptr) {
        if (lid >= (long)list[0]) return;
        lid = list[1 + lid];
        if (lid < lb || lid >= lb + nsel) return;
        oid = lid - lb; ostr = nsel;
    }
    if (lid >= nlines) return;
    const long o = lid / g.n_inner, kc = lid - o * g.n_inner;
    const long base = o * g.outer_stride + kc * inner_stride;
    const int n = g.n;
    double a0 = 0.0, cn = 0.0;
    // top-down: last component of B^-1 d, B^-1 e_0 ; 1/pivot_last
    double ip = 0.0, y = 0.0, e = 1.0, cprev = 0.0;
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * g.stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> g.lbit) & 1u, (f >> (g.lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0,
                                     in[p], coeff[p], HAS_DIR ? dval[p] : 0.0, HAS_Q ? qf[p] : 0.0, s, a, b, c, d);
        if (r == 0) { a0 = a; ip = 1.0 / b; y = d; }
        else { const double w = a * ip; ip = 1.0 / (b - w * cprev); y = d - w * y; e = -w * e; }
        cprev = c;
        if (r == n - 1) cn = c;
    }
    const double gL = y * ip, aL = a0 * (e * ip), cL = cn * ip;
    // bottom-up
    double jp = 0.0, z = 0.0, f2 = 1.0, anext = 0.0;
    for (int r = n - 1; r >= 0; --r) {
        const long p = base + (long)r * g.stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> g.lbit) & 1u, (f >> (g.lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0,
                                     in[p], coeff[p], HAS_DIR ? dval[p] : 0.0, HAS_Q ? qf[p] : 0.0, s, a, b, c, d);
        if (r == n - 1) { jp = 1.0 / b; z = d; }
        else { const double w = c * jp; jp = 1.0 / (b - w * anext); z = d - w * z; f2 = -w * f2; }
        anext = a;
    }
    cond[oid] = z * jp; cond[ostr + oid] = a0 * jp; cond[2 * ostr + oid] = cn * (f2 * jp);
    cond[3 * ostr + oid] = gL; cond[4 * ostr + oid] = aL; cond[5 * ostr + oid] = cL;
}

// K5c: interface solve.  cond_all: [nranks][6][nlines] (all-gathered).  For this rank, merge the slabs below
// and above it, solve the 2x2 system for its own first/last unknown and emit the neighbours' boundary
// values: xlo = last unknown of the slab below, xhi = first unknown of the slab above.
__global__ __launch_bounds__(256) void k_interface(const double *__restrict__ cond_all, int nranks, int rank,
                                                   long nlines, double *__restrict__ xlo, double *__restrict__ xhi)
{
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines) return;
    auto ld = [&](int r) {
        const double *q = cond_all + (long)r * 6 * nlines + id;
        Cond k;
        k.gF = q[0]; k.aF = q[nlines]; k.cF = q[2 * nlines]; k.gL = q[3 * nlines]; k.aL = q[4 * nlines];
        k.cL = q[5 * nlines];
        return k;
    };
    const Cond C = ld(rank);
    Cond P = {0, 0, 0, 0, 0, 0}, S = {0, 0, 0, 0, 0, 0};   // empty neighbours: decoupled zeros
    if (rank > 0) {
        P = ld(0);
        for (int r = 1; r < rank; ++r) P = merge_cond(P, ld(r));
    }
    if (rank < nranks - 1) {
        S = ld(nranks - 1);
        for (int r = nranks - 2; r > rank; --r) S = merge_cond(ld(r), S);
    }
    // unknowns f = x_first(C), l = x_last(C);  x_last(P) = P.gL - P.cL f ;  x_first(S) = S.gF - S.aF l
    //   f = C.gF - C.aF (P.gL - P.cL f) - C.cF (S.gF - S.aF l)
    //   l = C.gL - C.aL (P.gL - P.cL f) - C.cL (S.gF - S.aF l)
    const double m00 = 1.0 - C.aF * P.cL, m01 = -C.cF * S.aF;
    const double m10 = -C.aL * P.cL, m11 = 1.0 - C.cL * S.aF;
    const double r0 = C.gF - C.aF * P.gL - C.cF * S.gF;
    const double r1 = C.gL - C.aL * P.gL - C.cL * S.gF;
    const double idet = 1.0 / (m00 * m11 - m01 * m10);
    const double f = (r0 * m11 - m01 * r1) * idet;
    const double l = (m00 * r1 - m10 * r0) * idet;
    xlo[id] = P.gL - P.cL * f;
    xhi[id] = S.gF - S.aF * l;
}

// Neighbour-only form of the interface system, valid when the far-side couplings of the boundary windows
// (aL of the window that ends a slab, cF of the window that starts one) have decayed below rounding:
//   x_last(r)    = gL  - cL  * x_first(r+1)        (window = last rows of slab r)
//   x_first(r+1) = gF' - aF' * x_last(r)           (window = first rows of slab r+1)
// my_lo / my_hi: [6][nlines] condensations of this slab's first / last window; prev_hi: rows (gL,aL,cL) of the
// slab below; next_lo: rows (gF,aF) of the slab above.  NULL neighbour -> 0.
__global__ __launch_bounds__(256) void k_interface_pair(const double *__restrict__ my_lo, const double *__restrict__ my_hi,
                                                        const double *__restrict__ prev_hi, const double *__restrict__ next_lo,
                                                        long nlines, double *__restrict__ xlo, double *__restrict__ xhi)
{
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines) return;
    double lo = 0.0, hi = 0.0;
    if (prev_hi) {
        const double gLp = prev_hi[id], cLp = prev_hi[2 * nlines + id];
        const double gF = my_lo[id], aF = my_lo[nlines + id];
        lo = (gLp - cLp * gF) / (1.0 - cLp * aF);
    }
    if (next_lo) {
        const double gFn = next_lo[id], aFn = next_lo[nlines + id];
        const double gL = my_hi[3 * nlines + id], cL = my_hi[5 * nlines + id];
        hi = (gFn - aFn * gL) / (1.0 - aFn * cL);
    }
    xlo[id] = lo;
    xhi[id] = hi;
}

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
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * g.stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> g.lbit) & 1u, (f >> (g.lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0, in[p],
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

static int explicit_jr()
{
    static int v = 0;
    if (v == 0) {
        const char *e = getenv("ADI_EXPLICIT_JR");
        v = e ? atoi(e) : 8;
        if (v < 1 || v > 64) v = 8;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_explicit_v2(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                     double *__restrict__ R0, Lay L, double invdx2, double f,
                                                     int jslab, int ktiles, long ntiles, int kExplicitJR)
{
#pragma clang fp contract(off)
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int jc_per_slab = (jslab + kExplicitJR - 1) / kExplicitJR;
    const long per_plane = (long)jc_per_slab * ktiles;
    const long per_slab = per_plane * nx;
    // block-uniform 32-bit arithmetic (64-bit divisions cost hundreds of scalar instructions per block)
    const unsigned t32 = (unsigned)tile, pps = (unsigned)per_plane, psl = (unsigned)per_slab;
    const int slab = (int)(t32 / psl);
    unsigned rem = t32 - (unsigned)slab * psl;
    const int i = (int)(rem / pps);
    rem -= (unsigned)i * pps;
    const int jc = (int)(rem / (unsigned)ktiles), kt = (int)(rem - (unsigned)jc * (unsigned)ktiles);
    const int jbeg = slab * jslab + jc * kExplicitJR;
    int jend = jbeg + kExplicitJR;
    if (jend > (slab + 1) * jslab) jend = (slab + 1) * jslab;
    if (jend > ny) jend = ny;
    const int k0 = kt * 512 + 2 * (int)threadIdx.x;
    const bool kin = k0 < nz;            // nz is even: both cells of the pair are inside
    const int lane = threadIdx.x & 63;
    const long sx = L.sx, sy = nz;
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

// k_explicit_v3: 2.5-D blocking.  A block owns a tile of JT3 j-rows x 512 k and MARCHES ALONG i over a chunk of
// planes with a three-plane register window, so every T value is loaded from memory once per chunk (plus the two
// j-halo rows per plane, which the neighbouring tile of the same XCD has just touched); the j-neighbours of a row are
// in the same thread's registers, the k-neighbours in the adjacent lanes.  Same expression order as v2 / the reference.

template <int JT3>
__global__ __launch_bounds__(256) void k_explicit_v3(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                     double *__restrict__ R0, Lay L, double invdx2, double f,
                                                     int jslab, int ktiles, int ichunk, long ntiles, int i_begin,
                                                     int i_end)
{
#pragma clang fp contract(off)
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    // tile order: [slab][i-chunk][j-tile in slab][k-tile]
    const int jt_per_slab = (jslab + JT3 - 1) / JT3;
    const int nchunk = (i_end - i_begin + ichunk - 1) / ichunk;
    const long per_chunk = (long)jt_per_slab * ktiles;
    const long per_slab = per_chunk * nchunk;
    const unsigned t32 = (unsigned)tile, pch = (unsigned)per_chunk, psl = (unsigned)per_slab;
    const int slab = (int)(t32 / psl);
    unsigned rem = t32 - (unsigned)slab * psl;
    const int ic = (int)(rem / pch);
    rem -= (unsigned)ic * pch;
    const int jt = (int)(rem / (unsigned)ktiles), kt = (int)(rem - (unsigned)jt * (unsigned)ktiles);
    const int j0 = slab * jslab + jt * JT3;
    int jend = j0 + JT3;
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
    double2 tm[JT3], tc[JT3], tp[JT3];
    const long pbase = (long)j0 * sy + k0;
#pragma unroll
    for (int r = 0; r < JT3; ++r) {
        tm[r] = zero2; tc[r] = zero2; tp[r] = zero2;
        if (kin && j0 + r < jend) {
            tc[r] = *reinterpret_cast<const double2 *>(T + (long)i0 * sx + pbase + (long)r * sy);
            if (i0 > 0) tm[r] = *reinterpret_cast<const double2 *>(T + (long)(i0 - 1) * sx + pbase + (long)r * sy);
        }
    }
    for (int i = i0; i < i1; ++i) {
        const long p = (long)i * sx + pbase;
        unsigned fl[JT3];
        double2 hm = zero2, hp = zero2;
#pragma unroll
        for (int r = 0; r < JT3; ++r) {
            fl[r] = 0;
            if (kin && j0 + r < jend) {
                fl[r] = *reinterpret_cast<const uint16_t *>(flags + p + (long)r * sy);
                if (i + 1 < nx) tp[r] = *reinterpret_cast<const double2 *>(T + p + sx + (long)r * sy);
            }
        }
        if (kin) {
            if (j0 > 0) hm = *reinterpret_cast<const double2 *>(T + p - sy);
            if (jend < ny) hp = *reinterpret_cast<const double2 *>(T + p + (long)(jend - j0) * sy);
        }
#pragma unroll
        for (int r = 0; r < JT3; ++r) {
            const bool rin = j0 + r < jend;
            const long q = p + (long)r * sy;
            double kl = __shfl_up(tc[r].y, 1), kr = __shfl_down(tc[r].x, 1);
            if (lane == 0) kl = (kin && rin && k0 > 0) ? T[q - 1] : 0.0;
            if (lane == 63) kr = (kin && rin && k0 + 2 < nz) ? T[q + 2] : 0.0;
            const double2 jm = (r == 0) ? hm : tc[r > 0 ? r - 1 : 0];
            const double2 jp = (j0 + r + 1 == jend) ? hp : tc[r + 1 < JT3 ? r + 1 : JT3 - 1];
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
            if (kin && rin) *reinterpret_cast<double2 *>(R0 + q) = make_double2(r0v, r1v);
        }
#pragma unroll
        for (int r = 0; r < JT3; ++r) { tm[r] = tc[r]; tc[r] = tp[r]; }
    }
}

// k_explicit_v5 (default): v3 software-pipelined one plane ahead.  The loads issued while plane i is computed are those of
// plane i+2 (state), and of plane i+1 (flags, j-halo rows, k-edge values), so that a wave always has a full plane of
// loads in flight behind its arithmetic and stores instead of load -> wait -> compute -> store in sequence.
// The two k-edge values of a row come from ONE load (lane 0 fetches k0-1, lane 63 fetches k0+2).
// Measured at 512^3: v3 0.50 ms, v5 with 4-row tiles 0.47 ms (158 VGPRs, 3 waves/SIMD), with 2-row tiles 0.466 ms
// (96 VGPRs, 5 waves/SIMD) although those re-read twice as many j-halo rows: the stage is bound by loads in flight per
// wave, not by traffic -- a variant that shared the halo rows of a 16/32-row block tile through LDS (one barrier per
// plane) cut the traffic and ran no faster, so it was dropped; the re-read rows are served by L2 / Infinity Cache.
// DOTS (slab decomposition, pass A folded into the explicit stage): while marching, every line's R0 values are also
// accumulated into the two dot products the reduced interface system needs from a line whose rows are uniform --
// su = sum_i u[i] R0_i, sv = sum_i u[n-1-i] R0_i with u = first column of tridiag(-tg, 1+2tg, -tg)^-1 (wu, n values) --
// one partial pair per chunk of planes: part[chunk][2][ny*nz], summed in order by k_dots_finish.
template <int JT, bool DOTS = false>
__global__ __launch_bounds__(256) void k_explicit_v5(const double *__restrict__ T, const uint8_t *__restrict__ flags,
                                                     double *__restrict__ R0, Lay L, double invdx2, double f,
                                                     int jslab, int ktiles, int ichunk, long ntiles, int i_begin,
                                                     int i_end, const double *__restrict__ wu = nullptr,
                                                     double *__restrict__ part = nullptr, int i_org = 0, int n_line = 0)
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
            if (kin && rin) st_stream2(reinterpret_cast<double2 *>(R0 + q), make_double2(r0v, r1v));
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

// generic form (odd nz or unaligned views): one cell per thread
__global__ __launch_bounds__(256) void k_explicit(const double *__restrict__ T, const uint8_t *__restrict__ flags,
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
__global__ __launch_bounds__(256) void k_build_flags(const uint8_t *__restrict__ mask, Lay L, uint8_t *__restrict__ flags)
{
    int i, j, k;
    long p;
    if (!cell_of((long)blockIdx.x * blockDim.x + threadIdx.x, L, i, j, k, p)) return;
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

__global__ __launch_bounds__(256) void k_build_coeffs(const uint8_t *__restrict__ mask, Lay L, double A, double Ccell,
                                                      FaceSpec h, FaceSpec q, double *__restrict__ c0,
                                                      double *__restrict__ c1, double *__restrict__ c2,
                                                      double *__restrict__ q0, double *__restrict__ q1,
                                                      double *__restrict__ q2)
{
#pragma clang fp contract(off)
    int i, j, k;
    long p;
    if (!cell_of((long)blockIdx.x * blockDim.x + threadIdx.x, L, i, j, k, p)) return;
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
// lines per strided tile (8 B * lines contiguous per row); ADI_STRIDED_LINES overrides for tuning runs.
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

static int contig_rows_per_lane(int n)
{
    static int m16 = -1;
    if (m16 < 0) m16 = getenv("ADI_CONTIG_M16") ? 1 : 0;   // tuning knob
    if (m16 && n > 256 && n % 16 == 0) return 16;
    return n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : 16));
}
static int strided_rows_per_thread(int n)
{
    static int m16 = -1;
    if (m16 < 0) m16 = getenv("ADI_STRIDED_M16") ? 1 : 0;   // tuning knob: 16 rows per thread from n = 257
    if (m16 && n > 256) return 16;
    return n <= 16 ? 2 : (n <= 32 ? 4 : (n <= 512 ? 8 : 16));
}

static LineGeom line_geom(int axis, const Lay &L, long *inner_stride)
{
    LineGeom g;
    g.lbit = 1 + 2 * axis;
    if (axis == 0) { g.n = L.nx; g.stride = L.sx; g.n_inner = L.ny * L.nz; g.n_outer = 1; g.outer_stride = 0; *inner_stride = 1; }
    else if (axis == 1) { g.n = L.ny; g.stride = L.nz; g.n_inner = L.nz; g.n_outer = L.nx; g.outer_stride = L.sx; *inner_stride = 1; }
    else { g.n = L.nz; g.stride = 1; g.n_inner = L.ny; g.n_outer = L.nx; g.outer_stride = L.sx; *inner_stride = L.nz; }
    return g;
}

// sparse packs + a workspace for the unit queue -> FAST kernel first, GENERAL kernel on what it queued
static bool use_fast(const SweepScal &s, void *work, size_t work_bytes, long nunits)
{
    static int off = -1;
    if (off < 0) off = getenv("ADI_NO_FAST") ? 1 : 0;
    return !off && s.sparse && work != nullptr && work_bytes >= (size_t)(nunits + 1) * sizeof(unsigned);
}

template <int MF, bool HAS_DIR, bool HAS_Q>
static void launch_contig_fast(const double *in, const uint8_t *flags, const double *coeff, const uint8_t *dmask,
                               const double *dval, const double *qf, double *out, const Lay &L, SweepScal s, bool vec,
                               long nunits_f, unsigned *queue, hipStream_t st)
{
    const int Lpf = next_pow2(L.nz / MF);
    const unsigned grid = (unsigned)((nunits_f + 3) / 4);
    const UniC<MF> U = make_unic<MF>(s.tg);
    static int nocoal = -1;
    if (nocoal < 0) nocoal = getenv("ADI_NO_COAL") ? 1 : 0;
    const int lwf = 64 / Lpf;
    // coalesced + LDS-transposed access: whole units of contiguous lines (full last unit, no plane straddling)
    const bool coal = vec && !nocoal && MF >= 4 && Lpf * MF == L.nz && (L.ny % lwf == 0) &&
                      (((long)L.nx * L.ny) % lwf == 0);
    if (coal)
        hipLaunchKernelGGL((k_sweep_contig_fast<(MF >= 4 ? MF : 4), 2, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, make_unic<(MF >= 4 ? MF : 4)>(s.tg));
    else if (vec)
        hipLaunchKernelGGL((k_sweep_contig_fast<MF, 1, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, U);
    else
        hipLaunchKernelGGL((k_sweep_contig_fast<MF, 0, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags,
                           coeff, dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, U);
}

template <int M, bool HAS_DIR, bool HAS_Q>
static void launch_contig(const double *in, const uint8_t *flags, const double *coeff, const uint8_t *dmask,
                          const double *dval, const double *qf, double *out, const Lay &L, SweepScal s,
                          void *work, size_t work_bytes, hipStream_t st)
{
    const int n = L.nz;
    const long nlines = (long)L.nx * L.ny;
    const int Lp = next_pow2((n + M - 1) / M);
    const int lw = 64 / Lp;
    const long nunits = (nlines + lw - 1) / lw;
    const unsigned grid = (unsigned)((nunits + 3) / 4);
    const bool aligned = (((uintptr_t)in | (uintptr_t)coeff | (uintptr_t)out | (uintptr_t)dval | (uintptr_t)qf) & 15) == 0 &&
                         (((uintptr_t)flags | (uintptr_t)dmask) & 7) == 0 && (L.sx % 8 == 0);
    const bool vec = aligned && (n % M == 0);
    // FAST kernel: as many rows per lane as divide the line (16, else 8), so that one in-wave PCR serves several lines --
    // n = 512: two lines per wave, n = 256: four (with the GENERAL kernel's 4 rows per lane the PCR ran over 64 lanes per
    // line: 155 Gcell/s at 256^3 against 311 at nz = 512)
    static int fm = -1;
    if (fm < 0) { const char *e = getenv("ADI_CONTIG_FAST_M"); fm = e ? atoi(e) : 0; }
    int Mf = M;
    if (fm == 0) {
        if (n >= 128 && n % 16 == 0 && n / 16 <= 64) Mf = 16;
        else if (n >= 64 && n % 8 == 0 && n / 8 <= 64 && M < 8) Mf = 8;
    } else if (fm == 1) {
        Mf = (n > 256 && n % 16 == 0 && n / 16 <= 64) ? 16 : M;     // the earlier rule, for comparison
    }
    const int lwf = 64 / next_pow2((n + Mf - 1) / Mf);
    const long nunits_f = (nlines + lwf - 1) / lwf;
    const bool fast = use_fast(s, work, work_bytes, nunits_f) && (n % Mf == 0) && (lwf % lw == 0);
    unsigned *queue = fast ? (unsigned *)work : nullptr;
    unsigned ggrid = grid;
    if (fast) {
        (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        const bool vecf = aligned && (n % Mf == 0);
        if (Mf == 16 && M != 16)
            launch_contig_fast<16, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        else if (Mf == 8 && M != 8)
            launch_contig_fast<8, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        else
            launch_contig_fast<M, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        ggrid = grid < 2048u ? grid : 2048u;
    }
    const int ratio = fast ? lwf / lw : 1;
    // coalesced + LDS-transposed access with streaming loads/stores: 1.01 -> 0.92 ms on the dense general-pack sweep at
    // 512^3 (with the default cache policy the transposition alone gained nothing); ADI_GENERAL_COAL=0 turns it off
    static int gcoal = -1;
    if (gcoal < 0) { const char *e = getenv("ADI_GENERAL_COAL"); gcoal = (e && atoi(e) == 0) ? 0 : 1; }
    const bool coal = vec && gcoal && M >= 4 && Lp * M == n && (L.ny % lw == 0) && (nlines % lw == 0);
    if (coal)
        hipLaunchKernelGGL((k_sweep_contig<(M >= 4 ? M : 4), 2, HAS_DIR, HAS_Q>), dim3(ggrid), dim3(256), 0, st, in, flags,
                           coeff, dmask, dval, qf, out, L, Lp, s, nunits, queue, ratio);
    else if (vec)
        hipLaunchKernelGGL((k_sweep_contig<M, 1, HAS_DIR, HAS_Q>), dim3(ggrid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lp, s, nunits, queue, ratio);
    else
        hipLaunchKernelGGL((k_sweep_contig<M, 0, HAS_DIR, HAS_Q>), dim3(ggrid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lp, s, nunits, queue, ratio);
}

// Tiling of a strided sweep.  `lines` adjacent lines x all segments per workgroup.  The FAST kernel (uniform
// interior, ~70 VGPRs) takes Mf = 16 rows per thread for n > 256 so that 16 lines (whole 128-byte DRAM bursts per
// row) still fit a 512-thread workgroup; the GENERAL kernel keeps Mg = 8 rows (register budget) and, when it runs
// behind a FAST kernel, the same `lines`, so both see the same tile ids in the unit queue.
struct StridedPlan {
    int Mg, Lpg, lines_g, tiles_inner_g;    // GENERAL kernel: rows per thread, segments per line, lines per tile
    long ntiles_g;
    int Mf, Lpf, lines_f, tiles_inner_f;    // FAST kernel (Mf = 0: not available)
    long ntiles_f;
    int ratio;                              // lines_f / lines_g: GENERAL tiles per queued FAST tile
    size_t lds_g, lds_f;
};

// Tiling of a strided sweep: `lines` adjacent lines x all segments per workgroup.  The pure-streaming rate of this
// access pattern grows with the contiguous bytes per row (measured at 17 B/cell: 16 lines 4.3 TB/s, 32 lines
// 5.1 TB/s), so the FAST kernel (uniform interior, registers only for `in`) takes n/16 rows per thread and 32 lines
// in a 512-thread workgroup; the GENERAL kernel keeps 8 rows per thread (register budget) on tiles of 16 or 32 of
// the same lines, `ratio` of them per FAST tile.
static int strided_m32()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ADI_STRIDED_M32"); v = e ? atoi(e) : 1; }
    return v;
}

static StridedPlan strided_plan(const LineGeom &g, bool want_fast, bool wide_ok, bool fused = false)
{
    StridedPlan P;
    const int n = g.n;
    P.Mg = strided_rows_per_thread(n);
    P.Lpg = next_pow2((n + P.Mg - 1) / P.Mg);
    P.Mf = 0; P.Lpf = 0; P.lines_f = 0; P.tiles_inner_f = 0; P.ntiles_f = 0; P.ratio = 1; P.lds_f = 0;
    int lines = strided_lines_pref();
    if (P.Mg > 8 && lines * P.Lpg > 512) lines = 8;
    const int maxg = (P.Mg <= 8) ? 1024 : 512;
    if (want_fast && n >= 64 && (long)n * g.stride < (1L << 31)) {   // 32-bit element offsets in the FAST kernels
        int mf = 0, lf = 0;
        // 32-line tiles with n/16 rows per thread: only where the kernel keeps few arrays alive (pass A); the solve
        // kernel needs > 200 VGPRs at 32 rows per thread and runs faster on 16-line tiles with 16 rows
        if (wide_ok && n % 16 == 0 && (n / 16 == 8 || n / 16 == 16 || n / 16 == 32)) { mf = n / 16; lf = 32; }   // Lpf = 16
        else if (n > 512 && n % 32 == 0 && n / 32 <= 32 && !fused && strided_m32() && g.stride <= 131072) {
            // long lines, rows less than 1 MiB apart: 32 rows per thread keep 16-line tiles in 512 threads (1024 x 128 x 256:
            // 112 -> 182 Gcell/s); with 2 MiB planes the 8-line tiles of 16 rows are the faster ones (233 vs 212)
            mf = 32; lf = 16;
        } else {
            const int m2 = (n > 256) ? 16 : 8;
            if (n % m2 == 0 && n / m2 <= 64) { mf = m2; lf = (16 * next_pow2(n / m2) > 512) ? 8 : 16; }
        }
        if (mf) {
            const int lpf = next_pow2(n / mf);
            int lg = lf;                                  // GENERAL lines: lf, or lf/2 when the workgroup gets too big
            while (lg * P.Lpg > maxg && lg > 8) lg >>= 1;
            if (lf * lpf <= 512 && lf * lpf >= 256 && lg * P.Lpg <= maxg && lg * P.Lpg >= 64) {
                P.Mf = mf; P.Lpf = lpf; P.lines_f = lf; lines = lg; P.ratio = lf / lg;
            }
        }
    }
    if (P.Mf == 0) while (lines * P.Lpg < 256) lines <<= 1;
    P.lines_g = lines;
    P.tiles_inner_g = (g.n_inner + lines - 1) / lines;
    P.ntiles_g = (long)P.tiles_inner_g * g.n_outer;
    P.lds_g = (size_t)7 * lines * (P.Lpg + 1) * sizeof(double);
    if (P.Mf) {
        P.tiles_inner_f = (g.n_inner + P.lines_f - 1) / P.lines_f;
        P.ntiles_f = (long)P.tiles_inner_f * g.n_outer;
        P.lds_f = (size_t)7 * P.lines_f * (P.Lpf + 1) * sizeof(double);
    }
    return P;
}

// the fused FAST kernels shuffle k-neighbours inside 16-lane DPP rows: 16 lines per tile, at most 16 rows per thread
static bool fuse_fast_ok(const StridedPlan &P, const Lay &L, const Fuse &fz)
{
    return P.Mf != 0 && P.lines_f == 16 && P.Mf <= 16 && L.nz % 16 == 0 && fz.wbytes != 0;
}

static void fuse_tile_order(Fuse &fz, const StridedPlan &P, const Lay &L)
{
    static int kg = -1;
    if (kg < 0) { const char *e = getenv("ADI_FUSE_KG"); kg = e ? atoi(e) : 8; if (kg < 0 || (kg & (kg - 1))) kg = 8; }
    fz.kt = 0; fz.ny = L.ny; fz.kg = 0;
    if (P.Mf != 0 && kg > 0 && L.nz % P.lines_f == 0) {
        const int kt = L.nz / P.lines_f;
        int k2 = kg;
        while (k2 > 1 && kt % k2 != 0) k2 >>= 1;
        fz.kt = kt; fz.kg = k2;          // tiles_inner_f == ny * kt: the remap is a permutation of the tile ids
    }
}

template <int MF, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_strided_fast(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                                const uint8_t *dmask, const double *dval, const double *qf, double *out,
                                const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue,
                                hipStream_t st, const Fuse &fz)
{
    if (s.box && MF <= 16) // all-solid box (caller's hint): the build without surface-segment lanes (fused: no spills)
        hipLaunchKernelGGL((k_sweep_strided_fast<MF, HAS_DIR, HAS_Q, FUSE, (MF > 16)>), dim3((unsigned)P.ntiles_f),
                           dim3(P.lines_f * P.Lpf), P.lds_f, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpf, P.lines_f,
                           P.tiles_inner_f, P.ntiles_f, xlo, xhi, s, queue, make_unic<MF>(s.tg), fz);
    else
        hipLaunchKernelGGL((k_sweep_strided_fast<MF, HAS_DIR, HAS_Q, FUSE, true>), dim3((unsigned)P.ntiles_f),
                           dim3(P.lines_f * P.Lpf), P.lds_f, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpf, P.lines_f,
                           P.tiles_inner_f, P.ntiles_f, xlo, xhi, s, queue, make_unic<MF>(s.tg), fz);
}

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_strided(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                           const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                           const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                           const Fuse &fz)
{
    unsigned ggrid = (unsigned)P.ntiles_g;
    if (queue != nullptr) {
        (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        if (P.Mf == 32) launch_strided_fast<32, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 16) launch_strided_fast<16, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else launch_strided_fast<8, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        ggrid = P.ntiles_g < 1024 ? (unsigned)P.ntiles_g : 1024u;
    }
    hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q, FUSE>), dim3(ggrid), dim3(P.lines_g * P.Lpg), P.lds_g, st, in,
                       flags, coeff, dmask, dval, qf, out, g, P.Lpg, P.lines_g, P.tiles_inner_g, P.ntiles_g, xlo, xhi,
                       s, queue, P.ratio, P.tiles_inner_f, fz);
}

template <int MF, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_condense_fast(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                                 const uint8_t *dmask, const double *dval, const double *qf, double *cond, long nlines,
                                 const LineGeom &g, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    hipLaunchKernelGGL((k_condense_strided_fast<MF, HAS_DIR, HAS_Q, FUSE>), dim3((unsigned)P.ntiles_f),
                       dim3(P.lines_f * P.Lpf), P.lds_f, st, in, flags, coeff, dmask, dval, qf, cond, nlines, g, P.Lpf,
                       P.lines_f, P.tiles_inner_f, P.ntiles_f, s, queue, make_unic<MF>(s.tg), fz);
}

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_condense(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                            const uint8_t *dmask, const double *dval, const double *qf, double *cond, long nlines,
                            const LineGeom &g, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz)
{
    unsigned ggrid = (unsigned)P.ntiles_g;
    if (queue != nullptr) {
        (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        // (the 32-row wide tiling is not built with the fused loader: the host plans fused passes without it)
        if (P.Mf == 32) launch_condense_fast<32, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz);
        else if (P.Mf == 16) launch_condense_fast<16, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz);
        else launch_condense_fast<8, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz);
        ggrid = P.ntiles_g < 1024 ? (unsigned)P.ntiles_g : 1024u;
    }
    hipLaunchKernelGGL((k_condense_strided<M, HAS_DIR, HAS_Q, FUSE>), dim3(ggrid), dim3(P.lines_g * P.Lpg), P.lds_g, st,
                       in, flags, coeff, dmask, dval, qf, cond, nlines, g, P.Lpg, P.lines_g, P.tiles_inner_g, P.ntiles_g,
                       s, queue, P.ratio, P.tiles_inner_f, fz);
}

template <bool HAS_DIR, bool HAS_Q>
static int sweep_dispatch(int axis, const double *in, const uint8_t *flags, const double *coeff, const uint8_t *dmask,
                          const double *dval, const double *qf, const Lay &L, SweepScal s, double *out,
                          const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                          const Fuse *fzp = nullptr)
{
    long inner_stride;
    const LineGeom g = line_geom(axis, L, &inner_stride);
    const int n = g.n;
    if (fzp != nullptr && (axis != 0 || n > kMaxFastLine))
        return set_err(ADI_ERR_UNSUPPORTED, "fused explicit + sweep: axis 0 with at most %d planes only", kMaxFastLine);
    if (n > kMaxFastLine || (axis == 2 && (xlo || xhi))) {
        const size_t need = (size_t)2 * L.nx * L.sx * sizeof(double);
        if (work == nullptr || work_bytes < need)
            return set_err(ADI_ERR_ARG, "adi_sweep: this sweep (line length %d) needs a workspace of %zu bytes", n, need);
        double *wc = (double *)work, *wd = wc + (size_t)L.nx * L.sx;
        const long nl = (long)g.n_inner * g.n_outer;
        hipLaunchKernelGGL((k_sweep_generic<HAS_DIR, HAS_Q>), dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, in,
                           flags, coeff, dmask, dval, qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
        return ADI_OK;
    }
    if (axis == 2) {
        switch (contig_rows_per_lane(n)) {
            case 2: launch_contig<2, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, work, work_bytes, st); break;
            case 4: launch_contig<4, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, work, work_bytes, st); break;
            case 8: launch_contig<8, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, work, work_bytes, st); break;
            default: launch_contig<16, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, work, work_bytes, st); break;
        }
    } else {
        static int wide = -1;
        if (wide < 0) wide = getenv("ADI_STRIDED_WIDE") ? atoi(getenv("ADI_STRIDED_WIDE")) : 0;
        StridedPlan P = strided_plan(g, s.sparse != 0 && work != nullptr, wide != 0 && fzp == nullptr, fzp != nullptr);
        unsigned *queue = nullptr;
        if (P.Mf && use_fast(s, work, work_bytes, P.ntiles_f)) queue = (unsigned *)work;
        else if (P.Mf) P = strided_plan(g, false, false);
        if (fzp != nullptr) {
            Fuse fz = *fzp;
            if (queue != nullptr && !fuse_fast_ok(P, L, fz)) { queue = nullptr; P = strided_plan(g, false, false); }
            fuse_tile_order(fz, P, L);
            switch (P.Mg) {
                case 2: launch_strided<2, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
                case 4: launch_strided<4, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
                case 8: launch_strided<8, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
                default: launch_strided<16, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
            }
            return ADI_OK;
        }
        const Fuse fz = Fuse();
        switch (P.Mg) {
            case 2: launch_strided<2, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 4: launch_strided<4, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 8: launch_strided<8, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
            default: launch_strided<16, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz); break;
        }
    }
    return ADI_OK;
}

// whole-segment condition of the tiled pass-A kernels (also what adi_explicit_fused_supported reports)
static bool condense_is_tiled(int axis, int n)
{
    if (axis == 2 || n > kMaxFastLine) return false;
    const int mg = strided_rows_per_thread(n);
    return (n % mg == 0) && (n / mg <= 64);
}

template <bool HAS_DIR, bool HAS_Q>
static int condense_dispatch(int axis, const double *in, const uint8_t *flags, const double *coeff,
                             const uint8_t *dmask, const double *dval, const double *qf, const Lay &L, SweepScal s,
                             double *cond, void *work, size_t work_bytes, hipStream_t st, const Fuse *fzp = nullptr)
{
    long inner_stride;
    const LineGeom g = line_geom(axis, L, &inner_stride);
    const long nlines = (long)g.n_inner * g.n_outer;
    const int n = g.n;
    bool tiled = false;
    StridedPlan P;
    if (axis != 2 && n <= kMaxFastLine) {
        P = strided_plan(g, s.sparse != 0 && work != nullptr, fzp == nullptr, fzp != nullptr);
        tiled = (n % P.Mg == 0) && (n / P.Mg <= 64);     // the tiled kernels need whole segments
    }
    if (fzp != nullptr && (axis != 0 || !tiled))
        return set_err(ADI_ERR_UNSUPPORTED, "fused explicit + condensation: axis 0, whole segments only (n = %d)", n);
    if (tiled) {
        unsigned *queue = nullptr;
        if (P.Mf && use_fast(s, work, work_bytes, P.ntiles_f)) queue = (unsigned *)work;
        else if (P.Mf) P = strided_plan(g, false, false);
        if (fzp != nullptr) {
            Fuse fz = *fzp;
            if (queue != nullptr && !fuse_fast_ok(P, L, fz)) { queue = nullptr; P = strided_plan(g, false, false); }
            fuse_tile_order(fz, P, L);
            switch (P.Mg) {
                case 2: launch_condense<2, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
                case 4: launch_condense<4, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
                case 8: launch_condense<8, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
                default: launch_condense<16, HAS_DIR, HAS_Q, true>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
            }
            return ADI_OK;
        }
        const Fuse fz = Fuse();
        switch (P.Mg) {
            case 2: launch_condense<2, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
            case 4: launch_condense<4, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
            case 8: launch_condense<8, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
            default: launch_condense<16, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, cond, nlines, g, s, queue, st, fz); break;
        }
    } else {
        hipLaunchKernelGGL((k_condense_generic<HAS_DIR, HAS_Q>), dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0,
                           st, in, flags, coeff, dmask, dval, qf, cond, nlines, g, inner_stride, s);
    }
    return ADI_OK;
}

static int make_lay(int nx, int ny, int nz, long plane_stride, Lay *L)
{
    if (nx <= 0 || ny <= 0 || nz <= 0) return set_err(ADI_ERR_ARG, "bad grid %d x %d x %d", nx, ny, nz);
    const long dense = (long)ny * nz;
    if (plane_stride != 0 && plane_stride < dense)
        return set_err(ADI_ERR_ARG, "plane_stride %ld < ny*nz = %ld", plane_stride, dense);
    if (dense > 0x7fffffffL) return set_err(ADI_ERR_UNSUPPORTED, "plane of %ld cells is too large", dense);
    L->nx = nx; L->ny = ny; L->nz = nz;
    L->sx = plane_stride ? plane_stride : dense;
    return ADI_OK;
}

static unsigned cell_blocks(const Lay &L) { return (unsigned)(((long)L.nx * L.ny * L.nz + 255) / 256); }

}  // namespace adi

using namespace adi;

extern "C" {

long adi_recommended_plane_stride(int ny, int nz)
{
    // planes whose byte size is a multiple of 16 KiB alias on the HBM channel interleave when walked with
    // that stride (axis-0 sweeps): pad by 256 elements (2 KiB; a sweep over 64..4608 showed 256 a few percent ahead on the axis-0 sweep).
    const long dense = (long)ny * nz;
    static long pad = -1;
    if (pad < 0) { const char *e = getenv("ADI_PLANE_PAD"); pad = e ? atol(e) : 256; if (pad < 0 || pad % 8) pad = 256; }
    return (dense * 8 % 16384 == 0) ? dense + pad : dense;
}

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

int adi_build_coeffs(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, double dx, double rho,
                     double cp, const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                     const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                     double *const *d_coeff, double *const *d_qflux, void *stream)
{
    ADI_REQUIRE(d_mask && h_mode && h_scalar && q_mode && q_scalar && d_coeff && d_qflux, "adi_build_coeffs: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
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
    hipLaunchKernelGGL(k_build_coeffs, dim3(cell_blocks(L)), dim3(256), 0, as_stream(stream), d_mask, L, A, Ccell, h,
                       q, d_coeff[0], d_coeff[1], d_coeff[2], d_qflux[0], d_qflux[1], d_qflux[2]);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_build_nbr_flags(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, uint8_t *d_flags,
                        void *stream)
{
    ADI_REQUIRE(d_mask && d_flags, "adi_build_nbr_flags: null argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    hipLaunchKernelGGL(k_build_flags, dim3(cell_blocks(L)), dim3(256), 0, as_stream(stream), d_mask, L, d_flags);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
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
    static int ver = 0;
    if (ver == 0) { const char *e = getenv("ADI_EXPLICIT_VER"); ver = e ? atoi(e) : 5; }
    if (fast && ver >= 4) {
        const int jslab = (ny + 7) / 8;
        const int nslab = (ny + jslab - 1) / jslab;
        const int ktiles = (nz + 511) / 512;
        int ichunk = np >= 512 ? 32 : (np / 16 < 4 ? 4 : np / 16);
        { const char *e = getenv("ADI_EXPLICIT_ICHUNK"); if (e && atoi(e) >= 1) ichunk = atoi(e); }
        const int nchunk = (np + ichunk - 1) / ichunk;
        static int jt5 = 0;
        if (jt5 == 0) { const char *e = getenv("ADI_EXPLICIT_JT"); jt5 = (e && atoi(e) == 4) ? 4 : 2; }
        const long ntiles = (long)nslab * nchunk * ((jslab + jt5 - 1) / jt5) * ktiles;
        if (jt5 == 2)
            hipLaunchKernelGGL(k_explicit_v5<2>, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                               L, invdx2, f, jslab, ktiles, ichunk, ntiles, i_begin, i_end, (const double *)nullptr,
                               (double *)nullptr, 0, 0);
        else
            hipLaunchKernelGGL(k_explicit_v5<4>, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                               L, invdx2, f, jslab, ktiles, ichunk, ntiles, i_begin, i_end, (const double *)nullptr,
                               (double *)nullptr, 0, 0);
    } else if (fast && (ver == 3 || np != nx)) {
        const int jslab = (ny + 7) / 8;
        const int nslab = (ny + jslab - 1) / jslab;
        const int ktiles = (nz + 511) / 512;
        // planes marched per block: 32 amortises the leading halo plane; short plane ranges (the boundary windows of a
        // slab) get shorter chunks so that the launch still has ~16 chunks' worth of blocks
        int ichunk = np >= 512 ? 32 : (np / 16 < 4 ? 4 : np / 16);
        { const char *e = getenv("ADI_EXPLICIT_ICHUNK"); if (e && atoi(e) >= 1) ichunk = atoi(e); }
        const int nchunk = (np + ichunk - 1) / ichunk;
        static int jt = 0;
        if (jt == 0) { const char *e = getenv("ADI_EXPLICIT_JT"); jt = (e && atoi(e) == 8) ? 8 : 4; }
        const long ntiles = (long)nslab * nchunk * ((jslab + jt - 1) / jt) * ktiles;
        if (jt == 8)
            hipLaunchKernelGGL(k_explicit_v3<8>, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                               L, invdx2, f, jslab, ktiles, ichunk, ntiles, i_begin, i_end);
        else
            hipLaunchKernelGGL(k_explicit_v3<4>, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0,
                               L, invdx2, f, jslab, ktiles, ichunk, ntiles, i_begin, i_end);
    } else if (fast) {
        const int jslab = (ny + 7) / 8;
        const int nslab = (ny + jslab - 1) / jslab;
        const int ktiles = (nz + 511) / 512;
        const int kExplicitJR = explicit_jr();
        const long ntiles = (long)nslab * nx * ((jslab + kExplicitJR - 1) / kExplicitJR) * ktiles;
        hipLaunchKernelGGL(k_explicit_v2, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), d_T, d_flags, d_R0, L,
                           invdx2, f, jslab, ktiles, ntiles, kExplicitJR);
    } else {
        const long cells = (long)np * ny * nz;
        hipLaunchKernelGGL(k_explicit, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, as_stream(stream), d_T,
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

static int variant_flags(int variant, bool *has_dir, bool *has_q);

// ---- pass A folded into the explicit stage (slab decomposition) ------------------------------------------------------
static int dots_ichunk(int np) { return np >= 512 ? 32 : (np / 16 < 4 ? 4 : np / 16); }

int adi_axis0_dots_supported(int nx, int ny, int nz, long plane_stride)
{
    static int off = -1;
    if (off < 0) off = getenv("ADI_NO_DOTS") ? 1 : 0;
    Lay L;
    if (off || make_lay(nx, ny, nz, plane_stride, &L) != ADI_OK) return 0;
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
                       d_weights, d_part, i_org, n_line);
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
    // the lines that are not uniform: the serial two-recurrence condensation from the stored R0 (grid sized for all
    // lines; the kernel returns beyond the list's count)
    long inner_stride;
    const LineGeom g = line_geom(0, L, &inner_stride);
    const unsigned gl = (unsigned)((nlines + 255) / 256);
    if (has_dir && has_q) hipLaunchKernelGGL((k_condense_generic<true, true>), dim3(gl), dim3(256), 0, st, d_R0, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, d_cond, nlines, g, inner_stride, s, d_list, line_begin, nsel);
    else if (has_q) hipLaunchKernelGGL((k_condense_generic<false, true>), dim3(gl), dim3(256), 0, st, d_R0, d_flags, d_coeff, nullptr, nullptr, d_qflux, d_cond, nlines, g, inner_stride, s, d_list, line_begin, nsel);
    else if (has_dir) hipLaunchKernelGGL((k_condense_generic<true, false>), dim3(gl), dim3(256), 0, st, d_R0, d_flags, d_coeff, d_dir_mask, d_dir_val, nullptr, d_cond, nlines, g, inner_stride, s, d_list, line_begin, nsel);
    else hipLaunchKernelGGL((k_condense_generic<false, false>), dim3(gl), dim3(256), 0, st, d_R0, d_flags, d_coeff, nullptr, nullptr, nullptr, d_cond, nlines, g, inner_stride, s, d_list, line_begin, nsel);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_sweep_workspace_bytes(int axis, int nx, int ny, int nz, long plane_stride, size_t *bytes)
{
    ADI_REQUIRE(axis >= 0 && axis < 3 && bytes, "adi_sweep_workspace_bytes: bad argument");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    const int nn[3] = {nx, ny, nz};
    // long lines: c', d' scratch; otherwise the unit queue of the FAST/GENERAL kernel pair (one id per tile or
    // line group; nx*ny*nz/8 + 1 ids is an upper bound for every tiling this library uses)
    const size_t cells = (size_t)nx * ny * nz;
    *bytes = nn[axis] > kMaxFastLine ? (size_t)2 * nx * L.sx * sizeof(double) : (cells / 8 + 64) * sizeof(unsigned);
    return ADI_OK;
}

static int variant_flags(int variant, bool *has_dir, bool *has_q)
{
    if (variant < 0 || variant > 3) return set_err(ADI_ERR_ARG, "bad sweep variant %d", variant);
    *has_dir = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_Q);
    *has_q = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_DIR);
    return ADI_OK;
}

static int sweep_entry(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                       const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
                       long plane_stride, int sparse, double theta, double gam, double dt, double Tinf, double *d_out,
                       const double *d_xlo, const double *d_xhi, void *d_work, size_t work_bytes, void *stream,
                       const Fuse *fz)
{
    ADI_REQUIRE(axis >= 0 && axis < 3, "adi_sweep: bad axis %d", axis);
    bool has_dir, has_q;
    if (int rc = variant_flags(variant, &has_dir, &has_q)) return rc;
    ADI_REQUIRE(d_in && d_flags && d_coeff && d_out, "adi_sweep: null argument");
    ADI_REQUIRE(d_in != d_out, "adi_sweep: output aliases input");
    ADI_REQUIRE(!has_dir || (d_dir_mask && d_dir_val), "adi_sweep: variant needs Dirichlet arrays");
    ADI_REQUIRE(!has_q || d_qflux, "adi_sweep: variant needs the flux array");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    SweepScal s;
    s.tg = theta * gam;
    s.dt = dt;
    s.Tinf = Tinf;
    s.sparse = (sparse & 1) ? 1 : 0;
    s.box = (sparse & 2) ? 1 : 0;
    hipStream_t st = as_stream(stream);
    int rc;
    if (has_dir && has_q) rc = sweep_dispatch<true, true>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, L, s, d_out, d_xlo, d_xhi, d_work, work_bytes, st, fz);
    else if (has_q) rc = sweep_dispatch<false, true>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, d_qflux, L, s, d_out, d_xlo, d_xhi, d_work, work_bytes, st, fz);
    else if (has_dir) rc = sweep_dispatch<true, false>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, nullptr, L, s, d_out, d_xlo, d_xhi, d_work, work_bytes, st, fz);
    else rc = sweep_dispatch<false, false>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, nullptr, L, s, d_out, d_xlo, d_xhi, d_work, work_bytes, st, fz);
    if (rc != ADI_OK) return rc;
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_sweep(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
              const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
              long plane_stride, int sparse, double theta, double gam, double dt, double Tinf, double *d_out,
              const double *d_xlo, const double *d_xhi, void *d_work, size_t work_bytes, void *stream)
{
    return sweep_entry(axis, variant, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                       sparse, theta, gam, dt, Tinf, d_out, d_xlo, d_xhi, d_work, work_bytes, stream, nullptr);
}

static Fuse make_fuse(int nx, int ny, int nz, long plane_stride, double dx, double dt, double kappa, double theta,
                      long valid_lo, long valid_hi)
{
    Fuse fz;
    fz.invdx2 = 1.0 / (dx * dx);                 // the same two expressions as adi_explicit_rhs_planes
    fz.f = dt * kappa * (1.0 - theta);
    fz.sy = nz;
    fz.vlo = valid_lo;
    fz.vhi = valid_hi;
    fz.kt = 0; fz.ny = 0; fz.kg = 0;
    fz.r0_out = nullptr;
    // window of the state the FAST kernel's buffer descriptor covers: the box, one plane + one row + one tile around it
    const long sx = plane_stride ? plane_stride : (long)ny * nz;
    const long box_end = (long)(nx - 1) * sx + (long)ny * nz;
    const long wlo = valid_lo > -(sx + nz + 16) ? valid_lo : -(sx + nz + 16);
    const long whi = valid_hi < box_end + sx + nz + 32 ? valid_hi : box_end + sx + nz + 32;
    fz.wlo = wlo;
    fz.wbytes = ((whi - wlo) * 8 < 0x7ffff000L) ? (unsigned)((whi - wlo) * 8) : 0u;
    return fz;
}

int adi_explicit_fused_supported(int nx, int ny, int nz, long plane_stride, int pass)
{
    static int off = -1;
    if (off < 0) off = getenv("ADI_NO_FUSE") ? 1 : 0;
    if (off || nx <= 0 || ny <= 0 || nz <= 0) return 0;
    // the FAST fused kernel addresses the state through one buffer descriptor (box + a plane either side): beyond
    // 2 GiB only the GENERAL fused kernel could run, and the separate explicit stage + sweep are faster than that
    const long sx = plane_stride ? plane_stride : (long)ny * nz;
    if (((long)nx + 2) * sx * 8 + ((long)nz + 64) * 16 >= 0x7ffff000L) return 0;
    // lines the FAST fused kernel cannot tile (16-line tiles of at most 32 segments of 8 / 16 rows, nz a multiple of 16)
    // would all run through the GENERAL fused kernel, which is slower than the separate explicit stage + sweep
    // (640 x 512 x 512: 5.1 ms fused-GENERAL against 2.4 ms): decline, except for short lines where nothing is tiled anyway
    if (nx >= 64) {
        const bool fast_ok = (nz % 16 == 0) && (nx <= 256 ? (nx % 8 == 0) : (nx % 16 == 0 && nx / 16 <= 32));
        if (!fast_ok) return 0;
    }
    return pass == 0 ? (nx <= kMaxFastLine) : (condense_is_tiled(0, nx) ? 1 : 0);
}

int adi_explicit_sweep0(int variant, const double *d_T, long valid_lo, long valid_hi, const uint8_t *d_flags,
                        const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                        const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse, double dx,
                        double dt, double kappa, double theta, double Tinf, double *d_out, const double *d_xlo,
                        const double *d_xhi, void *d_work, size_t work_bytes, void *stream)
{
    ADI_REQUIRE(valid_lo <= 0 && valid_hi >= (long)(nx - 1) * (plane_stride ? plane_stride : (long)ny * nz) + (long)ny * nz,
                "adi_explicit_sweep0: the readable range [%ld, %ld) does not cover the box", valid_lo, valid_hi);
    const Fuse fz = make_fuse(nx, ny, nz, plane_stride, dx, dt, kappa, theta, valid_lo, valid_hi);
    const double gam = kappa * dt / (dx * dx);   // adi3d_numba_coeff.py:292
    return sweep_entry(0, variant, d_T, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                       sparse, theta, gam, dt, Tinf, d_out, d_xlo, d_xhi, d_work, work_bytes, stream, &fz);
}

static int condense_entry(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                          const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny,
                          int nz, long plane_stride, int sparse, double theta, double gam, double dt, double Tinf,
                          double *d_cond, void *d_work, size_t work_bytes, void *stream, const Fuse *fz)
{
    ADI_REQUIRE(axis >= 0 && axis < 3, "adi_sweep_condense: bad axis %d", axis);
    bool has_dir, has_q;
    if (int rc = variant_flags(variant, &has_dir, &has_q)) return rc;
    ADI_REQUIRE(d_in && d_flags && d_coeff && d_cond, "adi_sweep_condense: null argument");
    ADI_REQUIRE(!has_dir || (d_dir_mask && d_dir_val), "adi_sweep_condense: variant needs Dirichlet arrays");
    ADI_REQUIRE(!has_q || d_qflux, "adi_sweep_condense: variant needs the flux array");
    Lay L;
    if (int rc = make_lay(nx, ny, nz, plane_stride, &L)) return rc;
    SweepScal s;
    s.tg = theta * gam;
    s.dt = dt;
    s.Tinf = Tinf;
    s.sparse = (sparse & 1) ? 1 : 0;
    s.box = (sparse & 2) ? 1 : 0;
    hipStream_t st = as_stream(stream);
    int rc;
    if (has_dir && has_q) rc = condense_dispatch<true, true>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, L, s, d_cond, d_work, work_bytes, st, fz);
    else if (has_q) rc = condense_dispatch<false, true>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, d_qflux, L, s, d_cond, d_work, work_bytes, st, fz);
    else if (has_dir) rc = condense_dispatch<true, false>(axis, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, nullptr, L, s, d_cond, d_work, work_bytes, st, fz);
    else rc = condense_dispatch<false, false>(axis, d_in, d_flags, d_coeff, nullptr, nullptr, nullptr, L, s, d_cond, d_work, work_bytes, st, fz);
    if (rc != ADI_OK) return rc;
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_sweep_condense(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                       const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny,
                       int nz, long plane_stride, int sparse, double theta, double gam, double dt, double Tinf,
                       double *d_cond, void *d_work, size_t work_bytes, void *stream)
{
    return condense_entry(axis, variant, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                          sparse, theta, gam, dt, Tinf, d_cond, d_work, work_bytes, stream, nullptr);
}

int adi_explicit_condense0(int variant, const double *d_T, long valid_lo, long valid_hi, const uint8_t *d_flags,
                           const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                           const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse, double dx,
                           double dt, double kappa, double theta, double Tinf, double *d_cond, double *d_R0_out,
                           void *d_work, size_t work_bytes, void *stream)
{
    ADI_REQUIRE(valid_lo <= 0 && valid_hi >= (long)(nx - 1) * (plane_stride ? plane_stride : (long)ny * nz) + (long)ny * nz,
                "adi_explicit_condense0: the readable range [%ld, %ld) does not cover the box", valid_lo, valid_hi);
    ADI_REQUIRE(d_R0_out != d_T, "adi_explicit_condense0: R0 output aliases the state");
    Fuse fz = make_fuse(nx, ny, nz, plane_stride, dx, dt, kappa, theta, valid_lo, valid_hi);
    fz.r0_out = d_R0_out;
    const double gam = kappa * dt / (dx * dx);
    return condense_entry(0, variant, d_T, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                          sparse, theta, gam, dt, Tinf, d_cond, d_work, work_bytes, stream, &fz);
}

int adi_interface_solve(const double *d_cond_all, int nranks, int rank, long nlines, double *d_xlo, double *d_xhi,
                        void *stream)
{
    ADI_REQUIRE(d_cond_all && d_xlo && d_xhi && nranks >= 1 && rank >= 0 && rank < nranks && nlines > 0,
                "adi_interface_solve: bad argument");
    hipLaunchKernelGGL(k_interface, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream), d_cond_all,
                       nranks, rank, nlines, d_xlo, d_xhi);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_interface_pair(const double *d_my_lo, const double *d_my_hi, const double *d_prev_hi, const double *d_next_lo,
                       long nlines, double *d_xlo, double *d_xhi, void *stream)
{
    ADI_REQUIRE(d_xlo && d_xhi && nlines > 0, "adi_interface_pair: bad argument");
    ADI_REQUIRE((!d_prev_hi || d_my_lo) && (!d_next_lo || d_my_hi), "adi_interface_pair: missing own window");
    hipLaunchKernelGGL(k_interface_pair, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream), d_my_lo,
                       d_my_hi, d_prev_hi, d_next_lo, nlines, d_xlo, d_xhi);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_step(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b, const uint8_t *d_flags,
             const double *const *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
             const double *const *d_qflux, int variant, int sparse, int nx, int ny, int nz, long plane_stride,
             double dx, double rho, double cp, double k, double dt, double theta, double Tinf, void *d_work,
             size_t work_bytes, void *stream)
{
    ADI_REQUIRE(d_T_in && d_T_out && d_tmp_a && d_tmp_b && d_coeff, "adi_step: null argument");
    ADI_REQUIRE(d_tmp_a != d_tmp_b && d_tmp_a != d_T_in && d_tmp_b != d_T_in && d_T_out != d_tmp_a && d_T_out != d_T_in,
                "adi_step: buffers must be distinct (T_out may equal tmp_b only)");
    // kappa, gam: adi3d_numba_coeff.py:292
    const double kappa = k / (rho * cp);
    const double gam = kappa * dt / (dx * dx);
    const double *q0 = d_qflux ? d_qflux[0] : nullptr, *q1 = d_qflux ? d_qflux[1] : nullptr, *q2 = d_qflux ? d_qflux[2] : nullptr;
    int rc;
    if (adi_explicit_fused_supported(nx, ny, nz, plane_stride, 0)) {
        // stages 1+2 in one pass: R0 is evaluated inside the loads of the axis-0 sweep
        const long sxe = plane_stride ? plane_stride : (long)ny * nz;
        rc = adi_explicit_sweep0(variant, d_T_in, 0, (long)(nx - 1) * sxe + (long)ny * nz, d_flags, d_coeff[0], d_dir_mask,
                                 d_dir_val, q0, nx, ny, nz, plane_stride, sparse, dx, dt, kappa, theta, Tinf, d_tmp_b,
                                 nullptr, nullptr, d_work, work_bytes, stream);
    } else {
        rc = adi_explicit_rhs(d_T_in, d_flags, nx, ny, nz, plane_stride, dx, dt, kappa, theta, d_tmp_a, stream);
        if (rc) return rc;
        rc = adi_sweep(0, variant, d_tmp_a, d_flags, d_coeff[0], d_dir_mask, d_dir_val, q0, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_tmp_b, nullptr, nullptr, d_work, work_bytes, stream);
    }
    if (rc) return rc;
    rc = adi_sweep(1, variant, d_tmp_b, d_flags, d_coeff[1], d_dir_mask, d_dir_val, q1, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_tmp_a, nullptr, nullptr, d_work, work_bytes, stream);
    if (rc) return rc;
    return adi_sweep(2, variant, d_tmp_a, d_flags, d_coeff[2], d_dir_mask, d_dir_val, q2, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_T_out, nullptr, nullptr, d_work, work_bytes, stream);
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
