// adi_contig_dev.hpp -- device code shared by the translation units of the contiguous-axis sweep (adi_sweep_contig.hip: the
// GENERAL kernel and the FAST kernel with 2 / 4 / 8 / 16 rows per lane; adi_sweep_contig_x.hip: the FAST kernel with 20 / 24 /
// 28 rows per lane for lines those row counts cut into a power-of-two number of segments).
#pragma once
#include "adi_cart_host.hpp"

namespace adi {

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
            } else if (M % 8 == 0) {
#pragma unroll
                for (int h = 0; h < M / 8; ++h) {
                    const uint64_t w = *reinterpret_cast<const uint64_t *>(p + base + 8 * h);
#pragma unroll
                    for (int r = 0; r < 8; ++r) b[8 * h + r] = (unsigned)((w >> (8 * r)) & 0xffull);
                }
            } else if (M % 4 == 0) {                   // 20, 28 rows per lane: chunks start on 4-byte boundaries
#pragma unroll
                for (int h = 0; h < M / 4; ++h) {
                    const unsigned w = *reinterpret_cast<const uint32_t *>(p + base + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[4 * h + r] = (w >> (8 * r)) & 0xffu;
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r)
            if (active && r0 + r < n) b[r] = p[base + r];
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
    __shared__ __align__(16) double strips[MODE >= 2 ? 4 * 32 * (M + 2) : 2];
    double *strip = strips + (MODE >= 2 ? wave * 32 * (M + 2) : 0);
    // MODE 2: the 64/Lp lines of a unit are consecutive in memory (host checks ny % lw == 0), so the wave's 64*M
    // doubles start at the base of lane 0
    const long wbase = __shfl(base, 0);
    double d[M];
    unsigned fb[M], db[M];
    load_bytes_contig<M, VEC>(flags, base, r0, n, active, fb);
    if (HAS_DIR) load_bytes_contig<M, VEC>(dmask, base, r0, n, active, db);
    if constexpr (MODE == 2) coal_load<M>(in + wbase, strip, lane, d);
    else if constexpr (MODE == 3) {
        coal_load_r<M>(in + wbase, strip, lane, d, n, Lp);          // segment count not a power of two: lanes beyond it are padding
        if (r0 >= n) {
#pragma unroll
            for (int r = 0; r < M; ++r) d[r] = 0.0;
        }
    } else load_rows_contig<M, VEC>(in, base, r0, n, active, d);
    // the two ends of a line are always exposed: fetch their coefficient / flux with the first batch of loads
    const bool sp0 = active && li == 0, spS = active && (r0 + M == n);
    // (per-face scalars, SweepScal::fconst: nothing is loaded, the values follow from the flags below)
    const bool ldc = !s.fconst;
    const double co0s = (ldc && sp0) ? coeff[base] : 0.0, coSs = (ldc && spS) ? coeff[base + M - 1] : 0.0;
    double q0s = 0.0, qSs = 0.0;
    if (HAS_Q) { q0s = (ldc && sp0) ? qf[base] : 0.0; qSs = (ldc && spS) ? qf[base + M - 1] : 0.0; }
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
    const bool l0 = (fb[0] >> 5) & 1u, h0 = (fb[0] >> 6) & 1u, lS = (fb[M - 1] >> 5) & 1u, hS = (fb[M - 1] >> 6) & 1u;
    const double co0 = e0 ? ((ldc && sp0) ? co0s : pack_co(s, coeff + base, l0, h0)) : 0.0;
    const double coS = eS ? ((ldc && spS) ? coSs : pack_co(s, coeff + base + M - 1, lS, hS)) : 0.0;
    double q0 = 0.0, qS = 0.0, dvS = 0.0;
    if (HAS_Q) {
        q0 = e0 ? ((ldc && sp0) ? q0s : pack_q<HAS_Q>(s, qf + base, l0, h0)) : 0.0;
        qS = eS ? ((ldc && spS) ? qSs : pack_q<HAS_Q>(s, qf + base + M - 1, lS, hS)) : 0.0;
    }
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
        coal_store<M>(out + wbase, strip, lane, d, s.nt != 0);      // (the host takes this mode only when no lane is padding)
    } else if constexpr (MODE == 3) {
        coal_store_r<M>(out + wbase, strip, lane, d, n, Lp, !pad, s.nt != 0);
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


template <int MF, bool HAS_DIR, bool HAS_Q>
inline void launch_contig_fast(const double *in, const uint8_t *flags, const double *coeff, const uint8_t *dmask,
                               const double *dval, const double *qf, double *out, const Lay &L, SweepScal s, bool vec,
                               long nunits_f, unsigned *queue, hipStream_t st)
{
    const int Lpf = next_pow2(L.nz / MF);
    const unsigned grid = (unsigned)((nunits_f + 3) / 4);
    const UniC<MF> U = make_unic<MF>(s.tg);
    const int lwf = 64 / Lpf;
    // coalesced + LDS-transposed access: whole units of contiguous lines (full last unit, no plane straddling)
    const bool coal = vec && MF >= 4 && Lpf * MF == L.nz && (L.ny % lwf == 0) &&
                      (((long)L.nx * L.ny) % lwf == 0);
    // ... and the same access for segment counts that are not a power of two (16 rows per lane: 272 ... 496-row lines, the
    // lengths padded extents often end on): whole units of whole lines, lanes beyond the last segment are padding
    // Only with the caller's all-solid hint (SweepScal::box): where every line ends in a surface segment -- a padded nz -- the
    // kernel is bound by the serial work of those lanes, not by its loads, and the extra index arithmetic of this mode costs
    // more than the coalescing gains (256 x 256 x 257 on a 272-row box: 170 Gcell/s with lane-owned chunks, 148 with this mode;
    // solid 256 x 256 x 496: 227 -> 283).
    const bool coal_r = s.box != 0 && vec && MF == 16 && L.nz % MF == 0 && Lpf * MF != L.nz && (L.ny % lwf == 0) &&
                        (((long)L.nx * L.ny) % lwf == 0);
    if (coal)
        hipLaunchKernelGGL((k_sweep_contig_fast<(MF >= 4 ? MF : 4), 2, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, make_unic<(MF >= 4 ? MF : 4)>(s.tg));
    else if (coal_r) {
        if constexpr (MF == 16)
            hipLaunchKernelGGL((k_sweep_contig_fast<16, 3, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags, coeff,
                               dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, U);
    } else if (vec)
        hipLaunchKernelGGL((k_sweep_contig_fast<MF, 1, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags, coeff,
                           dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, U);
    else
        hipLaunchKernelGGL((k_sweep_contig_fast<MF, 0, HAS_DIR, HAS_Q>), dim3(grid), dim3(256), 0, st, in, flags,
                           coeff, dmask, dval, qf, out, L, Lpf, s, nunits_f, queue, U);
}

// adi_sweep_contig_x.hip: launch_contig_fast<mf, ...> for mf = 20, 24, 28
void contig_fast_exact(int mf, bool has_dir, bool has_q, const double *in, const uint8_t *flags, const double *coeff,
                       const uint8_t *dmask, const double *dval, const double *qf, double *out, const Lay &L, SweepScal s,
                       bool vec, long nunits_f, unsigned *queue, hipStream_t st);

}  // namespace adi
