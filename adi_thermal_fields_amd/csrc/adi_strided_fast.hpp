// adi_strided_fast.hpp -- the FAST kernel of the strided-axis sweeps and its launcher, shared by adi_sweep_strided.hip (8 / 16 /
// 32 rows per thread, and the form with the explicit stage folded in) and adi_sweep_strided_x.hip (20 / 24 / 28 rows per thread
// for lines those row counts cut into 16 or 32 segments).
#pragma once
#include "adi_cart_host.hpp"
#include "adi_strided_dev.hpp"

namespace adi {

// FC: the pack was built from per-face scalars (SweepScal::fconst): coefficient / flux of an exposed row from its flags,
// no load path in the kernel at all (instantiated in adi_sweep_strided_fc.hip)
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool MIXED = true, bool FC = false>
__global__ __launch_bounds__(512, FUSE ? ADI_FUSE_OCC : 1) void k_sweep_strided_fast(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ xlo, const double *__restrict__ xhi, SweepScal s, unsigned *__restrict__ queue,
    UniC<M> U, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    if constexpr (!FUSE && ADI_LOAD_PRIO != 0) __builtin_amdgcn_s_setprio(ADI_LOAD_PRIO);   // loading waves first (adi_cart_dev.hpp)
    long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    if (FUSE && fz.kg > 0) tile = tile_jfast(tile, fz);
    if constexpr (!FUSE) {
        // deferred interface correction (SweepScal::c_*): every plane of the slab re-reads the same two correction planes.
        // Give each XCD a fixed eighth of the k-tiles over ALL planes instead of a chunk of whole planes: its share of the
        // two planes (2 x ny x nz/8 values: 512 KiB at 512^2) then stays in its L2 for the whole launch.
        if (s.c_w != nullptr && (tiles_inner & 7) == 0) {
            const unsigned kx = (unsigned)tiles_inner >> 3, x = blockIdx.x & 7u, idx = blockIdx.x >> 3;
            const unsigned pl = idx / kx;
            tile = (long)pl * tiles_inner + (long)(x * kx + (idx - pl * kx));
        }
    }
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
    // flag bytes through packed 16-byte loads (load_bytes_packed16): 16-line tiles whose byte rows are 16-byte aligned
    uint8_t *strip = nullptr;
    if (M % 4 == 0 && M <= 16 && LINES == 16 && (g.stride & 15) == 0 && (tbase & 15) == 0 && ((uintptr_t)flags & 15) == 0)
        strip = reinterpret_cast<uint8_t *>(sm + 7 * LINES * (Lp + 1)) + (size_t)sg * (LINES * M);
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
                                                               kind, Lm, strip) || pad;
        if (pad) kind = SEG_PAD;
    } else {
        // whole tiles whose rows fit 31-bit byte offsets take the buffer-addressed loader (block-uniform choice)
        const bool whole = kBufStrided && (ti + 1) * LINES <= g.n_inner && Lp * M == g.n &&
                           (long)g.n * g.stride * 8 < 0x7fffffffL;
        if (whole)
            lane_fast = fast_segment_load_buf<M, HAS_DIR>(in + tbase, flags + tbase, HAS_DIR ? dmask + tbase : dmask, g, voff, d,
                                                          f0, fS, dirS, kind, Lm, strip);
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
    if constexpr (!FUSE) {
        // deferred interface correction of a slab decomposition (SweepScal::c_*): block-uniform, off in ordinary sweeps
        if (s.c_w != nullptr)
            corr_apply<M>(s, corr_weights(s, to), to, (voff + (unsigned)(ti * LINES)) * 8u, (unsigned)(g.stride * 8), d,
                          Lp * M == g.n && (ti + 1) * LINES <= g.n_inner);
    }
    // back to normal priority for the solver phase -- AFTER the correction loads of a slab run (they are loads too: the
    // corrected axis-1 sweep of the weak rehearsal 0.528 -> 0.490 ms with them inside the raised window)
    if constexpr (!FUSE && ADI_LOAD_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    double a0, b0, aS, bS, cS;
    fast_segment_ends<M, HAS_DIR, HAS_Q, (FC ? 1 : 2)>(coeff, dval, qf, g, base, r0, f0, fS, dirS, s, d, a0, b0, aS, bS, cS);
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
            mixed_lane_condense<M, HAS_Q, (FC ? 1 : 2)>(kind, Lm, U, s, coeff + base + (long)r0 * g.stride,
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
        for (int r = 0; r < M; ++r) buf_store_f64(rO, voff * 8u, (unsigned)r * (unsigned)(g.stride * 8), d[r], s.nt != 0);
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) (out_t + (size_t)r * g.stride)[voff] = d[r];
    }
}

template <int MF, bool HAS_DIR, bool HAS_Q, bool FUSE, bool FC>
inline void launch_strided_fast_t(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                                  const uint8_t *dmask, const double *dval, const double *qf, double *out,
                                  const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue,
                                  hipStream_t st, const Fuse &fz)
{
    if (s.box && MF <= 16) // all-solid box (caller's hint): the build without surface-segment lanes (fused: no spills)
        hipLaunchKernelGGL((k_sweep_strided_fast<MF, HAS_DIR, HAS_Q, FUSE, (MF > 16), FC>), dim3((unsigned)P.ntiles_f),
                           dim3(P.lines_f * P.Lpf), P.lds_f, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpf, P.lines_f,
                           P.tiles_inner_f, P.ntiles_f, xlo, xhi, s, queue, make_unic<MF>(s.tg), fz);
    else
        hipLaunchKernelGGL((k_sweep_strided_fast<MF, HAS_DIR, HAS_Q, FUSE, true, FC>), dim3((unsigned)P.ntiles_f),
                           dim3(P.lines_f * P.Lpf), P.lds_f, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpf, P.lines_f,
                           P.tiles_inner_f, P.ntiles_f, xlo, xhi, s, queue, make_unic<MF>(s.tg), fz);
}

// adi_sweep_strided_fc.hip: the FC = true instantiations (8 / 16 / 32 rows per thread, no Dirichlet cells, fused or not)
void strided_fast_fc(int mf, bool has_q, bool fuse, const StridedPlan &P, const double *in, const uint8_t *flags,
                     const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                     const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz);

template <int MF, bool HAS_DIR, bool HAS_Q, bool FUSE>
inline void launch_strided_fast(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                                const uint8_t *dmask, const double *dval, const double *qf, double *out,
                                const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue,
                                hipStream_t st, const Fuse &fz)
{
    if (!HAS_DIR && s.fconst && (MF == 8 || MF == 16 || MF == 32))
        strided_fast_fc(MF, HAS_Q, FUSE, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
    else
        launch_strided_fast_t<MF, HAS_DIR, HAS_Q, FUSE, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue,
                                                               st, fz);
}

// adi_sweep_strided_fx.hip: the fused kernel with 10 / 12 / 14 rows per thread (exact fits, no Dirichlet cells)
void strided_fast_fused_exact(int mf, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                              const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                              const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz);
void strided_fast_fused_exact_odd(int mf, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                              const double *coeff, const double *qf, double *out, const LineGeom &g, const double *xlo,
                              const double *xhi, SweepScal s, unsigned *queue, hipStream_t st, const Fuse &fz);   // 9 / 11 / 13 / 15 rows (adi_sweep_strided_fy.hip)

// adi_sweep_strided_x.hip: launch_strided_fast<mf, ..., false> for mf = 20, 24, 28
void strided_fast_exact(int mf, bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                        const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                        const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                        const Fuse &fz);
void strided_fast_exact2(int mf, bool has_dir, bool has_q, const StridedPlan &P, const double *in, const uint8_t *flags,
                        const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                        const LineGeom &g, const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                        const Fuse &fz);   // 18 / 22 / 26 / 30 rows (adi_sweep_strided_y.hip)

}  // namespace adi
