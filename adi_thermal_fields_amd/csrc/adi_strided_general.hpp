// adi_strided_general.hpp -- the GENERAL kernel of the strided-axis sweeps (memory axes 0 and 1) and its launcher, shared by
// adi_sweep_strided.hip (packs read from their arrays), adi_sweep_strided_gc.hip (packs built from per-face scalars: the
// coefficients follow from the flags, SweepScal::fconst) and adi_sweep_strided_gk.hip (the deferred interface correction of a
// slab decomposition, adi_sweep_corrected).  Both choices are TEMPLATE parameters of the 8- and 16-row kernels: this kernel
// sits at the 128-VGPR budget of two 512-thread workgroups per CU, and behind a run-time branch the mere presence of the
// other path sets the register count of both (round 3 shipped them as run-time branches: 16 -> 56 B of scratch in the 42 B/cell
// kernel, 1.200 -> 1.261 ms for the axis-0 sweep of 512^3; tests/test_kernel_footprint.py now reads the scratch of every
// kernel out of the code objects).
#pragma once
#include "adi_strided_fast.hpp"

namespace adi {

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
// FCM: where the Robin coefficient / Neumann flux of an exposed row comes from -- 0: decided at run time (s.fconst), 1: the
// flags (no load path at all), 2: the pack arrays.  CORR: the kernel may be asked for the deferred interface correction
// (s.c_w, checked at run time); false: the code is not there.  Rows per thread below 8 keep <0, true>: short lines, no
// register pressure.
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool WHOLE = false, int FCM = 0, bool CORR = true>
__device__ __forceinline__ void strided_tile_general(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, const LineGeom &g, int Lp, int LINES, int tiles_inner, long tile,
    const double *__restrict__ xlo, const double *__restrict__ xhi, const SweepScal &s, double *sm,
    const Fuse &fz, const int tid)
{
    const long to = (long)((unsigned)tile / (unsigned)tiles_inner);   // block-uniform, < 2^31 tiles
    const int ti = (int)(tile - to * tiles_inner);
    const int kk = tid & (LINES - 1), sg = tid >> (__ffs(LINES) - 1);   // LINES is a power of two
    const int kcol = ti * LINES + kk;
    const bool active = kcol < g.n_inner;
    const long base = to * g.outer_stride + kcol;
    const int r0 = sg * M;
    const long line_id = to * (long)g.n_inner + kcol;

    // WHOLE (host: every tile of the launch is whole, rows within 31-bit byte offsets of the tile base): the buffer-addressed
    // loader and stores.  A separate instantiation, not a run-time branch: the flat-addressed path's per-row pointers would
    // set the register count of both (148 VGPRs against 128: scratch)
    constexpr bool whole = WHOLE && !FUSE;
    const long tbase = to * g.outer_stride + (long)ti * LINES;
    const unsigned voff = (unsigned)((long)r0 * g.stride + kk);
    // off-diagonals as one bit per row (MaskedCoef): 4*M VGPRs less than two arrays of doubles, which is what keeps this
    // kernel's 8 rows x (b, d, pivots + the raw rows still in flight) inside 128 VGPRs without scratch
    MaskedCoef a, c;
    a.m = 0; c.m = 0; a.v = -s.tg; c.v = -s.tg;
    double b[M], d[M];
    {
        SegRaw<M> R;
        // packed byte loads (see load_segment_raw): whole 8-line tiles of 8-row segments, 8-byte aligned byte rows
        uint8_t *bstrip = nullptr;
        if (M == 8 && LINES == 8 && !FUSE && (ti + 1) * LINES <= g.n_inner && Lp * M == g.n && (g.stride & 7) == 0 &&
            ((to * g.outer_stride) & 7) == 0 && (((uintptr_t)flags | (uintptr_t)(HAS_DIR ? dmask : flags)) & 7) == 0)
            bstrip = reinterpret_cast<uint8_t *>(sm + 7 * LINES * (Lp + 1));
        if (whole) {
            // (16-line tiles of 8-row segments: the packed 16-byte form; rows and tile base 16-byte aligned)
            uint8_t *bstrip16 = nullptr;
            if (M == 8 && LINES == 16 && (g.stride & 15) == 0 && (tbase & 15) == 0 &&
                (((uintptr_t)flags | (uintptr_t)(HAS_DIR ? dmask : flags)) & 15) == 0)
                bstrip16 = reinterpret_cast<uint8_t *>(sm + 7 * LINES * (Lp + 1));
            load_segment_raw_buf<M, HAS_DIR, HAS_Q, FCM>(in + tbase, flags + tbase, coeff + tbase, HAS_DIR ? dmask + tbase : dmask,
                                                    HAS_DIR ? dval + tbase : dval, HAS_Q ? qf + tbase : qf, g, voff, s, R, bstrip,
                                                    bstrip16, (unsigned)tid);
        }
        else
            load_segment_raw<M, HAS_DIR, HAS_Q, FUSE, FCM>(in, flags, coeff, dmask, dval, qf, g, base, r0, active, s, R, fz, bstrip, tid);
        if constexpr (!FUSE && CORR) {
            // deferred interface correction of a slab decomposition (SweepScal::c_*): block-uniform, off in ordinary sweeps
            if (s.c_w != nullptr)
                corr_apply<M>(s, corr_weights(s, to), to, (voff + (unsigned)(ti * LINES)) * 8u, (unsigned)(g.stride * 8), R.vin,
                              Lp * M == g.n && (ti + 1) * LINES <= g.n_inner);
        }
#pragma unroll
        for (int r = 0; r < M; ++r) {
            double ar, cr;
            assemble_one<M, HAS_DIR, HAS_Q, FCM>(R, r, g.lbit, s, ar, b[r], cr, d[r]);
            a.m |= (ar != 0.0 ? 1u : 0u) << r;          // (assemble_row: a, c are -theta*gamma or 0)
            c.m |= (cr != 0.0 ? 1u : 0u) << r;
        }
    }
    // line ends: fold the coupling to the neighbouring GPU's row into the right-hand side
    if (r0 == 0) {
        if (xlo != nullptr && active) d[0] = __builtin_fma(-a[0], xlo[line_id], d[0]);
        a.m &= ~1u;
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (r0 + r == g.n - 1) {
            if (xhi != nullptr && active) d[r] = __builtin_fma(-c[r], xhi[line_id], d[r]);
            c.m &= ~(1u << r);
        }
    double ip[M - 1];
    Cond k;
    condense<M>(a, b, c, d, ip, k);
    double xL, xS;
    tile_separators(sm, tid, kk, sg, Lp, LINES, a[M - 1], b[M - 1], c[M - 1], d[M - 1], k, xL, xS);
    double x[M];
    back_solve<M>(a, c, d, ip, xL, xS, x);
    if (whole) {
        const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)(out + tbase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < M; ++r) __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(x[r]), rO, voff * 8u, (unsigned)r * (unsigned)(g.stride * 8), 0);
        return;
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
        if (active && (r0 + r) < g.n) out[base + (long)(r0 + r) * g.stride] = x[r];   // (nt stores: 5-10 % slower on the 64-byte row pieces of these tiles)
}

// GENERAL kernel: every tile (QUEUED = false) or the tiles a FAST kernel queued (QUEUED = true: a grid-stride loop over the
// unit queue).  Two instantiations, not a run-time branch: with both bodies in one kernel the values the loop keeps alive
// across its iterations were spilled (16 - 20 B of scratch per lane in the 42 B/cell kernel), and scratch set up for the
// direct form as well.
#ifndef ADI_GEN_OCC
#define ADI_GEN_OCC 1     // waves per SIMD the direct 8-row buffer-addressed build is compiled for (A/B: scripts/ab_build.sh)
#endif
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool WHOLE = false, int FCM = 0, bool CORR = true, bool QUEUED = false>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512, (M == 8 && WHOLE && !QUEUED) ? ADI_GEN_OCC : 1) void k_sweep_strided(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ xlo, const double *__restrict__ xhi, SweepScal s, const unsigned *__restrict__ queue,
    int ratio, int tiles_inner_f, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    if constexpr (!QUEUED) {
        strided_tile_general<M, HAS_DIR, HAS_Q, FUSE, WHOLE, FCM, CORR>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES, tiles_inner,
                                                                        xcd_chunk_tile(blockIdx.x, ntiles), xlo, xhi, s, sm, fz, (int)threadIdx.x);
    } else {
        // a queued unit is a tile of the FAST kernel = `ratio` adjacent tiles of this kernel
        const long cnt = (long)queue[0] * ratio;
        for (long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const long u = queue[1 + (unsigned)i / (unsigned)ratio];
            const long to = (long)((unsigned)u / (unsigned)tiles_inner_f);
            const long tig = (u - to * tiles_inner_f) * ratio + ((unsigned)i % (unsigned)ratio);
            // the thread index goes through an opaque move in every iteration: otherwise everything that depends on it alone --
            // per-row offsets and pointers, 2*M registers in the flat-addressed form -- is hoisted out of the loop and kept alive
            // across it, which is what this kernel's scratch was
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
            if (tig < tiles_inner)
                strided_tile_general<M, HAS_DIR, HAS_Q, FUSE, WHOLE, FCM, CORR>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES,
                                                                                tiles_inner, to * tiles_inner + tig, xlo, xhi, s, sm, fz, tid);
            __syncthreads();   // the LDS arrays are reused by the next tile
        }
    }
}


// Launch of the GENERAL kernel on the tiles of plan P (queue == nullptr: every tile; else the units a FAST kernel queued).
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE, int FCM, bool CORR, bool QUEUED>
inline void launch_strided_general_q(const StridedPlan &P, unsigned ggrid, const double *in, const uint8_t *flags,
                                     const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                                     const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                                     const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    // every tile whole and within 31-bit byte offsets of its base: the buffer-addressed instantiation (8 / 16 rows per thread)
    if constexpr (!FUSE && M >= 8) {
        const bool whole = kBufStrided && g.n_inner % P.lines_g == 0 && P.Lpg * M == g.n && (long)g.n * g.stride * 8 < 0x7fffffffL;
        if (whole) {
            hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q, false, true, FCM, CORR, QUEUED>), dim3(ggrid),
                               dim3(P.lines_g * P.Lpg), P.lds_g, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpg, P.lines_g,
                               P.tiles_inner_g, P.ntiles_g, xlo, xhi, s, queue, P.ratio, P.tiles_inner_f, fz);
            return;
        }
    }
    hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q, FUSE, false, FCM, CORR, QUEUED>), dim3(ggrid), dim3(P.lines_g * P.Lpg),
                       P.lds_g, st, in, flags, coeff, dmask, dval, qf, out, g, P.Lpg, P.lines_g, P.tiles_inner_g, P.ntiles_g, xlo,
                       xhi, s, queue, P.ratio, P.tiles_inner_f, fz);
}
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE, int FCM, bool CORR>
inline void launch_strided_general_t(const StridedPlan &P, unsigned ggrid, const double *in, const uint8_t *flags,
                                     const double *coeff, const uint8_t *dmask, const double *dval, const double *qf, double *out,
                                     const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                                     const unsigned *queue, hipStream_t st, const Fuse &fz)
{
    if (queue != nullptr)
        launch_strided_general_q<M, HAS_DIR, HAS_Q, FUSE, FCM, CORR, true>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi,
                                                                           s, queue, st, fz);
    else
        launch_strided_general_q<M, HAS_DIR, HAS_Q, FUSE, FCM, CORR, false>(P, ggrid, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi,
                                                                            s, queue, st, fz);
}

// adi_sweep_strided_gc.hip: <FCM = 1, CORR = false> for 8 / 16 rows per thread (fused or not)
void strided_general_fc(int m, bool has_dir, bool has_q, bool fuse, const StridedPlan &P, unsigned ggrid, const double *in,
                        const uint8_t *flags, const double *coeff, const uint8_t *dmask, const double *dval, const double *qf,
                        double *out, const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                        const unsigned *queue, hipStream_t st, const Fuse &fz);
// adi_sweep_strided_gk.hip: <FCM = 0, CORR = true> for 8 / 16 rows per thread (adi_sweep_corrected: never fused)
void strided_general_corr(int m, bool has_dir, bool has_q, const StridedPlan &P, unsigned ggrid, const double *in,
                          const uint8_t *flags, const double *coeff, const uint8_t *dmask, const double *dval, const double *qf,
                          double *out, const LineGeom &g, const double *xlo, const double *xhi, const SweepScal &s,
                          const unsigned *queue, hipStream_t st, const Fuse &fz);

}  // namespace adi
