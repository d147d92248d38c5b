// adi_sweep_strided.hip -- K2: the batched tridiagonal sweeps along the strided axes (memory axes 0 and 1): sweep_axis0/1
// + thomas_solve of adi3d_numba_coeff.py:133-202, :121-130 (identity-row form of adi3d_gpu_coeff.py:154-191), the form
// with the explicit stage (lap1D_x/y/z + R0, :240-298) folded into the loads of the axis-0 sweep, and K4, the
// thread-per-line sweep for lines beyond 1024 rows.  Hand-written HIP for gfx950; HBM-bound, no MFMA.
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
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool WHOLE = false>
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
            load_segment_raw_buf<M, HAS_DIR, HAS_Q>(in + tbase, flags + tbase, coeff + tbase, HAS_DIR ? dmask + tbase : dmask,
                                                    HAS_DIR ? dval + tbase : dval, HAS_Q ? qf + tbase : qf, g, voff, s, R, bstrip,
                                                    bstrip16);
        }
        else
            load_segment_raw<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, g, base, r0, active, s, R, fz, bstrip);
        if constexpr (!FUSE) {
            // deferred interface correction of a slab decomposition (SweepScal::c_*): block-uniform, off in ordinary sweeps
            if (s.c_w != nullptr)
                corr_apply<M>(s, corr_weights(s, to), to, (voff + (unsigned)(ti * LINES)) * 8u, (unsigned)(g.stride * 8), R.vin,
                              Lp * M == g.n && (ti + 1) * LINES <= g.n_inner);
        }
#pragma unroll
        for (int r = 0; r < M; ++r) {
            double ar, cr;
            assemble_one<M, HAS_DIR, HAS_Q>(R, r, g.lbit, s, ar, b[r], cr, d[r]);
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

// GENERAL kernel: every tile (queue == nullptr) or the tiles a FAST kernel queued
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, bool WHOLE = false>
__global__ __launch_bounds__(M <= 8 ? 1024 : 512) void k_sweep_strided(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    double *__restrict__ out, LineGeom g, int Lp, int LINES, int tiles_inner, long ntiles,
    const double *__restrict__ xlo, const double *__restrict__ xhi, SweepScal s, const unsigned *__restrict__ queue,
    int ratio, int tiles_inner_f, Fuse fz)
{
    extern __shared__ __align__(16) double sm[];
    if (queue == nullptr) {
        strided_tile_general<M, HAS_DIR, HAS_Q, FUSE, WHOLE>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES, tiles_inner,
                                                             xcd_chunk_tile(blockIdx.x, ntiles), xlo, xhi, s, sm, fz);
    } else {
        // a queued unit is a tile of the FAST kernel = `ratio` adjacent tiles of this kernel
        const long cnt = (long)queue[0] * ratio;
        for (long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const long u = queue[1 + (unsigned)i / (unsigned)ratio];
            const long to = (long)((unsigned)u / (unsigned)tiles_inner_f);
            const long tig = (u - to * tiles_inner_f) * ratio + ((unsigned)i % (unsigned)ratio);
            if (tig < tiles_inner)
                strided_tile_general<M, HAS_DIR, HAS_Q, FUSE, WHOLE>(in, flags, coeff, dmask, dval, qf, out, g, Lp, LINES,
                                                                     tiles_inner, to * tiles_inner + tig, xlo, xhi, s, sm, fz);
            __syncthreads();   // the LDS arrays are reused by the next tile
        }
    }
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
    const double2 cw = corr_weights(s, o);            // deferred interface correction (SweepScal::c_*), off in ordinary sweeps
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * g.stride;
        const unsigned f = flags[p];
        double a, b, c, d;
        double vin = in[p];
        const long cell = (long)r * g.stride + kc * inner_stride, qo = s.c_n - 1 - o;
        if (cw.x != 0.0)
            vin = __builtin_fma(s.c_wl != nullptr ? s.c_wl[(o < s.c_np ? o : s.c_np - 1) * s.c_ps + cell] : cw.x, s.c_lo[cell], vin);
        if (cw.y != 0.0)
            vin = __builtin_fma(s.c_wh != nullptr ? s.c_wh[(qo < s.c_np ? qo : s.c_np - 1) * s.c_ps + cell] : cw.y, s.c_hi[cell], vin);
        assemble_row<HAS_DIR, HAS_Q>(f & 1u, (f >> g.lbit) & 1u, (f >> (g.lbit + 1)) & 1u, HAS_DIR && dmask[p] != 0, vin,
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
// host-side launch logic
// ------------------------------------------------------------------------------------------------

template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE>
static void launch_strided(const StridedPlan &P, const double *in, const uint8_t *flags, const double *coeff,
                           const uint8_t *dmask, const double *dval, const double *qf, double *out, const LineGeom &g,
                           const double *xlo, const double *xhi, SweepScal s, unsigned *queue, hipStream_t st,
                           const Fuse &fz)
{
    unsigned ggrid = (unsigned)P.ntiles_g;
    if (queue != nullptr) {
        const bool nofb = s.nofb != 0;                 // promise: no tile will be queued (see SweepScal)
        if (nofb) queue = nullptr;
        else (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        if (P.Mf == 20 || P.Mf == 24 || P.Mf == 28)      // exact fits: instantiated in adi_sweep_strided_x.hip
            strided_fast_exact(P.Mf, HAS_DIR, HAS_Q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 18 || P.Mf == 22 || P.Mf == 26 || P.Mf == 30)   // ... and adi_sweep_strided_y.hip
            strided_fast_exact2(P.Mf, HAS_DIR, HAS_Q, P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 10 || P.Mf == 12 || P.Mf == 14)  // exact fits of the fused kernel (no Dirichlet cells): adi_sweep_strided_fx.hip
            strided_fast_fused_exact(P.Mf, HAS_Q, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 9 || P.Mf == 11 || P.Mf == 13 || P.Mf == 15)   // ... and the odd row counts: adi_sweep_strided_fy.hip
            strided_fast_fused_exact_odd(P.Mf, HAS_Q, P, in, flags, coeff, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 32) launch_strided_fast<32, HAS_DIR, HAS_Q, false>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else if (P.Mf == 16) launch_strided_fast<16, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        else launch_strided_fast<8, HAS_DIR, HAS_Q, FUSE>(P, in, flags, coeff, dmask, dval, qf, out, g, xlo, xhi, s, queue, st, fz);
        ggrid = P.ntiles_g < 1024 ? (unsigned)P.ntiles_g : 1024u;
        if (nofb) return;
    }
    // every tile whole and within 31-bit byte offsets of its base: the buffer-addressed instantiation (8 / 16 rows per thread)
    const bool whole = !FUSE && M >= 8 && kBufStrided && g.n_inner % P.lines_g == 0 && P.Lpg * M == g.n &&
                       (long)g.n * g.stride * 8 < 0x7fffffffL;
    if constexpr (!FUSE && M >= 8) {
        if (whole) {
            hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q, false, true>), dim3(ggrid), dim3(P.lines_g * P.Lpg), P.lds_g, st,
                               in, flags, coeff, dmask, dval, qf, out, g, P.Lpg, P.lines_g, P.tiles_inner_g, P.ntiles_g, xlo,
                               xhi, s, queue, P.ratio, P.tiles_inner_f, fz);
            return;
        }
    }
    hipLaunchKernelGGL((k_sweep_strided<M, HAS_DIR, HAS_Q, FUSE>), dim3(ggrid), dim3(P.lines_g * P.Lpg), P.lds_g, st, in,
                       flags, coeff, dmask, dval, qf, out, g, P.Lpg, P.lines_g, P.tiles_inner_g, P.ntiles_g, xlo, xhi,
                       s, queue, P.ratio, P.tiles_inner_f, fz);
}

template <bool HAS_DIR, bool HAS_Q>
static void strided_sweep_t(const SweepArgs &a, const Lay &L, const LineGeom &g, const SweepScal &s, double *out,
                            const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                            const Fuse *fzp)
{
    StridedPlan P = strided_plan(g, s.sparse != 0 && work != nullptr, false, fzp != nullptr, fzp != nullptr && !HAS_DIR);
    unsigned *queue = nullptr;
    if (P.Mf && use_fast(s, work, work_bytes, P.ntiles_f)) queue = (unsigned *)work;
    else if (P.Mf) P = strided_plan(g, false, false);
    if (fzp != nullptr) {
        Fuse fz = *fzp;
        if (queue != nullptr && !fuse_fast_ok(P, L, fz)) { queue = nullptr; P = strided_plan(g, false, false); }
        fuse_tile_order(fz, P, L);
        switch (P.Mg) {
            case 2: launch_strided<2, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 4: launch_strided<4, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            case 8: launch_strided<8, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
            default: launch_strided<16, HAS_DIR, HAS_Q, true>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        }
        return;
    }
    const Fuse fz = Fuse();
    switch (P.Mg) {
        case 2: launch_strided<2, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        case 4: launch_strided<4, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        case 8: launch_strided<8, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
        default: launch_strided<16, HAS_DIR, HAS_Q, false>(P, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, xlo, xhi, s, queue, st, fz); break;
    }
}

void strided_sweep(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const LineGeom &g, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                   const Fuse *fz)
{
    if (has_dir && has_q) strided_sweep_t<true, true>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else if (has_q) strided_sweep_t<false, true>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else if (has_dir) strided_sweep_t<true, false>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
    else strided_sweep_t<false, false>(a, L, g, s, out, xlo, xhi, work, work_bytes, st, fz);
}

void generic_sweep(bool has_dir, bool has_q, const SweepArgs &a, const LineGeom &g, long inner_stride, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, double *wc, double *wd, hipStream_t st)
{
    const long nl = (long)g.n_inner * g.n_outer;
    const dim3 grid((unsigned)((nl + 255) / 256)), block(256);
    if (has_dir && has_q) hipLaunchKernelGGL((k_sweep_generic<true, true>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else if (has_q) hipLaunchKernelGGL((k_sweep_generic<false, true>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else if (has_dir) hipLaunchKernelGGL((k_sweep_generic<true, false>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
    else hipLaunchKernelGGL((k_sweep_generic<false, false>), grid, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, g, inner_stride, xlo, xhi, wc, wd, s);
}

}  // namespace adi
