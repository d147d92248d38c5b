// adi_cart_host.hpp -- host-side planning shared by the Cartesian translation units of libadi_hip.so: line geometry of
// a sweep, the tiling of the strided kernels, and the launchers each translation unit exports to the C-ABI layer
// (adi_cart_api.hip).  The tiling constants are the measured optima of round 1; the run-time
// tuning knobs they were swept with are gone.
#pragma once
#include "adi_cart_dev.hpp"

namespace adi {

inline LineGeom line_geom(int axis, const Lay &L, long *inner_stride)
{
    LineGeom g;
    g.lbit = 1 + 2 * axis;
    if (axis == 0) { g.n = L.nx; g.stride = L.sx; g.n_inner = L.ny * L.nz; g.n_outer = 1; g.outer_stride = 0; *inner_stride = 1; }
    else if (axis == 1) { g.n = L.ny; g.stride = L.nz; g.n_inner = L.nz; g.n_outer = L.nx; g.outer_stride = L.sx; *inner_stride = 1; }
    else { g.n = L.nz; g.stride = 1; g.n_inner = L.ny; g.n_outer = L.nx; g.outer_stride = L.sx; *inner_stride = L.nz; }
    return g;
}

// sparse packs + a workspace for the unit queue -> FAST kernel first, GENERAL kernel on what it queued
inline bool use_fast(const SweepScal &s, void *work, size_t work_bytes, long nunits)
{
    return s.sparse && work != nullptr && work_bytes >= (size_t)(nunits + 1) * sizeof(unsigned);
}

inline int strided_rows_per_thread(int n) { return n <= 16 ? 2 : (n <= 32 ? 4 : (n <= 512 ? 8 : 16)); }

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

inline StridedPlan strided_plan(const LineGeom &g, bool want_fast, bool wide_ok, bool fused = false, bool fused_exact = false)
{
    StridedPlan P;
    const int n = g.n;
    P.Mg = strided_rows_per_thread(n);
    P.Lpg = next_pow2((n + P.Mg - 1) / P.Mg);
    P.Mf = 0; P.Lpf = 0; P.lines_f = 0; P.tiles_inner_f = 0; P.ntiles_f = 0; P.ratio = 1; P.lds_f = 0;
#ifndef ADI_DENSE_LINES
#define ADI_DENSE_LINES 8
#endif
    int lines = 8;                                    // lines per tile of the GENERAL kernel (8 B * lines contiguous per row)
    if (P.Mg > 8 && lines * P.Lpg > 512) lines = 8;
    if (!want_fast && P.Mg == 8 && ADI_DENSE_LINES * P.Lpg <= 1024) lines = ADI_DENSE_LINES;   // the dense (42 B/cell) sweeps
    const int maxg = (P.Mg <= 8) ? 1024 : 512;
    if (want_fast && n >= 64 && (long)n * g.stride < (1L << 31)) {   // 32-bit element offsets in the FAST kernels
        int mf = 0, lf = 0;
        // 32-line tiles with n/16 rows per thread: only where the kernel keeps few arrays alive (pass A); the solve
        // kernel needs > 200 VGPRs at 32 rows per thread and runs faster on 16-line tiles with 16 rows
        // exact fits first: 20 / 24 / 28 rows per thread where they cut the line into exactly 16 or 32 segments (n = 320, 384,
        // 448, 640, 768, 896) -- with 16 or 32 rows those lines fill 20 - 28 of 32 segment slots of every workgroup
        int exact = 0;
        // (late round 3: 18 / 22 / 26 / 30 rows too -- 288, 352, 416, 480 and 576 ... 960 rows; adi_sweep_strided_y.hip)
        if (!fused && !wide_ok && n >= 288 && !(n % 16 == 0 && ((n / 16) & (n / 16 - 1)) == 0))
            for (int m = 18; m <= 30 && !exact; m += 2)
                if (n % m == 0 && (n / m == 16 || n / m == 32)) exact = m;
        // ... and for the fused explicit + sweep kernel, which holds at most 16 rows per thread, 10 / 12 / 14 rows where they cut
        // the line into exactly 16 or 32 segments (n = 160, 192, 224, 320, 384, 448; adi_sweep_strided_fx.hip, round 3)
        // (... and 9 / 11 / 13 / 15 rows: 144, 176, 208, 240 and 288, 352, 416, 480; adi_sweep_strided_fy.hip -- every multiple of 16
        // up to 256 rows and of 32 up to 512 is an exact fit, which is what the padded extents round ragged lines up to)
        if (fused && fused_exact && n >= 144 && !(n % 16 == 0 && ((n / 16) & (n / 16 - 1)) == 0))
            for (int m = 9; m <= 15 && !exact; ++m)
                if (n % m == 0 && (n / m == 16 || n / m == 32)) exact = m;
        if (exact) { mf = exact; lf = 16; }
        else if (wide_ok && n % 16 == 0 && (n / 16 == 8 || n / 16 == 16 || n / 16 == 32)) { mf = n / 16; lf = 32; }   // Lpf = 16
        else if (n >= 512 && n % 32 == 0 && n / 32 <= 32 && !fused && g.stride <= 131072) {
            // long lines, rows less than 1 MiB apart: 32 rows per thread keep 16-line tiles in 512 threads (1024 x 128 x 256:
            // 112 -> 182 Gcell/s); with 2 MiB planes the 8-line tiles of 16 rows are the faster ones (233 vs 212).  From
            // n = 512 (16 segments: 256-thread workgroups, four per CU instead of two): axis 1 of 512^3 0.424 -> 0.414 ms
            mf = 32; lf = 16;
        } else {
            // 16 rows per thread from 160 rows where 16 divides the line (round 3, scripts/perf_map_small.py: with 8 rows a
            // 256-row line spread its interface solve over 32 lanes and the tile over 512 threads -- 256^3: fused kernel
            // 131 -> 159 Gcell/s, axis 1 226 -> 259, step 0.261 -> 0.230 ms; 160 / 192 / 224 rows along axis 0: 96 / 107 /
            // 120 -> 131 / 153 / 159 Gcell/s); beyond 256 rows always (as before: no FAST kernel when 16 does not divide)
            const int m2 = (n > 256 || (n >= 160 && n % 16 == 0)) ? 16 : 8;
            if (n % m2 == 0 && n / m2 <= 64) { mf = m2; lf = (16 * next_pow2(n / m2) > 512) ? 8 : 16; }
        }
        if (mf) {
            const int lpf = next_pow2(n / mf);
            int lg = lf;                                  // GENERAL lines: lf, or lf/2 when the workgroup gets too big
            while (lg * P.Lpg > maxg && lg > 8) lg >>= 1;
            // (workgroups of 128 threads are allowed since round 3: 64-row lines -- the slabs of a 512^3 grid cut over 8 GPUs --
            // had no FAST kernel at all, 16 lines x 8 segments being below the old minimum of 256: fused kernel 0.144 -> 0.095 ms)
            if (lf * lpf <= 512 && lf * lpf >= 128 && lg * P.Lpg <= maxg && lg * P.Lpg >= 64) {
                P.Mf = mf; P.Lpf = lpf; P.lines_f = lf; lines = lg; P.ratio = lf / lg;
            }
        }
    }
    if (P.Mf == 0) {
        while (lines * P.Lpg < 256) lines <<= 1;
        // (16-line tiles in 1024-thread workgroups for the dense 8-row kernel: measured slower, 1.30 vs 1.245 ms on axis 0 at 512^3)
    }
    P.lines_g = lines;
    P.tiles_inner_g = (g.n_inner + lines - 1) / lines;
    P.ntiles_g = (long)P.tiles_inner_g * g.n_outer;
    P.lds_g = (size_t)7 * lines * (P.Lpg + 1) * sizeof(double) + (size_t)lines * P.Lpg * 16;   // + byte-transposition strips
    if (P.Mf) {
        P.tiles_inner_f = (g.n_inner + P.lines_f - 1) / P.lines_f;
        P.ntiles_f = (long)P.tiles_inner_f * g.n_outer;
        P.lds_f = (size_t)7 * P.lines_f * (P.Lpf + 1) * sizeof(double) + (size_t)P.lines_f * P.Lpf * P.Mf;   // + byte strips
    }
    return P;
}

// the fused FAST kernels shuffle k-neighbours inside 16-lane DPP rows: 16 lines per tile, at most 16 rows per thread
inline bool fuse_fast_ok(const StridedPlan &P, const Lay &L, const Fuse &fz)
{
    return P.Mf != 0 && P.lines_f == 16 && P.Mf <= 16 && L.nz % 16 == 0 && fz.wbytes != 0;
}

inline void fuse_tile_order(Fuse &fz, const StridedPlan &P, const Lay &L)
{
    const int kg = 8;                // k-tiles per group walked j-fastest (swept 1..32 in round 1)
    fz.kt = 0; fz.ny = L.ny; fz.kg = 0;
    if (P.Mf != 0 && kg > 0 && L.nz % P.lines_f == 0) {
        const int kt = L.nz / P.lines_f;
        fz.kt = kt; fz.kg = kt < kg ? kt : kg;   // tiles_inner_f == ny * kt: the remap is a permutation of the tile ids (the
                                                 // k-tiles beyond the last whole group form a narrower one, tile_jfast)
    }
}

inline int make_lay(int nx, int ny, int nz, long plane_stride, Lay *L)
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

// rows per lane of the contiguous FAST kernel for lines of n rows (`m_general`: rows per lane of the GENERAL kernel the
// launcher was instantiated with); 0: the line has no FAST form.  launch_contig and the padding model below use it.
inline int contig_fast_rows(int n, int m_general)
{
    const auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (n >= 256 && !(n % 16 == 0 && pow2(n / 16)))
        for (int m = 20; m <= 28; m += 4)
            if (n % m == 0 && pow2(n / m) && n / m >= 8 && n / m <= 64) return m;
    if (n >= 128 && n % 16 == 0 && n / 16 <= 64) return 16;
    if (n >= 64 && n % 8 == 0 && n / 8 <= 64 && m_general < 8) return 8;
    return m_general;
}
inline int contig_rows_per_lane(int n) { return n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : 16)); }

// Padding model (adi_recommended_dims): the fraction of the lanes / segment slots of a workgroup that a line of n rows
// along `axis` fills -- 1 where a row count cuts it into a power-of-two number of segments, 17/32 for 272 rows on the FAST
// kernels -- times, where only the GENERAL kernels take the line, their measured rate relative to the FAST ones
// (scripts/perf_map.py on 250^3 / 300^3 against 256^3 / 320^3: explicit + axis-0 as two GENERAL-path kernels 0.38 of the
// fused FAST kernel, the other two sweeps 0.5).
inline double line_fill(int axis, int n, int ny, int nz)
{
    static const double general[3] = {0.38, 0.5, 0.5};
    const int mg = axis == 2 ? contig_rows_per_lane(n) : strided_rows_per_thread(n);
    const int sg = (n + mg - 1) / mg;
    const double gen = general[axis] * sg / next_pow2(sg);
    if (n < 64 || n > kMaxFastLine) return gen;
    if (axis == 2) {
        const int m = contig_fast_rows(n, contig_rows_per_lane(n));
        if (n % m != 0 || m < 8) return gen;
        // (a segment count that is not a power of two also loses the coalesced loads -- solid 400^3 222 Gcell/s, 496^3 259
        // against 355 - 365 at 448 / 512 rows -- but rating it 0.72 lower only traded cells for rate: 390^3 0.91 -> 0.97 ms,
        // 490^3 1.65 -> 1.60, 450^3 1.44 -> 1.48; not kept)
        return (double)(n / m) / next_pow2(n / m);
    }
    if (nz % 16 != 0) return gen;                        // the strided FAST tiles are 16 whole lines wide
    Lay L;
    L.nx = axis == 0 ? n : 64; L.ny = axis == 1 ? n : ny; L.nz = nz; L.sx = (long)L.ny * L.nz;
    long inner = 1;
    const LineGeom g = line_geom(axis, L, &inner);
    const StridedPlan P = strided_plan(g, true, false, axis == 0, axis == 0);
    if (!P.Mf) return gen;
    return (double)(n / P.Mf) / P.Lpf;
}

inline unsigned cell_blocks(const Lay &L) { return (unsigned)(((long)L.nx * L.ny * L.nz + 255) / 256); }

inline int variant_flags(int variant, bool *has_dir, bool *has_q)
{
    if (variant < 0 || variant > 3) return set_err(ADI_ERR_ARG, "bad sweep variant %d", variant);
    *has_dir = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_Q);
    *has_q = (variant == ADI_SWEEP_GENERAL || variant == ADI_SWEEP_NO_DIR);
    return ADI_OK;
}

// whole-segment condition of the tiled pass-A kernels (also what adi_explicit_fused_supported reports)
inline bool condense_is_tiled(int axis, int n)
{
    if (axis == 2 || n > kMaxFastLine) return false;
    const int mg = strided_rows_per_thread(n);
    return (n % mg == 0) && (n / mg <= 64);
}

inline int dots_ichunk(int np) { return np >= 512 ? 32 : (np / 16 < 4 ? 4 : np / 16); }

// ---- launchers exported by the kernel translation units ----------------------------------------------------------
// The arrays a variant does not read (has_dir / has_q false) may be null.
struct SweepArgs {
    const double *in;
    const uint8_t *flags;
    const double *coeff;
    const uint8_t *dmask;
    const double *dval;
    const double *qf;
};
// adi_sweep_contig.hip: memory axis 2
void contig_sweep(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const SweepScal &s, double *out, void *work,
                  size_t work_bytes, hipStream_t st);
// adi_sweep_strided.hip: memory axes 0 / 1 (fz != nullptr: the explicit stage folded into the loads, axis 0), and the
// thread-per-line sweep for lines beyond kMaxFastLine rows (HBM scratch wc, wd)
void strided_sweep(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const LineGeom &g, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, void *work, size_t work_bytes, hipStream_t st,
                   const Fuse *fz);
void generic_sweep(bool has_dir, bool has_q, const SweepArgs &a, const LineGeom &g, long inner_stride, const SweepScal &s,
                   double *out, const double *xlo, const double *xhi, double *wc, double *wd, hipStream_t st);
// adi_condense.hip: pass A of a sweep whose lines span several slabs
int condense_sweep(bool has_dir, bool has_q, int axis, const SweepArgs &a, const Lay &L, const SweepScal &s, double *cond,
                   void *work, size_t work_bytes, hipStream_t st, const Fuse *fz);
void condense_generic_lines(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const SweepScal &s, double *cond,
                            const unsigned *list, long line_begin, long nsel, hipStream_t st);

}  // namespace adi
