// adi_condense.hip -- K5: slab decomposition along memory axis 0 (BASELINE.json: "2/4/8-GPU runs exchange one ghost plane
// per sweep").  Pass A of a sweep whose lines continue on neighbouring GPUs condenses every local line to six numbers;
// k_interface / k_interface_pair solve the reduced interface system; pass B is the ordinary sweep with the two boundary
// values injected (adi_sweep_strided.hip).  No counterpart in the reference (single process); the per-line algebra is
// thomas_solve's (adi3d_numba_coeff.py:121-130) block elimination.
#include <vector>

#include "adi_cart_host.hpp"
#include "adi_strided_dev.hpp"

namespace adi {

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
    const SweepScal &s, double *sm, const Fuse &fz, const int tid)
{
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
        load_segment_raw<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, g, base, r0, active, s, R, fz, nullptr, tid);
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
                                                       tiles_inner, xcd_chunk_tile(blockIdx.x, ntiles), s, sm, fz, (int)threadIdx.x);
    } else {
        const long cnt = (long)queue[0] * ratio;
        for (long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const long u = queue[1 + (unsigned)i / (unsigned)ratio];
            const long to = (long)((unsigned)u / (unsigned)tiles_inner_f);
            const long tig = (u - to * tiles_inner_f) * ratio + ((unsigned)i % (unsigned)ratio);
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));      // (keeps the per-row offsets out of the loop-carried state, see k_sweep_strided)
            if (tig < tiles_inner)
                condense_tile_general<M, HAS_DIR, HAS_Q, FUSE>(in, flags, coeff, dmask, dval, qf, cond, nlines, g, Lp,
                                                               LINES, tiles_inner, to * tiles_inner + tig, s, sm, fz, tid);
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
            for (int r = 0; r < M; ++r) buf_store_f64(rO, voff * 8u, (unsigned)r * (unsigned)(g.stride * 8), d[r], s.nt != 0);
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

// The same solve for the deferred form without decay: only the right-hand sides (gF, gL) = (plane 0, plane n-1 of x0) are
// gathered every step, g_all [nranks][2][nlines]; the matrix entries are constants of the plan -- per line on the first and
// the last rank (their global end rows: mat_all [nranks][4][nlines] = aF, cF, aL, cL, gathered once), the same two numbers
// (-w0, -wn) for every line of a middle rank (scal_all [nranks][2]).  A third of the per-step payload and of this kernel's reads.
__global__ __launch_bounds__(256) void k_interface_uniform(const double *__restrict__ g_all, const double *__restrict__ mat_all,
                                                           const double *__restrict__ scal_all, int nranks, int rank,
                                                           long nlines, double *__restrict__ xlo, double *__restrict__ xhi)
{
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines) return;
    auto ld = [&](int r) {
        const double *g = g_all + (long)r * 2 * nlines + id;
        Cond k;
        k.gF = g[0]; k.gL = g[nlines];
        if (r == 0 || r == nranks - 1) {
            const double *m = mat_all + (long)r * 4 * nlines + id;
            k.aF = m[0]; k.cF = m[nlines]; k.aL = m[2 * nlines]; k.cL = m[3 * nlines];
        } else {
            const double w0 = scal_all[2 * r], wn = scal_all[2 * r + 1];
            k.aF = -w0; k.cF = -wn; k.aL = -wn; k.cL = -w0;
        }
        return k;
    };
    const Cond C = ld(rank);
    Cond P = {0, 0, 0, 0, 0, 0}, S = {0, 0, 0, 0, 0, 0};
    if (rank > 0) {
        P = ld(0);
        for (int r = 1; r < rank; ++r) P = merge_cond(P, ld(r));
    }
    if (rank < nranks - 1) {
        S = ld(nranks - 1);
        for (int r = nranks - 2; r > rank; --r) S = merge_cond(ld(r), S);
    }
    const double m00 = 1.0 - C.aF * P.cL, m01 = -C.cF * S.aF;      // (as in k_interface)
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
// host-side launch logic
// ------------------------------------------------------------------------------------------------
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

int condense_sweep(bool has_dir, bool has_q, int axis, const SweepArgs &a, const Lay &L, const SweepScal &s, double *cond,
                   void *work, size_t work_bytes, hipStream_t st, const Fuse *fz)
{
    if (has_dir && has_q) return condense_dispatch<true, true>(axis, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, L, s, cond, work, work_bytes, st, fz);
    if (has_q) return condense_dispatch<false, true>(axis, a.in, a.flags, a.coeff, nullptr, nullptr, a.qf, L, s, cond, work, work_bytes, st, fz);
    if (has_dir) return condense_dispatch<true, false>(axis, a.in, a.flags, a.coeff, a.dmask, a.dval, nullptr, L, s, cond, work, work_bytes, st, fz);
    return condense_dispatch<false, false>(axis, a.in, a.flags, a.coeff, nullptr, nullptr, nullptr, L, s, cond, work, work_bytes, st, fz);
}

// the serial two-recurrence condensation of the lines in `list` (axis 0; grid sized for all lines, the kernel returns
// beyond the list's count)
void condense_generic_lines(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const SweepScal &s, double *cond,
                            const unsigned *list, long line_begin, long nsel, hipStream_t st)
{
    long inner_stride;
    const LineGeom g = line_geom(0, L, &inner_stride);
    const long nlines = (long)L.ny * L.nz;
    const dim3 gl((unsigned)((nlines + 255) / 256)), block(256);
    if (has_dir && has_q) hipLaunchKernelGGL((k_condense_generic<true, true>), gl, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, cond, nlines, g, inner_stride, s, list, line_begin, nsel);
    else if (has_q) hipLaunchKernelGGL((k_condense_generic<false, true>), gl, block, 0, st, a.in, a.flags, a.coeff, nullptr, nullptr, a.qf, cond, nlines, g, inner_stride, s, list, line_begin, nsel);
    else if (has_dir) hipLaunchKernelGGL((k_condense_generic<true, false>), gl, block, 0, st, a.in, a.flags, a.coeff, a.dmask, a.dval, nullptr, cond, nlines, g, inner_stride, s, list, line_begin, nsel);
    else hipLaunchKernelGGL((k_condense_generic<false, false>), gl, block, 0, st, a.in, a.flags, a.coeff, nullptr, nullptr, nullptr, cond, nlines, g, inner_stride, s, list, line_begin, nsel);
}

// ------------------------------------------------------------------------------------------------
// K5c: interface values of the DEFERRED form.  Every rank has solved its part of every sharded-axis line with zero
// boundary values: x0.  By linearity the solution of the whole line is, inside the slab,
//     x[i] = x0[i] + xlo * w[i] + xhi * w[n-1-i],     w = theta*gamma * tridiag(-tg, 1+2tg, -tg)^-1 e_0,
// with xlo / xhi the unknowns adjacent to the slab on the neighbouring ranks.  Where w has decayed below rounding
// across a slab (the caller checks it, as for adi_interface_pair) the interface system splits into one 2 x 2 system per
// boundary: with L the last unknown of the slab below, F the first unknown of this slab and om = w[0],
//     L = x0_last(below) + om * F,    F = x0_first(mine) + om * L.
// first / last: planes 0 and n-1 of this rank's x0; prev_last / next_first: the adjacent planes of the neighbours' x0
// (null: no neighbour).  ulo = L of the boundary below, uhi = F of the boundary above; 0 where there is no neighbour.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_interface_deferred(const double *__restrict__ first, const double *__restrict__ last,
                                                            const double *__restrict__ prev_last,
                                                            const double *__restrict__ next_first, double om, double idet,
                                                            long nlines, double *__restrict__ ulo, double *__restrict__ uhi)
{
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    double lo = 0.0, hi = 0.0;
    if (prev_last != nullptr) {
        const double gL = prev_last[l];
        const double F = __builtin_fma(om, gL, first[l]) * idet;
        lo = __builtin_fma(om, F, gL);
    }
    if (next_first != nullptr) hi = __builtin_fma(om, last[l], next_first[l]) * idet;
    ulo[l] = lo;
    uhi[l] = hi;
}

// The same for lines that are NOT uniform (curved solids, voids, Dirichlet cells): every line has its own decaying
// homogeneous solutions, computed once per plan by two ordinary axis-0 sweeps with zero right-hand sides and unit boundary
// values.  om_*: their value in the plane next to the interface -- om_lo_own / om_hi_own of this rank's lines, om_hi_prev /
// om_lo_next received from the neighbours at plan time.  Per line and boundary a 2 x 2 system, as above with two weights.
__global__ __launch_bounds__(256) void k_interface_deferred_lines(
    const double *__restrict__ first, const double *__restrict__ last, const double *__restrict__ prev_last,
    const double *__restrict__ next_first, const double *__restrict__ om_lo_own, const double *__restrict__ om_hi_prev,
    const double *__restrict__ om_hi_own, const double *__restrict__ om_lo_next, long nlines, double *__restrict__ ulo,
    double *__restrict__ uhi, const uint8_t *__restrict__ uni_lo, const uint8_t *__restrict__ uni_hi,
    double *__restrict__ ulo_uni, double *__restrict__ uhi_uni)
{
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    double lo = 0.0, hi = 0.0;
    if (prev_last != nullptr) {
        const double gL = prev_last[l], wl = om_lo_own[l], wh = om_hi_prev[l];
        const double F = __builtin_fma(wl, gL, first[l]) / (1.0 - wl * wh);     // this rank's first unknown
        lo = __builtin_fma(wh, F, gL);                                          // the neighbour's last one: what w_lo multiplies
    }
    if (next_first != nullptr) {
        const double wh = om_hi_own[l], wl = om_lo_next[l];
        hi = __builtin_fma(wl, last[l], next_first[l]) / (1.0 - wl * wh);       // the neighbour's first unknown
    }
    ulo[l] = lo;
    uhi[l] = hi;
    // the same values on the lines whose rows within reach of the interface are uniform (they take the scalar weights inside
    // the axis-1 sweep); the others get their own weights from k_deferred_lines_apply, or none (lines outside the mask)
    if (ulo_uni != nullptr) ulo_uni[l] = (uni_lo != nullptr && uni_lo[l] != 0) ? lo : 0.0;
    if (uhi_uni != nullptr) uhi_uni[l] = (uni_hi != nullptr && uni_hi[l] != 0) ? hi : 0.0;
}

// x[i][cell[q]] += wc[r][q] * u[cell[q]]: the rank-one update of the flagged lines of one side of the slab (the lines that
// cross a void, the surface or a Dirichlet cell within reach of the interface).  Thread = one flagged line, blockIdx.y strides
// the K rows; consecutive q are mostly consecutive cells (runs along the contiguous axis), the compact weights are read
// coalesced.  The same fma as corr_apply: a line gives the same bits whichever way its correction is applied.
__global__ __launch_bounds__(256) void k_deferred_lines_apply(double *__restrict__ x, int nx, long sx, const int *__restrict__ cells,
                                                              long nflag, const double *__restrict__ wc, int K,
                                                              const double *__restrict__ u, int from_hi,
                                                              const int *__restrict__ nrows)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nflag) return;
    const long cell = cells[q];
    const double uu = u[cell];
    const int Kq = (nrows != nullptr && nrows[q] < K) ? nrows[q] : K;     // rows beyond carry exact zeros: not read
    for (int r = blockIdx.y; r < Kq; r += gridDim.y) {
        // a flagged line's weights are exact zeros beyond its first row outside the mask (identity rows cut the coupling): a
        // line through a cavity 150 planes from the interface carries 150 weights, not K; x + 0 * u = x is skipped unread
        const double w = wc[(long)r * nflag + q];
        if (w != 0.0) {
            const long i = from_hi ? (long)(nx - 1 - r) : (long)r;
            double *p = x + i * sx + cell;
            *p = __builtin_fma(w, uu, *p);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5d: the deferred form WITHOUT decay (thin slabs, strong scaling).  x = x0 + xlo * psi_lo + xhi * psi_hi still holds;
// what changes is that (1) the interface system couples all ranks -- the all-gather + k_interface of the `exact` form, fed
// with gF = x0_first, gL = x0_last (planes of x0) and the matrix entries below -- and (2) on the first / last rank the
// homogeneous solutions feel the global end row, whose diagonal differs from the uniform one by delta = dt * coeff - tg
// (a line start / end with its Robin coefficient).  Sherman-Morrison on the uniform matrix U (p0 = w0 / tg = (U^-1)_00,
// pn = wn / tg = (U^-1)_0,n-1), kappa = delta / (1 + delta p0):
//     first rank (row 0 modified):   psi_hi[i] = w[n-1-i] - kappa pn w[i]      (no psi_lo: nothing below)
//     last rank (row n-1 modified):  psi_lo[i] = w[i] - kappa pn w[n-1-i]      (no psi_hi)
// so the correction keeps the form  c_lo * w[i] + c_hi * w[n-1-i]  that adi_sweep_corrected applies, with per-line c_lo / c_hi.
// k_deferred_exact_setup: once per plan -- matrix entries mat [4][nlines] = (aF, cF, aL, cL) of this rank and kappa * pn per
// line and end (0 where the end row is an ordinary interior row).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deferred_exact_setup(const uint8_t *__restrict__ flags_first,
                                                              const uint8_t *__restrict__ flags_last,
                                                              const double *__restrict__ coeff_first,
                                                              const double *__restrict__ coeff_last, double tg, double dt,
                                                              double w0, double wn, long nlines, double *__restrict__ mat,
                                                              double *__restrict__ kap)
{
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    const bool has_lo = (flags_first[l] & 2u) != 0, has_hi = (flags_last[l] & 4u) != 0;   // the line continues below / above
    const double p0 = w0 / tg, pn = wn / tg;
    double aF = 0.0, cF = 0.0, aL = 0.0, cL = 0.0, klo = 0.0, khi = 0.0;
    if (has_lo && has_hi) {                       // a middle rank: psi_lo = w, psi_hi = reversed w
        aF = -w0; cF = -wn; aL = -wn; cL = -w0;
    } else if (has_hi) {                          // first rank: row 0 is the global line start
        const double delta = dt * coeff_first[l] - tg;
        const double kappa = delta / (1.0 + delta * p0);
        klo = kappa * pn;                         // psi_hi = rev(w) - klo * w
        cF = -(wn - klo * w0);
        cL = -(w0 - klo * wn);
    } else if (has_lo) {                          // last rank: row n-1 is the global line end
        const double delta = dt * coeff_last[l] - tg;
        const double kappa = delta / (1.0 + delta * p0);
        khi = kappa * pn;                         // psi_lo = w - khi * rev(w)
        aF = -(w0 - khi * wn);
        aL = -(wn - khi * w0);
    }
    mat[l] = aF; mat[nlines + l] = cF; mat[2 * nlines + l] = aL; mat[3 * nlines + l] = cL;
    kap[l] = klo; kap[nlines + l] = khi;
}

// per step, after the interface solve: the coefficients of w[i] and w[n-1-i] in the correction of every line
__global__ __launch_bounds__(256) void k_deferred_exact_coef(const double *__restrict__ xlo, const double *__restrict__ xhi,
                                                             const double *__restrict__ kap, long nlines,
                                                             double *__restrict__ clo, double *__restrict__ chi)
{
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    const double lo = xlo[l], hi = xhi[l];        // (0 where there is no neighbour: k_interface)
    clo[l] = __builtin_fma(-kap[l], hi, lo);            // first rank: xlo = 0, c_lo = -klo * xhi
    chi[l] = __builtin_fma(-kap[nlines + l], lo, hi);   // last rank:  xhi = 0, c_hi = -khi * xlo
}

}  // namespace adi

using namespace adi;

extern "C" {

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
int adi_axis0_deferred_setup(int n, double theta, double gam, double tol, double *d_w, double *h_omega, int *h_reach,
                             void *stream)
{
    ADI_REQUIRE(n >= 2 && d_w && h_omega && h_reach && tol >= 0.0, "adi_axis0_deferred_setup: bad argument");
    // w = tg * tridiag(-tg, 1+2tg, -tg)^-1 e_0 (n x n): Thomas in long double; the closure of the far end does not
    // matter where w has decayed there, which is the condition the caller tests (*h_reach < n)
    const long double tg = (long double)theta * (long double)gam, b = 1.0L + 2.0L * tg;
    std::vector<long double> cp(n), x(n);
    std::vector<double> w(n);
    long double piv = b;
    cp[0] = -tg / piv; x[0] = tg / piv;
    for (int i = 1; i < n; ++i) {
        piv = b + tg * cp[i - 1];
        cp[i] = -tg / piv;
        x[i] = (tg * x[i - 1]) / piv;
    }
    for (int i = n - 2; i >= 0; --i) x[i] -= cp[i] * x[i + 1];
    int reach = 0;
    for (int i = 0; i < n; ++i) {
        w[i] = (double)x[i];
        if (w[i] > tol) reach = i + 1; else w[i] = 0.0;      // (w decreases monotonically: exactly 0 beyond its reach)
    }
    *h_omega = w[0];
    *h_reach = reach;
    ADI_HIP_TRY(hipMemcpyAsync(d_w, w.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, as_stream(stream)));
    ADI_HIP_TRY(hipStreamSynchronize(as_stream(stream)));      // w lives on this stack frame
    return ADI_OK;
}

int adi_interface_deferred(const double *d_first, const double *d_last, const double *d_prev_last,
                           const double *d_next_first, double omega, long nlines, double *d_ulo, double *d_uhi,
                           void *stream)
{
    ADI_REQUIRE(d_ulo && d_uhi && nlines > 0, "adi_interface_deferred: bad argument");
    ADI_REQUIRE((!d_prev_last || d_first) && (!d_next_first || d_last), "adi_interface_deferred: missing own plane");
    ADI_REQUIRE(omega >= 0.0 && omega < 1.0, "adi_interface_deferred: omega = %g is not a decaying weight", omega);
    const double idet = 1.0 / (1.0 - omega * omega);
    hipLaunchKernelGGL(k_interface_deferred, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream),
                       d_first, d_last, d_prev_last, d_next_first, omega, idet, nlines, d_ulo, d_uhi);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}
int adi_interface_deferred_lines(const double *d_first, const double *d_last, const double *d_prev_last,
                                 const double *d_next_first, const double *d_om_lo_own, const double *d_om_hi_prev,
                                 const double *d_om_hi_own, const double *d_om_lo_next, long nlines, double *d_ulo,
                                 double *d_uhi, const uint8_t *d_uni_lo, const uint8_t *d_uni_hi, double *d_ulo_uni,
                                 double *d_uhi_uni, void *stream)
{
    ADI_REQUIRE(d_first && d_last && d_ulo && d_uhi && nlines > 0, "adi_interface_deferred_lines: bad argument");
    ADI_REQUIRE(!d_prev_last || (d_om_lo_own && d_om_hi_prev), "adi_interface_deferred_lines: lower boundary without its weights");
    ADI_REQUIRE(!d_next_first || (d_om_hi_own && d_om_lo_next), "adi_interface_deferred_lines: upper boundary without its weights");
    hipLaunchKernelGGL(k_interface_deferred_lines, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream),
                       d_first, d_last, d_prev_last, d_next_first, d_om_lo_own, d_om_hi_prev, d_om_hi_own, d_om_lo_next,
                       nlines, d_ulo, d_uhi, d_uni_lo, d_uni_hi, d_ulo_uni, d_uhi_uni);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_deferred_lines_apply(double *d_x, int nx, long plane_stride, long plane_cells, const int *d_cells, long nflag,
                             const double *d_wc, int K, const double *d_u, int from_high_end, const int *d_nrows, void *stream)
{
    ADI_REQUIRE(d_x && d_u && nx > 0 && plane_cells > 0 && plane_stride >= plane_cells && K >= 0 && K <= nx && nflag >= 0,
                "adi_deferred_lines_apply: bad argument");
    if (nflag == 0 || K == 0) return ADI_OK;
    ADI_REQUIRE(d_cells && d_wc, "adi_deferred_lines_apply: flagged lines without their cells / weights");
    const unsigned gy = (unsigned)(K < 64 ? K : 64);
    hipLaunchKernelGGL(k_deferred_lines_apply, dim3((unsigned)((nflag + 255) / 256), gy), dim3(256), 0, as_stream(stream), d_x, nx,
                       plane_stride, d_cells, nflag, d_wc, K, d_u, from_high_end, d_nrows);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_deferred_exact_setup(const uint8_t *d_flags_first, const uint8_t *d_flags_last, const double *d_coeff_first,
                             const double *d_coeff_last, double theta, double gam, double dt, double w0, double wn,
                             long nlines, double *d_mat, double *d_kap, void *stream)
{
    ADI_REQUIRE(d_flags_first && d_flags_last && d_coeff_first && d_coeff_last && d_mat && d_kap && nlines > 0,
                "adi_deferred_exact_setup: bad argument");
    ADI_REQUIRE(theta * gam > 0.0 && w0 > 0.0, "adi_deferred_exact_setup: needs theta*gam > 0");
    hipLaunchKernelGGL(k_deferred_exact_setup, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream),
                       d_flags_first, d_flags_last, d_coeff_first, d_coeff_last, theta * gam, dt, w0, wn, nlines, d_mat, d_kap);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_interface_solve_uniform(const double *d_g_all, const double *d_mat_all, const double *d_scal_all, int nranks, int rank,
                                long nlines, double *d_xlo, double *d_xhi, void *stream)
{
    ADI_REQUIRE(d_g_all && d_mat_all && d_scal_all && d_xlo && d_xhi && nranks >= 2 && rank >= 0 && rank < nranks && nlines > 0,
                "adi_interface_solve_uniform: bad argument");
    hipLaunchKernelGGL(k_interface_uniform, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream), d_g_all,
                       d_mat_all, d_scal_all, nranks, rank, nlines, d_xlo, d_xhi);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_deferred_exact_coef(const double *d_xlo, const double *d_xhi, const double *d_kap, long nlines, double *d_clo,
                            double *d_chi, void *stream)
{
    ADI_REQUIRE(d_xlo && d_xhi && d_kap && d_clo && d_chi && nlines > 0, "adi_deferred_exact_coef: bad argument");
    hipLaunchKernelGGL(k_deferred_exact_coef, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, as_stream(stream), d_xlo,
                       d_xhi, d_kap, nlines, d_clo, d_chi);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}
}  // extern "C"
