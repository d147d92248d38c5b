// adi_sweep_contig.hip -- K3: the batched tridiagonal sweep along the contiguous axis (memory axis 2; BASELINE.json's
// "x-sweep"): sweep_axis2 + thomas_solve of adi3d_numba_coeff.py:205-237, :121-130 in the full-length identity-row form
// of adi3d_gpu_coeff.py:154-191.  Hand-written HIP for gfx950; HBM-bound, no MFMA.
#include "adi_contig_dev.hpp"

namespace adi {

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
        if (s.fconst) {                    // per-face scalars: coefficient / flux of the exposed cells from their flags
#pragma unroll
            for (int r = 0; r < M; ++r) {
                const bool ex = axis_exposed(fb[r], 5), hl = (fb[r] >> 5) & 1u, hh = (fb[r] >> 6) & 1u;
                vco[r] = ex ? pack_co(s, nullptr, hl, hh) : 0.0;
                vq[r] = (HAS_Q && ex) ? pack_q<HAS_Q>(s, nullptr, hl, hh) : 0.0;
            }
        } else {
            load_rows_contig<M, VEC>(coeff, base, r0, n, active && need, vco);
            if (HAS_Q) load_rows_contig<M, VEC>(qf, base, r0, n, active && need, vq);
        }
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
        coal_store<M>(out + wbase, strip, lane, x, s.nt != 0);
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

// ------------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------------


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
    // Segment counts that are not a power of two leave lanes of the in-wave PCR idle (n = 320: 20 segments of 16 rows in 32
    // lanes, 37 % of every wave padding; 209 against 270 Gcell/s at nz = 256).  Where 20, 24 or 28 rows per lane cut the
    // line into exactly 16 or 32 segments the FAST kernel takes that many (n = 320, 384, 448, 640, 768, 896; lines beyond
    // kMaxFastLine = 1024 rows never get here: sweep_entry sends them to the thread-per-line kernel)
    const int Mf = contig_fast_rows(n, M);
    const int lwf = 64 / next_pow2((n + Mf - 1) / Mf);
    const long nunits_f = (nlines + lwf - 1) / lwf;
    const bool fast = use_fast(s, work, work_bytes, nunits_f) && (n % Mf == 0) && (lwf % lw == 0);
    unsigned *queue = fast ? (unsigned *)work : nullptr;
    unsigned ggrid = grid;
    if (fast) {
        const bool nofb = s.nofb != 0;                 // promise: no unit will be queued (see SweepScal)
        if (nofb) queue = nullptr;
        else (void)hipMemsetAsync(queue, 0, sizeof(unsigned), st);
        const bool vecf = aligned && (n % Mf == 0);
        if (Mf > 16)          // 20 / 24 / 28 rows per lane: instantiated in adi_sweep_contig_x.hip
            contig_fast_exact(Mf, HAS_DIR, HAS_Q, in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        else if (Mf == 16 && M != 16)
            launch_contig_fast<16, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        else if (Mf == 8 && M != 8)
            launch_contig_fast<8, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        else
            launch_contig_fast<M, HAS_DIR, HAS_Q>(in, flags, coeff, dmask, dval, qf, out, L, s, vecf, nunits_f, queue, st);
        ggrid = grid < 2048u ? grid : 2048u;
        if (nofb) return;
    }
    const int ratio = fast ? lwf / lw : 1;
    // coalesced + LDS-transposed access with streaming loads/stores: 1.01 -> 0.92 ms on the dense general-pack sweep at
    // 512^3 (with the default cache policy the transposition alone gained nothing)
    const bool coal = vec && M >= 4 && Lp * M == n && (L.ny % lw == 0) && (nlines % lw == 0);
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

template <bool HAS_DIR, bool HAS_Q>
static void contig_sweep_t(const SweepArgs &a, const Lay &L, const SweepScal &s, double *out, void *work, size_t work_bytes,
                           hipStream_t st)
{
    switch (contig_rows_per_lane(L.nz)) {
        case 2: launch_contig<2, HAS_DIR, HAS_Q>(a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, L, s, work, work_bytes, st); break;
        case 4: launch_contig<4, HAS_DIR, HAS_Q>(a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, L, s, work, work_bytes, st); break;
        case 8: launch_contig<8, HAS_DIR, HAS_Q>(a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, L, s, work, work_bytes, st); break;
        default: launch_contig<16, HAS_DIR, HAS_Q>(a.in, a.flags, a.coeff, a.dmask, a.dval, a.qf, out, L, s, work, work_bytes, st); break;
    }
}

void contig_sweep(bool has_dir, bool has_q, const SweepArgs &a, const Lay &L, const SweepScal &s, double *out, void *work,
                  size_t work_bytes, hipStream_t st)
{
    if (has_dir && has_q) contig_sweep_t<true, true>(a, L, s, out, work, work_bytes, st);
    else if (has_q) contig_sweep_t<false, true>(a, L, s, out, work, work_bytes, st);
    else if (has_dir) contig_sweep_t<true, false>(a, L, s, out, work, work_bytes, st);
    else contig_sweep_t<false, false>(a, L, s, out, work, work_bytes, st);
}

}  // namespace adi
