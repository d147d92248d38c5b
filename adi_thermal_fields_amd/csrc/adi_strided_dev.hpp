// adi_strided_dev.hpp -- segment loaders of the strided-axis kernels (sweeps along memory axes 0 and 1, and the slab
// condensation that shares their loads).  See adi_sweep_strided.hip for the tiling.
#pragma once
#include "adi_cart_dev.hpp"

namespace adi {

// The raw rows of a segment.  The flags bytes stay packed four to a register and the Dirichlet marks one bit per row until the
// row is assembled: as `unsigned fb[M]; bool dirb[M]` they held 2*M registers from the first load to the last assembled row of a
// kernel that sits at its 128-VGPR budget (the 42 B/cell kernel: 3 spilled VGPRs, 16 B of scratch; none with the packed form).
template <int M>
struct SegRaw {
    double vin[M], vco[M], vdv[M], vq[M];
    unsigned fw[(M + 3) / 4];
    unsigned dw;
    __device__ __forceinline__ unsigned f(int r) const { return (fw[r >> 2] >> (8 * (r & 3))) & 0xffu; }
    __device__ __forceinline__ bool dir(int r) const { return (dw >> r) & 1u; }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < (M + 3) / 4; ++i) fw[i] = 0u;
        dw = 0u;
    }
    __device__ __forceinline__ void or_f(int r, unsigned byte) { fw[r >> 2] |= (byte & 0xffu) << (8 * (r & 3)); }
    __device__ __forceinline__ void or_dir(int r, bool on) { dw |= (on ? 1u : 0u) << r; }
};
// bit r of the result: byte r of the packed 64-bit word is non-zero (Dirichlet marks of 8 rows)
__device__ __forceinline__ unsigned nonzero_bytes8(unsigned lo, unsigned hi)
{
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m |= (((lo >> (8 * r)) & 0xffu) != 0u ? 1u : 0u) << r;
        m |= (((hi >> (8 * r)) & 0xffu) != 0u ? 1u : 0u) << (r + 4);
    }
    return m;
}

// FCM: source of the coefficient / flux of an exposed row (pack_co): 0 run time, 1 the flags, 2 the arrays
template <int M, bool HAS_DIR, bool HAS_Q, bool FUSE = false, int FCM = 0>
__device__ __forceinline__ void load_segment_raw(
    const double *__restrict__ in, const uint8_t *__restrict__ flags, const double *__restrict__ coeff,
    const uint8_t *__restrict__ dmask, const double *__restrict__ dval, const double *__restrict__ qf,
    const LineGeom &g, long base, int r0, bool active, const SweepScal &s, SegRaw<M> &R, const Fuse &fz = Fuse(),
    uint8_t *bstrip = nullptr, int tid = threadIdx.x)
{
    // bstrip != nullptr (block-uniform; M == 8, 8-line tiles, whole tile, 8-byte aligned byte arrays): the flag and
    // Dirichlet bytes of a segment -- 8 rows x 8 lines -- are fetched as ONE 8-byte load per lane (lane kk takes row kk)
    // and transposed through a wave-private LDS strip, instead of 8 + 8 single-byte loads per thread.  These kernels
    // are bound by the issue rate of the vector-memory pipe (SQ_WAIT_INST_ANY 30 % of the wave cycles at 56 memory
    // instructions per 8 cells), not by bytes.
    R.clear();
    if (bstrip != nullptr) {
        if constexpr (M == 8) {
            const int kk = tid & 7;
            const long prow = base - kk + (long)(r0 + kk) * g.stride;       // row r0+kk, first line of the tile
            const unsigned long long fq = *reinterpret_cast<const unsigned long long *>(flags + prow);
            unsigned long long dq = 0;
            if (HAS_DIR) dq = *reinterpret_cast<const unsigned long long *>(dmask + prow);
            uint8_t *st = bstrip + (tid >> 3) * (HAS_DIR ? 128 : 64);   // this segment's strip: [line][row]
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                st[l * 8 + kk] = (uint8_t)(fq >> (8 * l));
                if (HAS_DIR) st[64 + l * 8 + kk] = (uint8_t)(dq >> (8 * l));
            }
            wave_lds_fence();
            const unsigned long long fpk = *reinterpret_cast<const unsigned long long *>(st + kk * 8);
            unsigned long long dpk = 0;
            if (HAS_DIR) dpk = *reinterpret_cast<const unsigned long long *>(st + 64 + kk * 8);
            wave_lds_fence();
            R.fw[0] = (unsigned)fpk; R.fw[1] = (unsigned)(fpk >> 32);
            R.dw = nonzero_bytes8((unsigned)dpk, (unsigned)(dpk >> 32));
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const bool ok = active && (r0 + r) < g.n;
        const long p = base + (long)(r0 + r) * g.stride;
        if (bstrip == nullptr) R.or_f(r, ok ? flags[p] : 0u);
        R.vin[r] = ok ? in[p] : 0.0;
    }
    if (FUSE) {
        // vin <- R0 of the explicit stage; every neighbour is loaded only where the flags byte says it exists
        // (rows beyond the line / inactive lanes have flags 0 and stay 0)
        double prev = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const long p = base + (long)(r0 + r) * g.stride;
            const unsigned f = R.f(r);
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
        const unsigned f = R.f(r);
        const bool need = ok && (!s.sparse || axis_exposed(f, g.lbit));
        if (HAS_DIR && bstrip == nullptr) R.or_dir(r, ok && dmask[p] != 0);
        const bool hl = (f >> g.lbit) & 1u, hh = (f >> (g.lbit + 1)) & 1u;
        if constexpr (FCM != 1) {                                                  // (FCM == 1: assemble_one forms them)
            R.vco[r] = need ? pack_co<FCM>(s, coeff + p, hl, hh) : 0.0;            // (fconst only comes with sparse)
            R.vq[r] = (HAS_Q && need) ? pack_q<HAS_Q, FCM>(s, qf + p, hl, hh) : 0.0;
        }
        R.vdv[r] = (HAS_DIR && ok && (!s.sparse || R.dir(r))) ? dval[p] : 0.0;
    }
}

// Flag (or Dirichlet) bytes of a whole segment of a 16-line tile -- M rows x 16 lines -- as ONE 16-byte load per lane (lane
// kk < M takes row kk of the segment) transposed through a wave-private LDS strip [line][row]: lane kk gets the bytes of
// its M rows packed four per register.  Replaces M single-byte loads per thread: the strided kernels are bound by the
// issue rate of the vector-memory pipe, not by bytes (SQ_WAIT_INST_ANY, profiles/r02_*).  `bt`: the tile's first byte
// of row 0 of the line (16-byte aligned rows: host / block-uniform check); voff_row0: r0*stride (elements), kk = lane & 15.
template <int M>
__device__ __forceinline__ void load_bytes_packed16(const uint8_t *bt, unsigned voff_row0, unsigned stride, int kk, uint8_t *strip,
                                                    unsigned (&w)[M / 4])
{
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)bt, 0, 0x7fffffff, 0x00020000);
    if (kk < M) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rB, voff_row0 + (unsigned)kk * stride, 0u, 0);
#pragma unroll
        for (int l = 0; l < 16; ++l) strip[l * M + kk] = (uint8_t)((q[l >> 2] >> (8 * (l & 3))) & 0xffu);
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < M / 4; ++i) w[i] = reinterpret_cast<const unsigned *>(strip + kk * M)[i];
    wave_lds_fence();
}
template <int M>
__device__ __forceinline__ unsigned packed_byte(const unsigned (&w)[M / 4], int r) { return (w[r >> 2] >> (8 * (r & 3))) & 0xffu; }

// The same for whole tiles (block-uniform precondition: every lane active, every thread owns M rows, rows within 31-bit
// byte offsets of the tile base) with buffer addressing: one descriptor per array based at the tile, the row offsets in
// scalar registers, ONE 32-bit per-thread offset for every load and store -- the flat-addressed form above keeps a 64-bit
// pointer per row alive from the first load to the last store (32 VGPRs at 8 rows), which is what pushed the 42 B/cell
// kernel over its 128-VGPR budget into scratch.
#ifndef ADI_GEN_PACK_AUX
#define ADI_GEN_PACK_AUX 0   // cache policy of the dense pack-array loads of the buffer-addressed GENERAL loader (2 = nt; A/B knob)
#endif
template <int AUX>
__device__ __forceinline__ double buf_load_f64_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    return __hiloint2double((int)v.y, (int)v.x);
}
template <int M, bool HAS_DIR, bool HAS_Q, int FCM = 0>
__device__ __forceinline__ void load_segment_raw_buf(
    const double *__restrict__ in_t, const uint8_t *__restrict__ flags_t, const double *__restrict__ coeff_t,
    const uint8_t *__restrict__ dmask_t, const double *__restrict__ dval_t, const double *__restrict__ qf_t,
    const LineGeom &g, unsigned voff, const SweepScal &s, SegRaw<M> &R, uint8_t *bstrip, uint8_t *bstrip16 = nullptr,
    unsigned tid = threadIdx.x)
{
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void *)in_t, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void *)flags_t, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void *)coeff_t, 0, 0x7fffffff, 0x00020000);
    const unsigned st = (unsigned)g.stride, st8 = st * 8u, vb = voff * 8u;
    R.clear();
    const bool packed = bstrip != nullptr || bstrip16 != nullptr;
    if (bstrip != nullptr) {
        if constexpr (M == 8) {
            // flag / Dirichlet bytes of the segment (8 rows x 8 lines) as ONE 8-byte load per lane, transposed in LDS
            const unsigned kk = tid & 7u;
            const unsigned prow = voff - kk + kk * st;                          // row kk of the segment, first line of the tile
            const u32x2 fq = __builtin_amdgcn_raw_buffer_load_b64(rF, prow, 0u, 0);
            u32x2 dq = {0u, 0u};
            if (HAS_DIR) {
                const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)dmask_t, 0, 0x7fffffff, 0x00020000);
                dq = __builtin_amdgcn_raw_buffer_load_b64(rM, prow, 0u, 0);
            }
            uint8_t *sp = bstrip + (tid >> 3) * (HAS_DIR ? 128 : 64);   // this segment's strip: [line][row]
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                sp[l * 8 + kk] = (uint8_t)((l < 4 ? fq.x : fq.y) >> (8 * (l & 3)));
                if (HAS_DIR) sp[64 + l * 8 + kk] = (uint8_t)((l < 4 ? dq.x : dq.y) >> (8 * (l & 3)));
            }
            wave_lds_fence();
            const unsigned long long fpk = *reinterpret_cast<const unsigned long long *>(sp + kk * 8);
            unsigned long long dpk = 0;
            if (HAS_DIR) dpk = *reinterpret_cast<const unsigned long long *>(sp + 64 + kk * 8);
            wave_lds_fence();
            R.fw[0] = (unsigned)fpk; R.fw[1] = (unsigned)(fpk >> 32);
            R.dw = nonzero_bytes8((unsigned)dpk, (unsigned)(dpk >> 32));
        }
    }
    if (bstrip16 != nullptr) {
        if constexpr (M == 8) {
            // 16-line tiles: the 8 rows x 16 lines of flag / Dirichlet bytes as one 16-byte load in 8 of the 16 lanes
            const unsigned k16 = tid & 15u;
            uint8_t *sp = bstrip16 + (tid >> 4) * 128;
            unsigned fw[2], dw[2] = {0u, 0u};
            load_bytes_packed16<8>(flags_t, voff - k16, st, (int)k16, sp, fw);
            if (HAS_DIR) load_bytes_packed16<8>(dmask_t, voff - k16, st, (int)k16, sp, dw);
            R.fw[0] = fw[0]; R.fw[1] = fw[1];
            R.dw = nonzero_bytes8(dw[0], dw[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        if (!packed) R.or_f(r, __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * st, 0));
        R.vin[r] = buf_load_f64(rT, vb, (unsigned)r * st8);
    }
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_Q ? qf_t : coeff_t), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_DIR ? dval_t : coeff_t), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_DIR ? dmask_t : flags_t), 0, 0x7fffffff, 0x00020000);
    if (FCM != 1 && !s.sparse) {                    // dense packs: every array in full, unconditional loads
#pragma unroll
        for (int r = 0; r < M; ++r) {
            if (HAS_DIR && !packed) R.or_dir(r, __builtin_amdgcn_raw_buffer_load_b8(rM, voff, (unsigned)r * st, 0) != 0);
            R.vco[r] = buf_load_f64_aux<ADI_GEN_PACK_AUX>(rC, vb, (unsigned)r * st8);
            R.vq[r] = HAS_Q ? buf_load_f64_aux<ADI_GEN_PACK_AUX>(rQ, vb, (unsigned)r * st8) : 0.0;
            R.vdv[r] = HAS_DIR ? buf_load_f64_aux<ADI_GEN_PACK_AUX>(rV, vb, (unsigned)r * st8) : 0.0;
        }
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const unsigned f = R.f(r);
            const bool need = axis_exposed(f, g.lbit);
            if (HAS_DIR && !packed) R.or_dir(r, __builtin_amdgcn_raw_buffer_load_b8(rM, voff, (unsigned)r * st, 0) != 0);
            if constexpr (FCM == 1) {
                // (assemble_one forms the coefficients from the flags)
            } else if (FCM == 0 && s.fconst) {
                const bool hl = (f >> g.lbit) & 1u, hh = (f >> (g.lbit + 1)) & 1u;
                R.vco[r] = need ? pack_co<1>(s, nullptr, hl, hh) : 0.0;
                R.vq[r] = (HAS_Q && need) ? pack_q<HAS_Q, 1>(s, nullptr, hl, hh) : 0.0;
            } else {
                R.vco[r] = need ? buf_load_f64(rC, vb, (unsigned)r * st8) : 0.0;
                R.vq[r] = (HAS_Q && need) ? buf_load_f64(rQ, vb, (unsigned)r * st8) : 0.0;
            }
            R.vdv[r] = (HAS_DIR && R.dir(r)) ? buf_load_f64(rV, vb, (unsigned)r * st8) : 0.0;
        }
    }
}

// FCM == 1 (coefficients from the flags): the loaders leave vco / vq alone and the two numbers are formed here, where the row
// is assembled -- formed in the loader they were sixteen more values alive across the whole load phase (straight-line selects:
// the scheduler hoists them all; 80 - 120 B of scratch in the fused builds)
template <int M, bool HAS_DIR, bool HAS_Q, int FCM = 0>
__device__ __forceinline__ void assemble_one(const SegRaw<M> &R, int r, int lbit, const SweepScal &s, double &a,
                                             double &b, double &c, double &d)
{
    const unsigned f = R.f(r);
    const bool hl = (f >> lbit) & 1u, hh = (f >> (lbit + 1)) & 1u;
    double co, q;
    if constexpr (FCM == 1) {
        const bool need = axis_exposed(f, lbit);
        co = need ? pack_co<1>(s, nullptr, hl, hh) : 0.0;
        q = (HAS_Q && need) ? pack_q<HAS_Q, 1>(s, nullptr, hl, hh) : 0.0;
    } else {
        co = R.vco[r]; q = R.vq[r];
    }
    assemble_row<HAS_DIR, HAS_Q>(f & 1u, hl, hh, HAS_DIR && R.dir(r), R.vin[r], co, R.vdv[r], q, s, a, b, c, d);
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


// The same for whole tiles (block-uniform precondition: every lane active, every thread owns M rows), buffer
// addressing: scalar row offsets, one per-thread offset, no per-row predicates and no 64-bit address arithmetic.
template <int M, bool HAS_DIR>
__device__ __forceinline__ bool fast_segment_load_buf(const double *__restrict__ in_t, const uint8_t *__restrict__ flags_t,
                                                      const uint8_t *__restrict__ dmask_t, const LineGeom &g, unsigned voff,
                                                      double (&d)[M], unsigned &f0, unsigned &fS, bool &dirS, int &kind,
                                                      int &Lm, uint8_t *strip = nullptr)
{
    const unsigned FULL = 1u | (3u << g.lbit), ROW0 = 1u | (2u << g.lbit);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void *)in_t, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void *)flags_t, 0, 0x7fffffff, 0x00020000);
    const unsigned st = (unsigned)g.stride;
    bool uni = true;
    unsigned inm = 0;
    f0 = 0; fS = 0;
    unsigned fw[M >= 4 ? M / 4 : 1];
    if constexpr (M % 4 == 0 && M <= 16) {
        if (strip != nullptr) load_bytes_packed16<M>(flags_t, voff - (threadIdx.x & 15u), st, (int)(threadIdx.x & 15u), strip, fw);
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        unsigned f;
        if constexpr (M % 4 == 0 && M <= 16) {
            f = (strip != nullptr) ? packed_byte<M>(fw, r) : __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * st, ADI_LOAD_AUX);
        } else {
            f = __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * st, ADI_LOAD_AUX);
        }
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
template <int M, bool HAS_DIR, bool MIXED = true>
__device__ __forceinline__ bool fast_segment_load_fused(const double *__restrict__ in, const uint8_t *__restrict__ flags_t,
                                                        const uint8_t *__restrict__ dmask_t, const LineGeom &g,
                                                        unsigned voff, int r0, int kk, long tbase, const Fuse &fz,
                                                        double (&d)[M], unsigned &f0, unsigned &fS, bool &dirS, int &kind,
                                                        int &Lm, uint8_t *strip = nullptr)
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
    if (strip != nullptr) {
        if constexpr (M % 4 == 0 && M <= 16) {
            unsigned fw[M / 4];
            load_bytes_packed16<M>(flags_t, voff - (threadIdx.x & 15u), (unsigned)g.stride, (int)(threadIdx.x & 15u), strip, fw);
#pragma unroll
            for (int r = 0; r < M; ++r) fb[r] = packed_byte<M>(fw, r);
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
        if (strip == nullptr) fb[r] = __builtin_amdgcn_raw_buffer_load_b8(rF, voff, (unsigned)r * (unsigned)g.stride, ADI_LOAD_AUX);
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
    constexpr int D0 = (M >= 8) ? (MIXED ? ADI_FUSE_D_MIXED : ADI_FUSE_D) : M;
    constexpr int D = D0 < M ? D0 : M;                      // (9 rows per thread: nothing beyond the segment is prefetched)
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
        // the result exists HERE: without this the optimiser merges the final `t + f*(..)` of the two paths and sinks it below
        // the loop, keeping the state row and the Laplacian alive instead of one result (13 VGPRs more at 16 rows)
        asm volatile("" : "+v"(d[r]));
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

// FAST kernel (sparse packs): tiles whose every segment is uniform-interior (see k_sweep_contig_fast)

}  // namespace adi
