// adi_cart_api.hip -- the C-ABI entry points of the Cartesian step (include/adi_hip.h): argument checks, the choice of
// kernel family per axis and line length, and adi_step = adi_step_numba_coeff (adi3d_numba_coeff.py:290-302).
// The kernels live in adi_explicit.hip, adi_sweep_contig.hip, adi_sweep_strided.hip and adi_condense.hip.
#include "adi_cart_host.hpp"

using namespace adi;

extern "C" {

long adi_recommended_plane_stride(int ny, int nz)
{
    // Planes whose byte size is a multiple of 16 KiB alias on the HBM channel interleave when walked with that stride (the
    // axis-0 sweeps: every row of a tile falls on the same channels -- 512^3 dense 2.7 TB/s against 4.0 padded), so the pitch
    // grows by a few hundred bytes.  HOW many depends on the plane: round 1 swept the pad on 2 MiB planes only and took 256
    // elements; round 4 (scripts/pitch_probe.py, fused explicit + axis-0 kernel, Gcell/s) found that very pad to be the WORST
    // choice for 512 KiB planes -- (512, 256, 256): 142 - 149 with 256, 166 - 175 with 16 ... 192 -- and 128 the worst for
    // 256 KiB planes (143 - 164 against 189 with 64), 64 the worst for 128 KiB planes (145 against 170), 1024 / 2048 for 1 / 2 MiB
    // planes (92 - 98 and 87 against 180 - 200).  The table below avoids every measured cliff:
    //     plane < 192 KiB: 256 elements    192 KiB ... < 768 KiB: 64    from 768 KiB: 256 (as before: on 1 and 2 MiB planes 128
    //     and 256 measure the same within the box-to-box scatter, and the headline workload keeps the pitch it was tuned on)
    const long dense = (long)ny * nz, bytes = dense * 8;
    if (bytes % 16384 != 0) return dense;
    return dense + ((bytes >= (192L << 10) && bytes < (768L << 10)) ? 64 : 256);
}

// Physical extents for a (nx, ny, nz) grid: per axis the length -- the logical one or a few multiples of 16 above it --
// that minimises  cells x (1 - w + w / fill)  with `fill` the share of a FAST workgroup's segment slots the lines fill
// (line_fill) and w the axis' share of the step (0.44 / 0.29 / 0.27 at 512^3).  The host allocates fields with these extents,
// marks the extra cells off-mask (identity rows, never read by an in-mask cell) and hands the logical box to the caller:
// a 257^3 grid runs as 272^3 on the FAST kernels instead of 257^3 on the GENERAL ones (0.61 -> 0.28 ms per step).
int adi_recommended_dims(int nx, int ny, int nz, int *px, int *py, int *pz)
{
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && px && py && pz, "adi_recommended_dims: bad argument");
    const int n[3] = {nx, ny, nz};
    int best[3] = {nx, ny, nz};
    const bool lines_fast = nx >= 64 || ny >= 64;                 // a strided FAST kernel could run if nz were a multiple of 16
    static const double w[3] = {0.44, 0.29, 0.27};
    // axis 2 first: the fill of the strided axes depends on nz being a multiple of 16
    for (int axis = 2; axis >= 0; --axis) {
        const int len = n[axis];
        if (len < 64 && !(axis == 2 && len > 16 && lines_fast)) continue;
        const int step = axis == 2 ? 16 : 8;                      // (the FAST strided kernels take 8 rows per thread below 160 rows)
        const int top = len + len / 8 + 16;       // (n/4 + 16 measured: 257^3 as 320^3 takes what it takes as 272^3, 0.42 ms, on 1.6x the memory)
        double best_cost = 0.0;
        for (int cand = len; cand <= top; cand = (cand / step + 1) * step) {
            const double fill = line_fill(axis, cand, axis == 1 ? cand : best[1], axis == 2 ? cand : best[2]);
            double cost = cand * (1.0 - w[axis] + w[axis] / fill);
            if (axis == 2 && cand % 16 != 0 && lines_fast) cost = cand / 0.45;   // ... and costs the other two axes their FAST kernels
            if (best_cost == 0.0 || cost < best_cost * 0.98) { best_cost = cost; best[axis] = cand; }   // 2 %: ties go to less memory
        }
    }
    *px = best[0]; *py = best[1]; *pz = best[2];
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
// h_face_consts of the sweep entry points: (c-, c+, q-, q+) of the sweep axis for packs built from per-face scalars
// (adi_face_constants); honoured only together with sparse reads -- stale or hand-built packs are read from their arrays
static void set_face_consts(SweepScal &s, const double *fcs)
{
    if (fcs != nullptr && s.sparse) {
        s.fconst = 1;
        for (int i = 0; i < 4; ++i) s.fc[i] = fcs[i];
    }
}

static int sweep_entry(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                       const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
                       long plane_stride, int sparse, double theta, double gam, double dt, double Tinf, double *d_out,
                       const double *d_xlo, const double *d_xhi, void *d_work, size_t work_bytes, void *stream,
                       const Fuse *fz, const double *fcs, const double *c_lo = nullptr, const double *c_hi = nullptr,
                       const double *c_w = nullptr)
{
    ADI_REQUIRE(axis >= 0 && axis < 3, "adi_sweep: bad axis %d", axis);
    bool has_dir, has_q;
    if (int rc = variant_flags(variant, &has_dir, &has_q)) return rc;
    ADI_REQUIRE(d_in && d_flags && d_coeff && d_out, "adi_sweep: null argument");
#ifndef ADI_ALLOW_INPLACE
    ADI_REQUIRE(d_in != d_out, "adi_sweep: output aliases input");
#endif
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
    s.nofb = (sparse & 4) ? 1 : 0;
    if (s.tg < kMixedMinTg) s.sparse = 0;   // a vanishing time step: GENERAL kernels only (mixed_condense's recurrence would overflow)
    s.nt = store_policy_nt(L.nx, L.sx);
    set_face_consts(s, fcs);
    if (c_w != nullptr && (c_lo != nullptr || c_hi != nullptr)) {
        // deferred interface correction (adi_sweep_corrected): the strided kernels of memory axis 1 add it to what they load
        ADI_REQUIRE(axis == 1 && fz == nullptr && !d_xlo && !d_xhi, "adi_sweep_corrected: axis 1 sweeps only");
        ADI_REQUIRE((long)ny * nz * 8 < 0x7fffffffL, "adi_sweep_corrected: plane of %ld cells is too large", (long)ny * nz);
        s.c_lo = c_lo; s.c_hi = c_hi; s.c_w = c_w; s.c_n = nx; s.c_bytes = (unsigned)((long)ny * nz * 8);
    }
    hipStream_t st = as_stream(stream);
    SweepArgs a;
    a.in = d_in; a.flags = d_flags; a.coeff = d_coeff;
    a.dmask = has_dir ? d_dir_mask : nullptr; a.dval = has_dir ? d_dir_val : nullptr; a.qf = has_q ? d_qflux : nullptr;
    long inner_stride;
    const LineGeom g = line_geom(axis, L, &inner_stride);
    const int n = g.n;
    if (fz != nullptr && (axis != 0 || n > kMaxFastLine))
        return set_err(ADI_ERR_UNSUPPORTED, "fused explicit + sweep: axis 0 with at most %d planes only", kMaxFastLine);
    if (n > kMaxFastLine || (axis == 2 && (d_xlo || d_xhi))) {
        // lines too long for the in-register partition kernels: thread-per-line Thomas with c', d' in HBM scratch
        const size_t need = (size_t)2 * L.nx * L.sx * sizeof(double);
        if (d_work == nullptr || work_bytes < need)
            return set_err(ADI_ERR_ARG, "adi_sweep: this sweep (line length %d) needs a workspace of %zu bytes", n, need);
        double *wc = (double *)d_work, *wd = wc + (size_t)L.nx * L.sx;
        generic_sweep(has_dir, has_q, a, g, inner_stride, s, d_out, d_xlo, d_xhi, wc, wd, st);
    } else if (axis == 2) {
        contig_sweep(has_dir, has_q, a, L, s, d_out, d_work, work_bytes, st);
    } else {
        strided_sweep(has_dir, has_q, a, L, g, s, d_out, d_xlo, d_xhi, d_work, work_bytes, st, fz);
    }
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_sweep(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
              const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
              long plane_stride, int sparse, double theta, double gam, double dt, double Tinf, double *d_out,
              const double *d_xlo, const double *d_xhi, const double *h_face_consts, void *d_work, size_t work_bytes,
              void *stream)
{
    return sweep_entry(axis, variant, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                       sparse, theta, gam, dt, Tinf, d_out, d_xlo, d_xhi, d_work, work_bytes, stream, nullptr,
                       h_face_consts);
}

int adi_sweep_corrected(int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                        const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny, int nz,
                        long plane_stride, int sparse, double theta, double gam, double dt, double Tinf, double *d_out,
                        const double *d_ulo, const double *d_uhi, const double *d_w, const double *h_face_consts, void *d_work,
                        size_t work_bytes, void *stream)
{
    ADI_REQUIRE(d_w != nullptr || (d_ulo == nullptr && d_uhi == nullptr), "adi_sweep_corrected: interface values without weights");
    return sweep_entry(1, variant, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                       sparse, theta, gam, dt, Tinf, d_out, nullptr, nullptr, d_work, work_bytes, stream, nullptr,
                       h_face_consts, d_ulo, d_uhi, d_w);
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
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
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
                        const double *d_xhi, const double *h_face_consts, void *d_work, size_t work_bytes, void *stream)
{
    ADI_REQUIRE(valid_lo <= 0 && valid_hi >= (long)(nx - 1) * (plane_stride ? plane_stride : (long)ny * nz) + (long)ny * nz,
                "adi_explicit_sweep0: the readable range [%ld, %ld) does not cover the box", valid_lo, valid_hi);
    const Fuse fz = make_fuse(nx, ny, nz, plane_stride, dx, dt, kappa, theta, valid_lo, valid_hi);
    const double gam = kappa * dt / (dx * dx);   // adi3d_numba_coeff.py:292
    return sweep_entry(0, variant, d_T, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                       sparse, theta, gam, dt, Tinf, d_out, d_xlo, d_xhi, d_work, work_bytes, stream, &fz, h_face_consts);
}
static int condense_entry(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                          const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny,
                          int nz, long plane_stride, int sparse, double theta, double gam, double dt, double Tinf,
                          double *d_cond, void *d_work, size_t work_bytes, void *stream, const Fuse *fz, const double *fcs)
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
    s.nofb = (sparse & 4) ? 1 : 0;
    if (s.tg < kMixedMinTg) s.sparse = 0;   // (as in sweep_entry)
    s.nt = store_policy_nt(L.nx, L.sx);
    set_face_consts(s, fcs);
    hipStream_t st = as_stream(stream);
    SweepArgs a;
    a.in = d_in; a.flags = d_flags; a.coeff = d_coeff;
    a.dmask = has_dir ? d_dir_mask : nullptr; a.dval = has_dir ? d_dir_val : nullptr; a.qf = has_q ? d_qflux : nullptr;
    const int rc = condense_sweep(has_dir, has_q, axis, a, L, s, d_cond, d_work, work_bytes, st, fz);
    if (rc != ADI_OK) return rc;
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}
int adi_sweep_condense(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                       const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux, int nx, int ny,
                       int nz, long plane_stride, int sparse, double theta, double gam, double dt, double Tinf,
                       double *d_cond, const double *h_face_consts, void *d_work, size_t work_bytes, void *stream)
{
    return condense_entry(axis, variant, d_in, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                          sparse, theta, gam, dt, Tinf, d_cond, d_work, work_bytes, stream, nullptr, h_face_consts);
}

int adi_explicit_condense0(int variant, const double *d_T, long valid_lo, long valid_hi, const uint8_t *d_flags,
                           const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                           const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse, double dx,
                           double dt, double kappa, double theta, double Tinf, double *d_cond, double *d_R0_out,
                           const double *h_face_consts, void *d_work, size_t work_bytes, void *stream)
{
    ADI_REQUIRE(valid_lo <= 0 && valid_hi >= (long)(nx - 1) * (plane_stride ? plane_stride : (long)ny * nz) + (long)ny * nz,
                "adi_explicit_condense0: the readable range [%ld, %ld) does not cover the box", valid_lo, valid_hi);
    ADI_REQUIRE(d_R0_out != d_T, "adi_explicit_condense0: R0 output aliases the state");
    Fuse fz = make_fuse(nx, ny, nz, plane_stride, dx, dt, kappa, theta, valid_lo, valid_hi);
    fz.r0_out = d_R0_out;
    const double gam = kappa * dt / (dx * dx);
    return condense_entry(0, variant, d_T, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, nx, ny, nz, plane_stride,
                          sparse, theta, gam, dt, Tinf, d_cond, d_work, work_bytes, stream, &fz, h_face_consts);
}
}  // extern "C"

// h_queued (optional, 3 host words): the first word of the workspace after each sweep = the number of units its FAST kernel
// queued (copied on the stream; valid once the stream has been synchronised)
static int step_impl(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b, const uint8_t *d_flags,
                     const double *const *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                     const double *const *d_qflux, int variant, int sparse, int nx, int ny, int nz, long plane_stride,
                     double dx, double rho, double cp, double k, double dt, double theta, double Tinf,
                     const double *h_face_consts, void *d_work, size_t work_bytes, void *stream, unsigned *h_queued)
{
    ADI_REQUIRE(d_T_in && d_T_out && d_tmp_a && d_tmp_b && d_coeff, "adi_step: null argument");
    const double *fc0 = h_face_consts, *fc1 = h_face_consts ? h_face_consts + 4 : nullptr, *fc2 = h_face_consts ? h_face_consts + 8 : nullptr;
    ADI_REQUIRE(d_tmp_a != d_tmp_b && d_tmp_a != d_T_in && d_tmp_b != d_T_in && d_T_out != d_tmp_a && d_T_out != d_T_in,
                "adi_step: buffers must be distinct (T_out may equal tmp_b only)");
    // kappa, gam: adi3d_numba_coeff.py:292
    const double kappa = k / (rho * cp);
    const double gam = kappa * dt / (dx * dx);
    const double *q0 = d_qflux ? d_qflux[0] : nullptr, *q1 = d_qflux ? d_qflux[1] : nullptr, *q2 = d_qflux ? d_qflux[2] : nullptr;
    int rc;
    // h_queued: a sweep that runs no FAST kernel (short lines, dense packs, a fused sweep the FAST kernel declines) never
    // touches the queue word, and for lines beyond kMaxFastLine the workspace holds c' / d' doubles: the word is zeroed
    // on the stream before every sweep and the long-line case reports 0 without reading it
    const int nn[3] = {nx, ny, nz};
    auto arm = [&](int axis) {
        if (h_queued != nullptr && nn[axis] <= kMaxFastLine)
            (void)hipMemsetAsync(d_work, 0, sizeof(unsigned), as_stream(stream));
    };
    auto report = [&](int axis) {
        if (h_queued == nullptr) return;
        if (nn[axis] > kMaxFastLine) { h_queued[axis] = 0u; return; }
        (void)hipMemcpyAsync(h_queued + axis, d_work, sizeof(unsigned), hipMemcpyDeviceToHost, as_stream(stream));
    };
    arm(0);
    if (adi_explicit_fused_supported(nx, ny, nz, plane_stride, 0)) {
        // stages 1+2 in one pass: R0 is evaluated inside the loads of the axis-0 sweep
        const long sxe = plane_stride ? plane_stride : (long)ny * nz;
        rc = adi_explicit_sweep0(variant, d_T_in, 0, (long)(nx - 1) * sxe + (long)ny * nz, d_flags, d_coeff[0], d_dir_mask,
                                 d_dir_val, q0, nx, ny, nz, plane_stride, sparse, dx, dt, kappa, theta, Tinf, d_tmp_b,
                                 nullptr, nullptr, fc0, d_work, work_bytes, stream);
    } else {
        rc = adi_explicit_rhs(d_T_in, d_flags, nx, ny, nz, plane_stride, dx, dt, kappa, theta, d_tmp_a, stream);
        if (rc) return rc;
        rc = adi_sweep(0, variant, d_tmp_a, d_flags, d_coeff[0], d_dir_mask, d_dir_val, q0, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_tmp_b, nullptr, nullptr, fc0, d_work, work_bytes, stream);
    }
    if (rc) return rc;
    report(0);
    arm(1);
    rc = adi_sweep(1, variant, d_tmp_b, d_flags, d_coeff[1], d_dir_mask, d_dir_val, q1, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_tmp_a, nullptr, nullptr, fc1, d_work, work_bytes, stream);
    if (rc) return rc;
    report(1);
    arm(2);
    rc = adi_sweep(2, variant, d_tmp_a, d_flags, d_coeff[2], d_dir_mask, d_dir_val, q2, nx, ny, nz, plane_stride, sparse, theta, gam, dt, Tinf, d_T_out, nullptr, nullptr, fc2, d_work, work_bytes, stream);
    if (rc) return rc;
    report(2);
    return ADI_OK;
}

extern "C" {

int adi_step(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b, const uint8_t *d_flags,
             const double *const *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
             const double *const *d_qflux, int variant, int sparse, int nx, int ny, int nz, long plane_stride,
             double dx, double rho, double cp, double k, double dt, double theta, double Tinf,
             const double *h_face_consts, void *d_work, size_t work_bytes, void *stream)
{
    return step_impl(d_T_in, d_T_out, d_tmp_a, d_tmp_b, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, variant, sparse, nx, ny,
                     nz, plane_stride, dx, rho, cp, k, dt, theta, Tinf, h_face_consts, d_work, work_bytes, stream, nullptr);
}

int adi_step_queued(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b, const uint8_t *d_flags,
                    const double *const *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                    const double *const *d_qflux, int variant, int sparse, int nx, int ny, int nz, long plane_stride,
                    double dx, double rho, double cp, double k, double dt, double theta, double Tinf,
                    const double *h_face_consts, void *d_work, size_t work_bytes, void *stream, unsigned *h_queued)
{
    ADI_REQUIRE(h_queued && d_work && work_bytes >= sizeof(unsigned), "adi_step_queued: needs a workspace and three host words");
    ADI_REQUIRE((sparse & 4) == 0, "adi_step_queued: the no-fallback promise skips the queue it is asked to report");
    h_queued[0] = h_queued[1] = h_queued[2] = 0xffffffffu;
    return step_impl(d_T_in, d_T_out, d_tmp_a, d_tmp_b, d_flags, d_coeff, d_dir_mask, d_dir_val, d_qflux, variant, sparse, nx, ny,
                     nz, plane_stride, dx, rho, cp, k, dt, theta, Tinf, h_face_consts, d_work, work_bytes, stream, h_queued);
}
}  // extern "C"
