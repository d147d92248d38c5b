/*
 * include/adi_hip.h -- C ABI of libadi_hip.so, the MI355X (gfx950) ADI heat-equation hot path.
 *
 * The reference (Matemusi/ADI_thermal_fields) has no FFI: its boundary is a Python module
 * namespace (`import adi3d_numba_coeff as adi` / `import adi3d_gpu_coeff as adi`,
 * waam_from_stl_v7_mm.py:321-335).  This header is what a binding for that namespace calls;
 * each entry point cites the reference function it replaces.  adi_thermal_fields_amd/ binds
 * it with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, every call returns int (ADI_OK == 0) and never throws.
 *   - adi_last_error() returns a thread-local, human-readable message for the last failure.
 *   - Fields are C-order (n0, n1, n2) fp64 arrays; masks are 1 byte per cell (0 / non-zero)
 *     (adi3d_numba_coeff.py:14-19, :29-36).  Axis 2 is contiguous.
 *   - "d_" pointers are DEVICE pointers owned by the caller (hipMalloc / torch); "h_" pointers are
 *     host pointers borrowed for the duration of the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Stateless entry points
 *     only enqueue work; they never synchronise.
 *   - Face order everywhere: 0 'x-', 1 'x+', 2 'y-', 3 'y+', 4 'z-', 5 'z+' (axis = face / 2).
 */
#ifndef ADI_HIP_H
#define ADI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADI_OK              0
#define ADI_ERR_ARG         1   /* bad argument (the Python host maps it to ValueError / AssertionError) */
#define ADI_ERR_HIP         2   /* a HIP runtime call failed */
#define ADI_ERR_UNSUPPORTED 3   /* valid request this build cannot serve */
#define ADI_ERR_STATE       4   /* context used out of order (e.g. step before build_coeffs) */

#define ADI_ABI_VERSION 18

/* per-face BC data mode for adi_build_coeffs */
#define ADI_FACE_NONE   0   /* face carries no data (robin_h is None / face absent from `neumann`) */
#define ADI_FACE_SCALAR 1
#define ADI_FACE_FIELD  2   /* per-voxel fp64 array */

/* z-end boundary kinds of the cylindrical path (adi3d_cyl_phi_v3.py:60-68, :271-296) */
#define ADI_ZBC_NEUMANN0  0
#define ADI_ZBC_DIRICHLET 1
#define ADI_ZBC_ROBIN     2

/* sweep kernel variants (which pack arrays a sweep reads; selects the byte count of the roofline) */
#define ADI_SWEEP_GENERAL   0   /* in, flags, coeff, dir_mask, dir_val, qflux -> out : 42 B/cell */
#define ADI_SWEEP_NO_DIR    1   /* no Dirichlet cells: in, mask, coeff, qflux -> out : 33 B/cell */
#define ADI_SWEEP_NO_Q      2   /* no Neumann flux:    in, mask, coeff, dir_mask, dir_val -> out : 34 B/cell */
#define ADI_SWEEP_LEAN      3   /* neither:            in, mask, coeff -> out : 25 B/cell */

int         adi_abi_version(void);
/* identity of this build: 16 hex digits of the SHA-256 over the library's sources and compile flags (independent of the
 * directory it was built in); profiles/pmc_traffic.json is stamped with it and bench.py quotes measured HBM traffic only
 * for a library that reports the same stamp.  "unstamped" for a library not built by adi_thermal_fields_amd/build.py. */
const char *adi_build_stamp(void);
const char *adi_last_error(void);
int         adi_device_count(int *count);
/* name: caller buffer of >= 256 bytes; cu_count / hbm_bytes / lds_per_cu may be NULL */
int         adi_device_info(int device, char *name, int *cu_count, size_t *hbm_bytes, size_t *lds_per_cu);
/* The constructors of the reference copy their array arguments and the step returns a new array
 * (adi3d_numba_coeff.py:18, :31-36, :135, :302): with HOST arrays at the boundary that is one transfer of every plane
 * in each direction.  nplanes planes of plane_bytes each, src_pitch / dst_pitch bytes apart (the device layout pads its
 * planes, the host array is dense), as ONE 2-D DMA on `stream`; to_device: 1 = host -> device, 0 = device -> host. */
int         adi_copy_planes(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t plane_bytes,
                            size_t nplanes, int to_device, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stateless device entry points (caller-owned device memory).
 *
 * Device layout of every field and mask: element (i, j, k) at i*plane_stride + j*nz + k, with
 * plane_stride >= ny*nz in ELEMENTS (0 means dense, ny*nz).  A padded plane stride keeps the rows of an
 * axis-0 line off the same HBM channels; adi_recommended_plane_stride() returns the pitch this library
 * wants for a given (ny, nz).  Per-line arrays (d_xlo, d_xhi, d_cond) are dense.
 * ---------------------------------------------------------------------------------------------- */
long adi_recommended_plane_stride(int ny, int nz);

/* Physical extents the library recommends for a logical (nx, ny, nz) grid (ABI v16): each >= the logical extent, chosen
 * per axis among the logical length and the next few multiples of 16 so that the lines fill the FAST kernels' workgroups
 * (a 257-row line runs on the GENERAL kernels, a 272-row one on the FAST ones).  The caller allocates fields with the
 * physical extents (plane stride from adi_recommended_plane_stride(*py, *pz)), marks every cell outside the logical box
 * off-mask -- off-mask cells are identity rows and are never read by an in-mask cell, so the results inside the logical box
 * are those of the logical grid (the reference treats the edge of the domain and an off-mask neighbour alike,
 * adi3d_numba_coeff.py:38-55) -- and passes the physical extents to every entry point.  adi3d_hip_coeff.Layout does this. */
int adi_recommended_dims(int nx, int ny, int nz, int *px, int *py, int *pz);

/* exposed_mask(mask, face): adi3d_numba_coeff.py:38-55 / adi3d_gpu_coeff.py:31-48 */
int adi_exposed_mask(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, int face,
                     uint8_t *d_exposed, void *stream);

/* Exposed faces per plane k of axis 2, counted from the neighbour-flags digest (adi_build_nbr_flags): the count loop of
 * quick_compare_layer_birth_robin_v3.py:97-108 (exposed x-/x+/y-/y+ faces of the 2-D section, the digital perimeter
 * divided by dx) for every layer of a 3-D mask at once.  face_bits: bit f set = count face f (0..5 = x-, x+, y-, y+, z-,
 * z+; 15 = the four lateral faces).  d_counts: nz unsigned 64-bit integers (zeroed here). */
int adi_count_exposed_faces(const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride, int face_bits,
                            unsigned long long *d_counts, void *stream);

/* A layer birth in one pass over the planes [k_begin, k_end) of axis 2 (activate_layer, waam_from_stl_v7_mm.py:487-495):
 * newborn = full & ~active; T[newborn] = Ts; active |= full.  *d_newborn (device, 64-bit) receives the number of
 * newborn cells.  The caller then rebuilds the flags and the packs, as the reference's drivers do (:494-495, :534). */
int adi_birth_planes(double *d_T, uint8_t *d_active, const uint8_t *d_full, int nx, int ny, int nz, long plane_stride,
                     int k_begin, int k_end, double Ts, unsigned long long *d_newborn, void *stream);

/*
 * precompute_coeff_packs_unified: adi3d_numba_coeff.py:57-118 / adi3d_gpu_coeff.py:50-110.
 * One pass over the grid writes the three Robin coefficient fields and the three Neumann flux fields:
 *   coeff[axis] += h_f * dx^2 / (rho cp dx^3) on cells exposed on face f ('-' face first, then '+'),
 *   qflux[axis] += q_f * dx^2 / (rho cp dx^3) on exposed cells.
 * h_mode/q_mode[6]: ADI_FACE_*; h_scalar/q_scalar[6]; h_field/q_field[6]: device pointers or NULL.
 * d_coeff[3], d_qflux[3]: device output fields (fully overwritten).
 */
int adi_build_coeffs(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, double dx,
                     double rho, double cp,
                     const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                     const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                     double *const *d_coeff, double *const *d_qflux, void *stream);
/* ... restricted to the planes [k_begin, k_end) of axis 2 (in-place update of packs built earlier: see
 * adi_build_nbr_flags_planes). */
int adi_build_coeffs_planes(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, double dx, double rho,
                            double cp, const int *h_mode, const double *h_scalar, const double *const *d_h_field,
                            const int *q_mode, const double *q_scalar, const double *const *d_q_field,
                            double *const *d_coeff, double *const *d_qflux, int k_begin, int k_end, void *stream);

/*
 * Neighbour flags, the device-side digest of the mask every step kernel reads instead of the raw mask:
 *   bit0 = cell in mask; bit(1+2a) / bit(2+2a) = the minus / plus neighbour along axis a exists and is in
 *   the mask (the `mask[i-1,j,k]` / `mask[i+1,j,k]` tests of adi3d_numba_coeff.py:150-153, :246-251).
 * Rebuild whenever the mask changes (same moment the packs are rebuilt, waam_from_stl_v7_mm.py:494-495, :534).
 * For a slab of a larger grid, build the flags on the slab plus its two halo planes and pass the interior.
 */
int adi_build_nbr_flags(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride,
                        uint8_t *d_flags, void *stream);
/* The same restricted to the planes [k_begin, k_end) of axis 2: after a layer birth only the planes of the layer and
 * the one below / above it change (waam_from_stl_v7_mm.py:487-495 rebuilds everything; the result is the same). */
int adi_build_nbr_flags_planes(const uint8_t *d_mask, int nx, int ny, int nz, long plane_stride, uint8_t *d_flags,
                               int k_begin, int k_end, void *stream);

/* lap1D_x/y/z + R0 = Tn + dt*kappa*(1-theta)*(Lx+Ly+Lz): adi3d_numba_coeff.py:240-288, :298 */
int adi_explicit_rhs(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride,
                     double dx, double dt, double kappa, double theta, double *d_R0, void *stream);
/* the same for planes [i_begin, i_end) of the array only (neighbour planes are still read where they exist): lets a
 * slab compute its interior planes while the halo planes are in flight */
int adi_explicit_rhs_planes(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride,
                            double dx, double dt, double kappa, double theta, double *d_R0,
                            int i_begin, int i_end, void *stream);

/*
 * sweep_axis0/1/2: adi3d_numba_coeff.py:133-237 (full-length identity-row form of
 * adi3d_gpu_coeff.py:154-191).  One independent tridiagonal system per grid line along `axis`.
 * variant: ADI_SWEEP_*; arrays a variant does not read may be NULL.  d_out must not alias d_in.
 * d_xlo / d_xhi (optional, dense, one value per line): value of the unknown just before the first / after
 * the last local row when the line continues on a neighbouring GPU (see adi_interface_solve); the coupling
 * itself is read from the halo bits of d_flags.  NULL for a stand-alone grid.
 * sparse != 0 asserts the invariant of packs built by adi_build_coeffs -- coeff/qflux of this axis are zero
 * except on cells that lack an in-mask neighbour along the axis -- so the kernel loads them only there
 * (and dir_val only where dir_mask is set).  Pass 0 for hand-built packs.  Bit 1 (value 2) is an optional hint: every
 * cell of the box is in the mask (an all-solid box), which lets the strided FAST kernels (fused and unfused) run their leaner build; results do not depend on it.
 * Bit 2 (value 4) is a PROMISE, not a hint: the FAST kernel takes every unit of this sweep, so the queue reset and the
 * GENERAL launch behind it are skipped (units the FAST kernel cannot take would be left unwritten).  Which units are queued
 * depends on the flags, the Dirichlet mask, the variant, `sparse` and the shape only -- never on the field -- so a caller
 * may set the bit after it has seen the queue of an identical call come back empty: after a sweep without the bit the first
 * 32-bit word of d_work is the number of queued units (adi3d_hip_coeff.py does exactly that, once per mask / pack version).
 * d_work/work_bytes (adi_sweep_workspace_bytes): c'/d' scratch for lines longer than the in-register limit,
 * otherwise the unit queue that lets a sparse sweep run as a FAST kernel (solid interior) followed by the
 * GENERAL kernel on the queued surface units; with NULL/0 the GENERAL kernel processes everything.

 * h_face_consts (ABI v14, optional, HOST pointer to 4 doubles; NULL: read the arrays): packs built from per-face SCALARS
 * (adi_face_constants) -- (c-, c+, q-, q+) = h_f dx^2 / (rho cp dx^3) and q_f dx^2 / (rho cp dx^3) of the minus / plus face
 * of the sweep axis.  The coefficient of a cell exposed along the axis is then (0 + [no minus neighbour] c-) + [no plus
 * neighbour] c+ -- the accumulation order of the reference (:93-99), hence the value stored in the array bit for bit -- and
 * follows from the flags byte alone: the kernels do not load coeff / qflux at all (on a curved solid those loads can only be
 * issued after the flags have arrived).  Honoured only together with `sparse` (stale / hand-built packs are read).
 */
int adi_sweep(int axis, int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
              const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux,
              int nx, int ny, int nz, long plane_stride, int sparse,
              double theta, double gam, double dt, double Tinf,
              double *d_out, const double *d_xlo, const double *d_xhi, const double *h_face_consts,
              void *d_work, size_t work_bytes, void *stream);
/* (c-, c+, q-, q+) per axis for adi_sweep & co.: h_consts[12] = [axis][4], h_valid[3] = 1 where both faces of the axis carry
 * scalars or nothing (ADI_FACE_SCALAR / ADI_FACE_NONE) for h AND q -- only then may h_consts + 4*axis be passed on.
 * Host arithmetic, the very expressions of adi_build_coeffs. */
int adi_face_constants(double dx, double rho, double cp, const int *h_mode, const double *h_scalar, const int *q_mode,
                       const double *q_scalar, double *h_consts, int *h_valid);
int adi_sweep_workspace_bytes(int axis, int nx, int ny, int nz, long plane_stride, size_t *bytes);

/*
 * Slab decomposition of a sweep (no counterpart in the reference, which is single-process): pass A.
 * For every line, the local rows are condensed to six numbers (d_cond: [6][nlines], dense) such that
 *   x_first = c0 - c1*xl - c2*xr,   x_last = c3 - c4*xl - c5*xr
 * with xl / xr the unknowns adjacent to the slab on the neighbouring GPUs.  Same inputs as adi_sweep.
 */
int adi_sweep_condense(int axis, int variant, const double *d_in, const uint8_t *d_flags,
                       const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                       const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse,
                       double theta, double gam, double dt, double Tinf, double *d_cond, const double *h_face_consts,
                       void *d_work, size_t work_bytes, void *stream);
/*
 * Explicit stage folded into the axis-0 sweep (ABI v7): lap1D_x/y/z + R0 (adi3d_numba_coeff.py:240-288, :292-298)
 * are evaluated inside the loads of sweep_axis0 (:133-166, :299), so R0 never travels through HBM -- the first two
 * stages of adi_step_numba_coeff cost one read of T and one write of U.  Same arithmetic in the same order as
 * adi_explicit_rhs followed by adi_sweep(axis 0): the result is bit-identical to running the two stages.
 * d_T: the state (the `Tn` of the step), box (nx, ny, nz) with plane_stride; neighbours outside the box are read
 * where the flags byte says they exist (halo planes of a slab, rows next to a sub-box of lines), so the caller states
 * which element offsets relative to d_T may be read: [valid_lo, valid_hi) must cover the box and lie inside the
 * allocation (whole buffer: valid_lo = -(offset of d_T in it), valid_hi = its length - that offset).
 * adi_explicit_fused_supported(pass): 1 when pass 0 (sweep) / pass 1 (condensation) can run fused on this box
 * (nx <= 1024 planes; pass 1: whole register segments); otherwise run adi_explicit_rhs + adi_sweep.
 * adi_explicit_condense0: d_R0_out (optional, same box layout as d_T) also receives R0, so that pass B can be the plain
 * adi_sweep on it instead of evaluating the explicit stage a second time.
 */
int adi_explicit_fused_supported(int nx, int ny, int nz, long plane_stride, int pass);
int adi_explicit_sweep0(int variant, const double *d_T, long valid_lo, long valid_hi, const uint8_t *d_flags,
                        const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                        const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse,
                        double dx, double dt, double kappa, double theta, double Tinf,
                        double *d_out, const double *d_xlo, const double *d_xhi, const double *h_face_consts,
                        void *d_work, size_t work_bytes, void *stream);
int adi_explicit_condense0(int variant, const double *d_T, long valid_lo, long valid_hi, const uint8_t *d_flags,
                           const double *d_coeff, const uint8_t *d_dir_mask, const double *d_dir_val,
                           const double *d_qflux, int nx, int ny, int nz, long plane_stride, int sparse,
                           double dx, double dt, double kappa, double theta, double Tinf,
                           double *d_cond, double *d_R0_out, const double *h_face_consts,
                           void *d_work, size_t work_bytes, void *stream);
/*
 * Pass A folded into the explicit stage (ABI v7): for a line whose rows are uniform along axis 0 (solid interior; at
 * most one end row differs) the six pass-A numbers follow from two dot products of R0 with fixed weights -- the first
 * column u of tridiag(-tg, 1+2tg, -tg)^-1 and its reverse -- which the marching explicit kernel accumulates while it
 * writes R0, so the slab's inputs are read once.  Lines that are not uniform are condensed from the stored R0.
 *   adi_axis0_dots_setup      u for (n, theta, gam) -> d_weights[n]                       (once per dt; synchronises)
 *   adi_axis0_classify        d_cls[ny*nz] (1 = uniform) and d_list (count + ids of the others)   (once per mask)
 *   adi_explicit_rhs_dots     adi_explicit_rhs_planes + partial dot products d_part (adi_axis0_dots_workspace).  The lines
 *                             are planes [i_org, i_org + n_line) of the array; a call may cover part of them in whole
 *                             chunks of adi_axis0_dots_ichunk(n_line) planes (interior first, halo-adjacent planes later)
 *   adi_axis0_dots_finish     -> d_cond [6][line_end - line_begin] of the lines [line_begin, line_end), the format of
 *                             adi_sweep_condense; arrays are those of the box (nx = i_end - i_begin planes)
 */
int adi_axis0_dots_supported(int nx, int ny, int nz, long plane_stride);
int adi_axis0_dots_workspace(int nx, int ny, int nz, size_t *part_bytes, size_t *list_bytes);
int adi_axis0_dots_setup(int n, double theta, double gam, double *d_weights, void *stream);
int adi_axis0_classify(const uint8_t *d_flags, const uint8_t *d_dir_mask, int nx, int ny, int nz, long plane_stride,
                       uint8_t *d_cls, unsigned *d_list, void *stream);
int adi_axis0_dots_ichunk(int n_line);   /* planes per chunk of partial sums for lines of n_line rows */
int adi_explicit_rhs_dots(const double *d_T, const uint8_t *d_flags, int nx, int ny, int nz, long plane_stride,
                          double dx, double dt, double kappa, double theta, double *d_R0, int i_begin, int i_end,
                          int i_org, int n_line, const double *d_weights, double *d_part, void *stream);
int adi_axis0_dots_finish(int variant, const double *d_part, const double *d_weights, const uint8_t *d_cls,
                          const unsigned *d_list, const double *d_R0, const uint8_t *d_flags, const double *d_coeff,
                          const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux,
                          int nx, int ny, int nz, long plane_stride, double theta, double gam, double dt, double Tinf,
                          long line_begin, long line_end, double *d_cond, void *stream);
/* d_cond_all: [nranks][6][nlines], the all-gathered pass-A output ordered by slab.  Solves the reduced
 * interface system of every line and writes this rank's boundary values for pass B (adi_sweep). */
int adi_interface_solve(const double *d_cond_all, int nranks, int rank, long nlines,
                        double *d_xlo, double *d_xhi, void *stream);
/* Neighbour-only interface solve, for slabs thick enough that the coupling across a boundary window has decayed
 * below rounding (the caller checks |aL| of its last-rows window and |cF| of its first-rows window).
 * d_my_lo / d_my_hi: [6][nlines] pass-A output of this slab's first / last window of planes;
 * d_prev_hi: rows 3..5 (gL, aL, cL) of the last window of the slab below, d_next_lo: rows 0..1 (gF, aF) of the
 * first window of the slab above; NULL where there is no neighbour (boundary value 0, never used). */
int adi_interface_pair(const double *d_my_lo, const double *d_my_hi, const double *d_prev_hi, const double *d_next_lo,
                       long nlines, double *d_xlo, double *d_xhi, void *stream);

/*
 * Deferred form of the slab decomposition (ABI v12; no counterpart in the reference, which is single-process).  For slabs
 * whose sharded-axis lines are all uniform (solid, no Dirichlet cell) the two-pass scheme above is not needed:
 *   1. every rank runs the ordinary single-domain kernel (adi_explicit_sweep0 / adi_sweep along axis 0, d_xlo = d_xhi =
 *      NULL) on its slab: each line solved with zero boundary values -> x0.  Planes 0 and nx-1 of x0 are exactly the
 *      pass-A right-hand sides (gF, gL), so "one ghost plane per sweep" travels to each neighbour and nothing else;
 *   2. adi_interface_deferred: the 2 x 2 system per boundary and line -> the two interface values of every line;
 *   3. by linearity  x = x0 + ulo * w[i] + uhi * w[nx-1-i]  with w the decaying homogeneous solution of the uniform
 *      row (adi_axis0_deferred_setup).  That rank-two update is not applied to x0 in memory: adi_sweep_corrected, the
 *      axis-1 sweep that consumes the field next (sweep_axis1, adi3d_numba_coeff.py:300), adds it to every value it
 *      loads -- two planes re-read from L2 instead of a second pass over the slab.
 * adi_axis0_deferred_setup: w (nx doubles, device) for (theta, gam); entries <= tol are stored as exactly 0 (the kernels
 *   skip them), *h_reach = number of non-zero entries, *h_omega = w[0].  The form is valid when the weights have decayed
 *   across the slab (*h_reach < nx): then neither the far end's closure nor the next-but-one slab matters.  Synchronises.
 * adi_interface_deferred: d_first / d_last = planes 0 / nx-1 of this rank's x0 (dense ny*nz), d_prev_last / d_next_first
 *   = the neighbours' adjacent planes (NULL: no neighbour) -> d_ulo, d_uhi (dense ny*nz; 0 where there is no neighbour).
 * adi_sweep_corrected: adi_sweep(axis = 1, ...) reading  in + w[i] * ulo[j][k] + w[nx-1-i] * uhi[j][k]  (d_ulo or d_uhi
 *   NULL: that term is absent; d_w NULL: plain adi_sweep).
 */
int adi_axis0_deferred_setup(int n, double theta, double gam, double tol, double *d_w, double *h_omega, int *h_reach,
                             void *stream);
int adi_interface_deferred(const double *d_first, const double *d_last, const double *d_prev_last,
                           const double *d_next_first, double omega, long nlines, double *d_ulo, double *d_uhi,
                           void *stream);
/* The deferred form WITHOUT decay (ABI v15: thin slabs / strong scaling, e.g. 512^3 over 8 GPUs = 64 planes at cfl 200).
 * x = x0 + c_lo w[i] + c_hi w[n-1-i] still holds (adi_axis0_deferred_setup with tol = 0: no weight is cut); the interface
 * system now couples all ranks, as in the two-pass `exact` form, but only its right-hand sides change from step to step:
 *   per step   all-gather [2][nlines] = (plane 0, plane nx-1 of x0) -> d_g_all [nranks][2][nlines]   (a third of the payload
 *              of the six-number block), adi_interface_solve_uniform -> (d_xlo, d_xhi), adi_deferred_exact_coef -> the
 *              per-line coefficients (d_clo, d_chi) of w[i] and w[nx-1-i] for adi_sweep_corrected;
 *   per plan   adi_deferred_exact_setup: d_mat [4][nlines] = (aF, cF, aL, cL) of this rank and d_kap [2][nlines], the
 *              Sherman-Morrison factors of a global end row (first / last rank: the line start / end with its Robin
 *              coefficient, read from planes 0 / nx-1 of the axis-0 coefficient array); all-gather d_mat ->
 *              d_mat_all [nranks][4][nlines] (read for ranks 0 and nranks-1 only: a middle rank's entries are the two numbers
 *              (w[0], w[nx-1]) of ITS slab, d_scal_all [nranks][2], gathered likewise -- slabs may differ in thickness).
 * d_flags_first / _last, d_coeff_first / _last: planes 0 and nx-1 (dense ny*nz). */
int adi_deferred_exact_setup(const uint8_t *d_flags_first, const uint8_t *d_flags_last, const double *d_coeff_first,
                             const double *d_coeff_last, double theta, double gam, double dt, double w0, double wn,
                             long nlines, double *d_mat, double *d_kap, void *stream);
int adi_interface_solve_uniform(const double *d_g_all, const double *d_mat_all, const double *d_scal_all, int nranks, int rank,
                                long nlines, double *d_xlo, double *d_xhi, void *stream);
int adi_deferred_exact_coef(const double *d_xlo, const double *d_xhi, const double *d_kap, long nlines, double *d_clo,
                            double *d_chi, void *stream);
/* The deferred form for lines that are NOT uniform (curved solids, voids, Dirichlet cells -- an STL part on slabs; ABI v17,
 * reworked in v18).  The algebra above needs no uniform rows, only the two homogeneous solutions of every line; where they decay
 * across the slab they are non-zero on K < nx planes at each end.  v17 streamed K planes of per-cell weights at each end into
 * the axis-1 sweep (at cfl 200: 399 of 512 planes, more than the field itself).  v18 sorts the lines once per plan, per side,
 * by what their first (last) K rows look like:
 *     uniform  all K rows solid interior rows, no Dirichlet cell: the line's homogeneous solution IS the scalar w[i] of the
 *              uniform deferred form on those rows (up to the decay tolerance) -> adi_sweep_corrected adds it, as for boxes;
 *     off      all K rows outside the mask: identity rows, weight 0;
 *     flagged  everything else (the line crosses a void, the surface or a Dirichlet cell within reach of the interface): its
 *              own K weights, kept compactly as d_wc [K][nflag], and applied IN MEMORY to the zero-boundary solution by
 *              adi_deferred_lines_apply before the axis-1 sweep -- a sparse pass over the flagged lines only.
 *   per plan   two ordinary adi_sweep(axis 0) calls on the slab with a zero field, Tinf = 0, no fluxes, zero Dirichlet values
 *              and d_xlo = 1 (resp. d_xhi = 1): w_lo, w_hi of every line; the caller checks that they are below its tolerance
 *              beyond K planes, sorts the lines, compacts the weights of the flagged ones and exchanges the planes next to the
 *              interfaces (om_* below) with the neighbours;
 *   per step   step 1 as above; adi_interface_deferred_lines: the 2 x 2 systems with per-line weights -> d_ulo, d_uhi (every
 *              line) and d_ulo_uni / d_uhi_uni = the same values on the uniform lines, 0 elsewhere (d_uni_lo / d_uni_hi: one
 *              byte per line, non-zero = uniform; all four NULL: not wanted); adi_deferred_lines_apply for each side;
 *              adi_sweep_corrected with the scalar weights and the *_uni planes.
 * om_lo_own / om_hi_own: plane 0 of w_lo / plane nx-1 of w_hi of this rank; om_hi_prev / om_lo_next: the same planes of the
 * neighbours (dense ny*nz; NULL together with the neighbour's plane where there is none). */
int adi_interface_deferred_lines(const double *d_first, const double *d_last, const double *d_prev_last,
                                 const double *d_next_first, const double *d_om_lo_own, const double *d_om_hi_prev,
                                 const double *d_om_hi_own, const double *d_om_lo_next, long nlines, double *d_ulo,
                                 double *d_uhi, const uint8_t *d_uni_lo, const uint8_t *d_uni_hi, double *d_ulo_uni,
                                 double *d_uhi_uni, void *stream);
/* x[i][cell[q]] += wc[r][q] * u[cell[q]]  for q < nflag, r < K, with plane i = r (from_high_end == 0) or nx-1-r: the rank-one
 * update of the flagged lines of one side.  d_x: the slab field (nx planes, plane_stride elements apart); d_cells: nflag
 * offsets of the flagged lines inside a plane (j*nz + k, ascending); d_wc: [K][nflag]; d_u: a dense (ny*nz) interface plane;
 * d_nrows (may be NULL): per flagged line the number of leading rows that can carry a non-zero weight -- a line's weights are
 * exact zeros beyond its first row outside the mask, and rows >= d_nrows[q] are not even read. */
int adi_deferred_lines_apply(double *d_x, int nx, long plane_stride, long plane_cells, const int *d_cells, long nflag,
                             const double *d_wc, int K, const double *d_u, int from_high_end, const int *d_nrows, void *stream);
int adi_sweep_corrected(int variant, const double *d_in, const uint8_t *d_flags, const double *d_coeff,
                        const uint8_t *d_dir_mask, const double *d_dir_val, const double *d_qflux,
                        int nx, int ny, int nz, long plane_stride, int sparse,
                        double theta, double gam, double dt, double Tinf,
                        double *d_out, const double *d_ulo, const double *d_uhi, const double *d_w,
                        const double *h_face_consts, void *d_work, size_t work_bytes, void *stream);

/*
 * adi_step_numba_coeff / adi_step_gpu_coeff: adi3d_numba_coeff.py:290-302, adi3d_gpu_coeff.py:213-230.
 * d_T_in is not modified; d_T_out receives the new field; d_tmp_a / d_tmp_b are two scratch fields.
 * d_coeff[3], d_qflux[3]: per-axis pack arrays; dir data shared by the three packs (:116-118).
 */
int adi_step(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b,
             const uint8_t *d_flags, const double *const *d_coeff, const uint8_t *d_dir_mask,
             const double *d_dir_val, const double *const *d_qflux, int variant, int sparse,
             int nx, int ny, int nz, long plane_stride, double dx, double rho, double cp, double k,
             double dt, double theta, double Tinf, const double *h_face_consts /* [3][4] or NULL, see adi_sweep */,
             void *d_work, size_t work_bytes, void *stream);
/* The same step, reporting in h_queued[axis] (three HOST words, valid once the stream is synchronised) how many units the
 * FAST kernel of each sweep handed to the GENERAL kernel.  That number depends on the flags, the Dirichlet mask, the
 * variant, `sparse` and the shape only, so three zeros license bit 2 of `sparse` (the no-fallback promise, see adi_sweep) on
 * every later step of the same configuration; adi_ctx_step does this on the first step after a mask / pack change. */
int adi_step_queued(const double *d_T_in, double *d_T_out, double *d_tmp_a, double *d_tmp_b,
             const uint8_t *d_flags, const double *const *d_coeff, const uint8_t *d_dir_mask,
             const double *d_dir_val, const double *const *d_qflux, int variant, int sparse,
             int nx, int ny, int nz, long plane_stride, double dx, double rho, double cp, double k,
             double dt, double theta, double Tinf, const double *h_face_consts,
             void *d_work, size_t work_bytes, void *stream, unsigned *h_queued);

/*
 * The callers either side of the path.  Masks here are DENSE uint8 (nx, ny, nz), C order.
 * adi_morph6: op 0 = dilate6 (waam_from_stl_v7_mm.py:73-82), op 1 = erode6 (:84-96); d_out must not alias d_in.
 * adi_flood_outside: flood_fill_outside (:106-134) as its comment describes it -- air connected to the outside of the
 *   box through 6-connectivity (the reference pads the solid with True and therefore returns all-False: DESIGN.md D8);
 *   line scans instead of one-cell dilations; d_flag: one int of device scratch; synchronises the stream.
 * adi_pack_frame_f32be: fp64 field in the padded-plane layout -> big-endian float32 in VTK point order (x fastest),
 *   the payload of a legacy-VTK BINARY SCALARS block (the reference writes ASCII: vtk_writer.py:4-30).
 */
int adi_morph6(int op, const uint8_t *d_in, uint8_t *d_out, int nx, int ny, int nz, void *stream);
int adi_flood_outside(const uint8_t *d_solid, uint8_t *d_outside, int nx, int ny, int nz, int *d_flag, int *rounds,
                      void *stream);
int adi_pack_frame_f32be(const double *d_T, int nx, int ny, int nz, long plane_stride, uint32_t *d_out, void *stream);

/* T[sel != 0] = value   (layer birth: waam_from_stl_v7_mm.py:487-495 `T[newborn] = Ts`); flat over n elements */
int adi_masked_fill(double *d_T, const uint8_t *d_sel, size_t n, double value, void *stream);
/* dst = a | b  (birth bookkeeping: mask_act |= newborn) */
int adi_mask_or(uint8_t *d_dst, const uint8_t *d_a, const uint8_t *d_b, size_t n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Cylindrical (r, phi, z) backward-Euler step: adi3d_cyl_phi_v3.py:332-350 (scheme="be").
 * Field layout (nr, nphi, nz) with the same padded-plane convention (plane = one radius).
 * d_S may be NULL (no source).  d_tmp_a/d_tmp_b: scratch fields.
 * d_active (optional, may be NULL): adi_step_masked of quick_spiral_deposition_gif_v5.py:31-70 --
 * void cells are clamped to T_void before and after the step, inactive axis-row cells to T_inner.
 * ---------------------------------------------------------------------------------------------- */
typedef struct adi_cyl_plan adi_cyl_plan;
/* GridCyl + Material + Params(dt) + RobinR + ZBC (adi3d_cyl_phi_v3.py:33-68) frozen into device tables:
 * the r coefficients of build_coeff_r (:155-202), the per-radius phi factors of phi_solve_spectral
 * (:302-329) and the z closures of build_coeff_z (:255-298).  Created on the current device.
 * ADI_ERR_ARG ("unknown zbc.kind_bot/top") mirrors the reference's ValueError (:283, :296). */
int adi_cyl_plan_create(int nr, int nphi, int nz, long plane_stride, double dr, double dphi, double dz,
                        double rho, double cp, double k, double dt,
                        double robin_h, double robin_Tinf,
                        int kind_bot, int kind_top, double h_bot, double h_top,
                        double Tinf_bot, double Tinf_top, double T_bot, double T_top,
                        adi_cyl_plan **out);
/* The same on an annular grid r_i = r_in + (i + 1/2) dr -- the grid quick_spiral_deposition_gif_v5.py:74-80
 * (build_grid_annular) and tests/test_spiral_vs_analytic.py ask for with GridCyl(..., R_in=R_in); the reference's GridCyl has
 * no such parameter (TypeError at HEAD).  Every formula is the reference's with r shifted: row 0 keeps the
 * zero-flux closure of :177-181 (now the inner wall) and its identity phi row (:315-317). */
int adi_cyl_plan_create_annular(int nr, int nphi, int nz, long plane_stride, double dr, double dphi, double dz,
                                double r_in, double rho, double cp, double k, double dt,
                                double robin_h, double robin_Tinf,
                                int kind_bot, int kind_top, double h_bot, double h_top,
                                double Tinf_bot, double Tinf_top, double T_bot, double T_top,
                                adi_cyl_plan **out);
int adi_cyl_plan_destroy(adi_cyl_plan *plan);
/* adi_step (BE): T_in is not modified and must differ from T_out.  d_tmp_a / d_tmp_b are unused since ABI v13 (may be
 * NULL): the r sweep writes T_out and the phi and z sweeps run in place on it. */
int adi_cyl_step(const adi_cyl_plan *plan, const double *d_T_in, double *d_T_out,
                 double *d_tmp_a, double *d_tmp_b, const double *d_S,
                 const uint8_t *d_active, double T_void, double T_inner, void *stream);
/* One sweep of the BE step (stage entry point for per-stage parity tests and benchmarks): axis 0 = r
 * (build_coeff_r + thomas_batch, adi3d_cyl_phi_v3.py:155-202, :71-87; d_S / d_active / T_void as in adi_cyl_step),
 * axis 1 = phi (phi_solve_spectral, :302-329), axis 2 = z (build_coeff_z + thomas_batch, :255-298; d_active / T_void /
 * T_inner: the post-clamp of adi_step_masked).  d_out may equal d_in (ABI v13): every sweep kernel reads only the rows it
 * writes, so a sweep can run in place -- which halves the working set of a loop that does not need the previous field. */
int adi_cyl_sweep(const adi_cyl_plan *plan, int axis, const double *d_in, double *d_out, const double *d_S,
                  const uint8_t *d_active, double T_void, double T_inner, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Context API: the library owns the device memory; callers hand over HOST arrays.
 * One context per GPU and thread; no hidden global state.
 * ---------------------------------------------------------------------------------------------- */
typedef struct adi_ctx adi_ctx;

/* Grid3D(nx,ny,nz,dx,mask): adi3d_numba_coeff.py:14-19 (mask copied) */
int adi_ctx_create(int nx, int ny, int nz, double dx, int device, adi_ctx **out);
int adi_ctx_destroy(adi_ctx *ctx);
/* grid.mask = ... (drivers rebind the mask between steps: single_track_on_plate.py:159-160) */
int adi_ctx_set_mask(adi_ctx *ctx, const uint8_t *h_mask);
/* precompute_coeff_packs_unified with host-side BC data; h_dir_mask/h_dir_val may be NULL */
int adi_ctx_build_coeffs(adi_ctx *ctx, double rho, double cp,
                         const int *h_mode, const double *h_scalar, const double *const *h_h_field,
                         const int *q_mode, const double *q_scalar, const double *const *h_q_field,
                         const uint8_t *h_dir_mask, const double *h_dir_val);
int adi_ctx_upload_T(adi_ctx *ctx, const double *h_T);
int adi_ctx_download_T(adi_ctx *ctx, double *h_T);
/* read the device-built pack arrays back (axis 0..2); any pointer may be NULL */
int adi_ctx_download_pack(adi_ctx *ctx, int axis, double *h_coeff, double *h_qflux);
/* nsteps calls of adi_step with the same packs and dt (drivers' inner loop,
 * quick_compare_dirichlet_robin.py:169-178); T stays resident on the device */
int adi_ctx_step(adi_ctx *ctx, double rho, double cp, double k, double dt, double theta, double Tinf, int nsteps);
/* wall time in milliseconds of the last adi_ctx_step, measured with HIP events on the context's stream */
int adi_ctx_last_step_ms(adi_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* ADI_HIP_H */
