/*
 * oracle/adi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's Cartesian masked-voxel ADI step
 * (Matemusi/ADI_thermal_fields, adi3d_numba_coeff.py).  It exists only to check
 * the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under adi_thermal_fields_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * bit-for-bit against golden vectors produced by importing the reference Python
 * in the build container (tests/golden/make_golden.py) and against the KAT1/KAT2
 * spot values recorded in SURVEY.md section 8(c).
 *
 * Every expression keeps the reference's evaluation order so that, compiled with
 * -ffp-contract=off on x86-64 (SSE2 doubles), results are bit-identical to the
 * CPython/NumPy evaluation of the reference.
 *
 * Layout: C-order (nx, ny, nz) arrays, fp64 fields, 1-byte bool masks
 * (adi3d_numba_coeff.py:14-19, :29-36).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, k) ((((size_t)(i)) * ny + (size_t)(j)) * nz + (size_t)(k))

/* face codes: 0 'x-', 1 'x+', 2 'y-', 3 'y+', 4 'z-', 5 'z+' */

/* adi3d_numba_coeff.py:38-55  exposed_mask(mask, face) */
int oracle_exposed_mask(const uint8_t *mask, int nx, int ny, int nz, int face, uint8_t *exp)
{
    if (face < 0 || face > 5) return 1; /* ValueError("bad face") :54 */
    int axis = face / 2, plus = face & 1;
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                int p[3] = {i, j, k};
                int n[3] = {nx, ny, nz};
                uint8_t m = mask[IDX(i, j, k)] ? 1 : 0;
                int q = p[axis] + (plus ? 1 : -1);
                uint8_t e;
                if (q < 0 || q >= n[axis]) {
                    e = m; /* domain-boundary plane counts as exposed */
                } else {
                    int pp[3] = {i, j, k};
                    pp[axis] = q;
                    e = (uint8_t)(m && !mask[IDX(pp[0], pp[1], pp[2])]);
                }
                exp[IDX(i, j, k)] = e;
            }
    return 0;
}

/*
 * adi3d_numba_coeff.py:57-118  precompute_coeff_packs_unified
 *
 * h_mode[f]: 0 = face has no Robin data (robin_h is None), 1 = scalar h_scalar[f],
 *            2 = per-voxel array h_field[f].  Same for q_mode / q_scalar / q_field
 *            (0 = face absent from the `neumann` dict or value None).
 * coeff[3], qflux[3]: outputs per axis (zero-initialised here, :90-92, :101-103).
 */
void oracle_build_coeffs(const uint8_t *mask, int nx, int ny, int nz, double dx,
                         double rho, double cp,
                         const int *h_mode, const double *h_scalar, const double *const *h_field,
                         const int *q_mode, const double *q_scalar, const double *const *q_field,
                         double *const *coeff, double *const *qflux)
{
    size_t N = (size_t)nx * ny * nz;
    double A = dx * dx;
    double V = pow(dx, 3.0);      /* dx**3, :67 (CPython float_pow -> libm pow) */
    double Ccell = rho * cp * V;  /* :68 */
    uint8_t *exp = (uint8_t *)malloc(N);
    for (int a = 0; a < 3; ++a) {
        memset(coeff[a], 0, N * sizeof(double));
        memset(qflux[a], 0, N * sizeof(double));
    }
    for (int f = 0; f < 6; ++f) { /* :93-99 */
        if (h_mode[f] == 0) continue;
        oracle_exposed_mask(mask, nx, ny, nz, f, exp);
        double *c = coeff[f / 2];
        for (size_t p = 0; p < N; ++p)
            if (exp[p]) {
                double h = (h_mode[f] == 1) ? h_scalar[f] : h_field[f][p];
                c[p] += (h * A / Ccell);
            }
    }
    for (int f = 0; f < 6; ++f) { /* :104-114 */
        if (q_mode[f] == 0) continue;
        oracle_exposed_mask(mask, nx, ny, nz, f, exp);
        double *q = qflux[f / 2];
        for (size_t p = 0; p < N; ++p) {
            double S = 0.0;
            if (exp[p]) {
                double qv = (q_mode[f] == 1) ? q_scalar[f] : q_field[f][p];
                S = qv * A / Ccell;
            }
            q[p] += S;
        }
    }
    free(exp);
}

/* adi3d_numba_coeff.py:240-288  lap1D_x / lap1D_y / lap1D_z (axis = 0/1/2) */
void oracle_lap1d(const double *T, const uint8_t *mask, int nx, int ny, int nz, double dx,
                  int axis, double *out)
{
    double invdx2 = 1.0 / (dx * dx);
    int n[3] = {nx, ny, nz};
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                size_t p = IDX(i, j, k);
                out[p] = 0.0;
                if (!mask[p]) continue;
                int c3[3] = {i, j, k};
                double s = 0.0, c = 0.0;
                if (c3[axis] - 1 >= 0) {
                    int q[3] = {i, j, k};
                    q[axis] -= 1;
                    size_t pm = IDX(q[0], q[1], q[2]);
                    if (mask[pm]) { s += T[pm]; c += 1.0; }
                }
                if (c3[axis] + 1 < n[axis]) {
                    int q[3] = {i, j, k};
                    q[axis] += 1;
                    size_t pp = IDX(q[0], q[1], q[2]);
                    if (mask[pp]) { s += T[pp]; c += 1.0; }
                }
                out[p] = (s - c * T[p]) * invdx2;
            }
}

/* adi3d_numba_coeff.py:298  R0 = Tn + dt*kappa*(1.0-theta)*(Lx+Ly+Lz) */
void oracle_explicit_rhs(const double *T, const uint8_t *mask, int nx, int ny, int nz, double dx,
                         double dt, double kappa, double theta, double *R0)
{
    size_t N = (size_t)nx * ny * nz;
    double *Lx = (double *)malloc(N * sizeof(double));
    double *Ly = (double *)malloc(N * sizeof(double));
    double *Lz = (double *)malloc(N * sizeof(double));
    oracle_lap1d(T, mask, nx, ny, nz, dx, 0, Lx);
    oracle_lap1d(T, mask, nx, ny, nz, dx, 1, Ly);
    oracle_lap1d(T, mask, nx, ny, nz, dx, 2, Lz);
    double f = dt * kappa * (1.0 - theta);
    for (size_t p = 0; p < N; ++p) R0[p] = T[p] + f * ((Lx[p] + Ly[p]) + Lz[p]);
    free(Lx); free(Ly); free(Lz);
}

/* adi3d_numba_coeff.py:121-130  thomas_solve (in-place elimination form, no pivoting) */
static void thomas_solve(double *a, double *b, double *c, double *d, double *x, int n)
{
    for (int i = 1; i < n; ++i) {
        double m = a[i] / b[i - 1];
        b[i] = b[i] - m * c[i - 1];
        d[i] = d[i] - m * d[i - 1];
    }
    x[n - 1] = d[n - 1] / b[n - 1];
    for (int i = n - 2; i >= 0; --i) x[i] = (d[i] - c[i] * x[i + 1]) / b[i];
}

/* exported for the unit test of the solver alone */
void oracle_thomas_solve(double *a, double *b, double *c, double *d, double *x, int n)
{
    thomas_solve(a, b, c, d, x, n);
}

/*
 * adi3d_numba_coeff.py:133-237  sweep_axis0 / sweep_axis1 / sweep_axis2.
 * `in` is the previous stage (R0, U or V); `out` receives a copy of it with the
 * in-mask cells of every line replaced by the tridiagonal solution (compacted
 * per-line systems, :139-166).  Off-mask cells keep `in`.
 */
void oracle_sweep_axis(int axis, const double *in, const uint8_t *mask,
                       const double *coeff_rob, const uint8_t *dir_mask, const double *dir_val,
                       const double *qflux, int nx, int ny, int nz,
                       double theta, double gam, double dt, double Tinf, double *out)
{
    size_t N = (size_t)nx * ny * nz;
    int n3[3] = {nx, ny, nz};
    int n = n3[axis];
    size_t stride3[3] = {(size_t)ny * nz, (size_t)nz, 1};
    size_t sa = stride3[axis];
    int o1 = (axis == 0) ? 1 : 0, o2 = (axis == 2) ? 1 : 2; /* the two other axes */
    double *a = (double *)malloc(sizeof(double) * n * 5);
    double *b = a + n, *c = b + n, *d = c + n, *x = d + n;
    int *idx = (int *)malloc(sizeof(int) * n);
    if (out != in) memcpy(out, in, N * sizeof(double)); /* out = R0.copy() :135 */
    for (int u = 0; u < n3[o1]; ++u)
        for (int v = 0; v < n3[o2]; ++v) {
            size_t base = (size_t)u * stride3[o1] + (size_t)v * stride3[o2];
            int cnt = 0;
            for (int r = 0; r < n; ++r) {
                size_t p = base + (size_t)r * sa;
                if (!mask[p]) continue;
                idx[cnt] = r;
                int nnb = 0;
                double left = 0.0, right = 0.0;
                if (r - 1 >= 0 && mask[p - sa]) { left = -theta * gam; nnb += 1; }
                if (r + 1 < n && mask[p + sa]) { right = -theta * gam; nnb += 1; }
                double diag = 1.0 + theta * gam * nnb + dt * coeff_rob[p];
                if (dir_mask[p]) {
                    a[cnt] = 0.0; c[cnt] = 0.0; b[cnt] = 1.0; d[cnt] = dir_val[p];
                } else {
                    a[cnt] = left; b[cnt] = diag; c[cnt] = right;
                    d[cnt] = out[p] + dt * qflux[p] + dt * coeff_rob[p] * Tinf;
                }
                cnt += 1;
            }
            if (cnt == 0) continue;
            thomas_solve(a, b, c, d, x, cnt);
            for (int m = 0; m < cnt; ++m) out[base + (size_t)idx[m] * sa] = x[m];
        }
    free(a); free(idx);
}

/*
 * adi3d_numba_coeff.py:290-302  adi_step_numba_coeff.
 * coeff[3], qflux[3]: per-axis pack arrays; dir_mask/dir_val shared by the packs
 * (the reference stores three identical copies, :116-118).
 * stages (optional, may be NULL): stages[0..3] receive R0, U, V, W for per-stage tests.
 */
void oracle_adi_step(const double *Tn, const uint8_t *mask, int nx, int ny, int nz, double dx,
                     double rho, double cp, double kcond, double dt, double theta,
                     const double *const *coeff, const uint8_t *dir_mask, const double *dir_val,
                     const double *const *qflux, double Tinf, double *W, double *const *stages)
{
    size_t N = (size_t)nx * ny * nz;
    double kappa = kcond / (rho * cp);
    double gam = kappa * dt / (dx * dx);
    double *R0 = (double *)malloc(N * sizeof(double));
    double *U = (double *)malloc(N * sizeof(double));
    oracle_explicit_rhs(Tn, mask, nx, ny, nz, dx, dt, kappa, theta, R0);
    oracle_sweep_axis(0, R0, mask, coeff[0], dir_mask, dir_val, qflux[0], nx, ny, nz, theta, gam, dt, Tinf, U);
    if (stages) { memcpy(stages[0], R0, N * sizeof(double)); memcpy(stages[1], U, N * sizeof(double)); }
    /* reuse R0 as V */
    oracle_sweep_axis(1, U, mask, coeff[1], dir_mask, dir_val, qflux[1], nx, ny, nz, theta, gam, dt, Tinf, R0);
    if (stages) memcpy(stages[2], R0, N * sizeof(double));
    oracle_sweep_axis(2, R0, mask, coeff[2], dir_mask, dir_val, qflux[2], nx, ny, nz, theta, gam, dt, Tinf, W);
    if (stages) memcpy(stages[3], W, N * sizeof(double));
    free(R0); free(U);
}

/* nsteps repeated steps with constant packs/dt (the drivers' inner loop,
 * quick_compare_dirichlet_robin.py:169-178); used for the CPU baseline timing. */
void oracle_adi_run(double *T, const uint8_t *mask, int nx, int ny, int nz, double dx,
                    double rho, double cp, double kcond, double dt, double theta,
                    const double *const *coeff, const uint8_t *dir_mask, const double *dir_val,
                    const double *const *qflux, double Tinf, int nsteps)
{
    size_t N = (size_t)nx * ny * nz;
    double *W = (double *)malloc(N * sizeof(double));
    for (int s = 0; s < nsteps; ++s) {
        oracle_adi_step(T, mask, nx, ny, nz, dx, rho, cp, kcond, dt, theta, coeff, dir_mask, dir_val,
                        qflux, Tinf, W, NULL);
        memcpy(T, W, N * sizeof(double));
    }
    free(W);
}
