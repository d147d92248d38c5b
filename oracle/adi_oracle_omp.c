/*
 * oracle/adi_oracle_omp.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * OpenMP-over-lines variant of oracle/adi_oracle.c, used ONLY by bench.py's
 * cpu_baseline leg to report an all-host-cores figure next to the faithful
 * single-thread one (the reference's Numba kernels are serial:
 * adi3d_numba_coeff.py:120-288 have no prange/parallel=True).
 * The per-cell arithmetic and its order are identical to adi_oracle.c, so the
 * result is bit-identical (lines are independent); tests/test_oracle_golden.py
 * checks that.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* adi3d_numba_coeff.py:240-288 + :298 fused per cell (same expression order) */
static void explicit_rhs_omp(const double *T, const uint8_t *mask, int nx, int ny, int nz, double dx,
                             double dt, double kappa, double theta, double *R0)
{
    double invdx2 = 1.0 / (dx * dx);
    double f = dt * kappa * (1.0 - theta);
    size_t sx = (size_t)ny * nz, sy = (size_t)nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                size_t p = ((size_t)i * ny + j) * nz + k;
                double L[3] = {0.0, 0.0, 0.0};
                if (mask[p]) {
                    int pos[3] = {i, j, k}, n3[3] = {nx, ny, nz};
                    size_t st[3] = {sx, sy, 1};
                    for (int a = 0; a < 3; ++a) {
                        double s = 0.0, c = 0.0;
                        if (pos[a] - 1 >= 0 && mask[p - st[a]]) { s += T[p - st[a]]; c += 1.0; }
                        if (pos[a] + 1 < n3[a] && mask[p + st[a]]) { s += T[p + st[a]]; c += 1.0; }
                        L[a] = (s - c * T[p]) * invdx2;
                    }
                }
                R0[p] = T[p] + f * ((L[0] + L[1]) + L[2]);
            }
}

/* adi3d_numba_coeff.py:133-237 with the two line loops parallelised */
static void sweep_axis_omp(int axis, const double *in, const uint8_t *mask,
                           const double *coeff_rob, const uint8_t *dir_mask, const double *dir_val,
                           const double *qflux, int nx, int ny, int nz,
                           double theta, double gam, double dt, double Tinf, double *out)
{
    size_t N = (size_t)nx * ny * nz;
    int n3[3] = {nx, ny, nz};
    int n = n3[axis];
    size_t stride3[3] = {(size_t)ny * nz, (size_t)nz, 1};
    size_t sa = stride3[axis];
    int o1 = (axis == 0) ? 1 : 0, o2 = (axis == 2) ? 1 : 2;
    if (out != in) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nx; ++i) memcpy(out + (size_t)i * ny * nz, in + (size_t)i * ny * nz, (size_t)ny * nz * sizeof(double));
    }
    (void)N;
#pragma omp parallel
    {
        double *a = (double *)malloc(sizeof(double) * n * 5);
        double *b = a + n, *c = b + n, *d = c + n, *x = d + n;
        int *idx = (int *)malloc(sizeof(int) * n);
#pragma omp for collapse(2) schedule(static)
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
                        d[cnt] = in[p] + dt * qflux[p] + dt * coeff_rob[p] * Tinf;
                    }
                    cnt += 1;
                }
                if (cnt == 0) continue;
                for (int i = 1; i < cnt; ++i) {
                    double m = a[i] / b[i - 1];
                    b[i] = b[i] - m * c[i - 1];
                    d[i] = d[i] - m * d[i - 1];
                }
                x[cnt - 1] = d[cnt - 1] / b[cnt - 1];
                for (int i = cnt - 2; i >= 0; --i) x[i] = (d[i] - c[i] * x[i + 1]) / b[i];
                for (int m = 0; m < cnt; ++m) out[base + (size_t)idx[m] * sa] = x[m];
            }
        free(a); free(idx);
    }
}

/* a copy of `src` whose pages are first touched by the threads that will work on them (planes of the slowest axis,
 * the static schedule of the loops above): NumPy's arrays are first touched by one thread, i.e. they all live on one
 * NUMA node, and an all-cores run of them measures that node's memory controller (round 1: 256 cores gave 4.4x one) */
static void *numa_copy(const void *src, size_t nplanes, size_t plane_bytes)
{
    char *dst = (char *)malloc(nplanes * plane_bytes);
    if (!dst) return NULL;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)nplanes; ++i) {
        if (src) memcpy(dst + (size_t)i * plane_bytes, (const char *)src + (size_t)i * plane_bytes, plane_bytes);
        else memset(dst + (size_t)i * plane_bytes, 0, plane_bytes);
    }
    return dst;
}

void oracle_omp_adi_run(double *T, const uint8_t *mask, int nx, int ny, int nz, double dx,
                        double rho, double cp, double kcond, double dt, double theta,
                        const double *const *coeff, const uint8_t *dir_mask, const double *dir_val,
                        const double *const *qflux, double Tinf, int nsteps)
{
    size_t N = (size_t)nx * ny * nz, P = (size_t)ny * nz;
    double kappa = kcond / (rho * cp);
    double gam = kappa * dt / (dx * dx);
    /* working set with parallel first touch (the arithmetic is unchanged: same cells, same order per cell) */
    double *Tl = (double *)numa_copy(T, nx, P * sizeof(double));
    double *A = (double *)numa_copy(NULL, nx, P * sizeof(double));
    double *B = (double *)numa_copy(NULL, nx, P * sizeof(double));
    uint8_t *ml = (uint8_t *)numa_copy(mask, nx, P), *dml = (uint8_t *)numa_copy(dir_mask, nx, P);
    double *dvl = (double *)numa_copy(dir_val, nx, P * sizeof(double));
    double *cl[3], *ql[3];
    for (int a = 0; a < 3; ++a) {
        cl[a] = (double *)numa_copy(coeff[a], nx, P * sizeof(double));
        ql[a] = (double *)numa_copy(qflux[a], nx, P * sizeof(double));
    }
    for (int s = 0; s < nsteps; ++s) {
        explicit_rhs_omp(Tl, ml, nx, ny, nz, dx, dt, kappa, theta, A);
        sweep_axis_omp(0, A, ml, cl[0], dml, dvl, ql[0], nx, ny, nz, theta, gam, dt, Tinf, B);
        sweep_axis_omp(1, B, ml, cl[1], dml, dvl, ql[1], nx, ny, nz, theta, gam, dt, Tinf, A);
        sweep_axis_omp(2, A, ml, cl[2], dml, dvl, ql[2], nx, ny, nz, theta, gam, dt, Tinf, Tl);
    }
    memcpy(T, Tl, N * sizeof(double));
    free(Tl); free(A); free(B); free(ml); free(dml); free(dvl);
    for (int a = 0; a < 3; ++a) { free(cl[a]); free(ql[a]); }
}
