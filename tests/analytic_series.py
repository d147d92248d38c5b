"""Closed-form transient solutions for a semi-infinite solid cylinder with convective (Robin) side losses -- the
analytic side of the reference's quick_compare_dirichlet_robin.py (series :10-16, :68-113) and
quick_compare_neumann_robin.py (:41-79) -- and the driver recipe both scripts share (voxel cylinder, side Robin,
far end held at ambient, sub-stepping with dt <= cfl dx^2 / alpha), written against any module with the reference's
operator surface.  The reference's scripts only plot these comparisons; tests/test_analytic_physics.py asserts them.

  eigenvalues   mu_n: roots of mu J0'(mu) + Bi J0(mu) = 0, i.e. Bi J0(mu) = mu J1(mu);  lambda_n = mu_n / R
  Dirichlet step at z = 0:   T = Tinf + sum_n A_n J0(lambda_n r) F(z, t; lambda_n),
        A_n = 2 (Ts - Tinf) J1(mu_n) / (mu_n (J0(mu_n)^2 + J1(mu_n)^2))
        F = 1/2 [ e^{-lz} erfc(z / 2 sqrt(at) - l sqrt(at)) + e^{lz} erfc(z / 2 sqrt(at) + l sqrt(at)) ]
  Neumann heating q0 at z = 0:   T = Tinf + (q0 / k) sum_n C_n J0(lambda_n r) K(z, t; lambda_n),
        C_n = A_n / (Ts - Tinf),   K = 1/(2l) [ e^{-lz} erfc(A - B) - e^{lz} erfc(A + B) ],  A = z / 2 sqrt(at), B = l sqrt(at)
"""
import math

import numpy as np


def robin_mu_roots(Bi, n_roots):
    """the first n_roots positive roots of Bi J0(mu) - mu J1(mu): exactly one lies between consecutive zeros of J1
    (0 = j_{1,0} < j_{1,1} < ...), where the function changes sign"""
    from scipy.optimize import brentq
    from scipy.special import j0, j1, jn_zeros
    edges = np.concatenate(([0.0], jn_zeros(1, n_roots)))
    f = lambda m: Bi * j0(m) - m * j1(m)
    return np.array([brentq(f, edges[i] + 1e-12, edges[i + 1] - 1e-12, xtol=1e-14, maxiter=200) for i in range(n_roots)])


def _modes(Bi, R, n_modes):
    from scipy.special import j0, j1
    mu = robin_mu_roots(Bi, n_modes)
    c = 2.0 * j1(mu) / (mu * (j0(mu) ** 2 + j1(mu) ** 2))
    return mu / R, c


def dirichlet_step_axis(z, t, alpha, R, Bi, Ts, Tinf, n_modes=18):
    """temperature on the axis r = 0 after the face z = 0 was switched to Ts at t = 0"""
    from scipy.special import erfc
    lam, c = _modes(Bi, R, n_modes)
    z = np.asarray(z, dtype=float)
    s = math.sqrt(alpha * max(float(t), 1e-15))
    A = z[None, :] / (2.0 * s)
    B = lam[:, None] * s
    F = 0.5 * (np.exp(-lam[:, None] * z[None, :]) * erfc(A - B) + np.exp(lam[:, None] * z[None, :]) * erfc(A + B))
    return Tinf + (Ts - Tinf) * (c[:, None] * F).sum(axis=0)


def neumann_heating_axis(z, t, alpha, R, Bi, q0, k, Tinf, n_modes=18):
    """temperature on the axis r = 0 under a constant flux q0 (into the body) through the face z = 0 since t = 0"""
    from scipy.special import erfc, erfcx
    lam, c = _modes(Bi, R, n_modes)
    z = np.asarray(z, dtype=float)
    s = math.sqrt(alpha * max(float(t), 1e-15))
    A = z[None, :] / (2.0 * s)
    B = lam[:, None] * s
    # e^{lz} erfc(A + B) = e^{lz - (A+B)^2} erfcx(A + B): no overflow for large lambda z
    K = 0.5 / lam[:, None] * (np.exp(-lam[:, None] * z[None, :]) * erfc(A - B)
                              - np.exp(lam[:, None] * z[None, :] - (A + B) ** 2) * erfcx(A + B))
    return Tinf + (q0 / k) * (c[:, None] * K).sum(axis=0)


def cylinder_mask(nx, ny, nz, dx, R):
    """voxel cylinder along axis 2, centred in the box (quick_compare_dirichlet_robin.py:116-123)"""
    xs = (np.arange(nx) + 0.5 - nx / 2.0) * dx
    ys = (np.arange(ny) + 0.5 - ny / 2.0) * dx
    m2 = np.sqrt(xs[:, None] ** 2 + ys[None, :] ** 2) <= R + 1e-12
    return np.ascontiguousarray(np.repeat(m2[:, :, None], nz, axis=2)), m2


def run_cylinder(api, kind, nxr, nz, times, mat, R=0.02, h_side=500.0, Tinf=20.0, Ts=1000.0, q0=2.0e5, theta=0.5, cfl=2.0,
                 to_state=lambda x: x, to_host=lambda x: np.asarray(x), h_scale=1.0):
    """the numerical side of both scripts: kind 'dirichlet' (face z- held at Ts, :129-144) or 'neumann' (flux q0 through
    z-, quick_compare_neumann_robin.py:91-102); returns the axis profiles T(r = 0, z) at `times` and the grid spacing"""
    alpha = mat['k'] / (mat['rho'] * mat['cp'])
    dx = R / float(nxr)
    nx = ny = int(round(2.0 * R / dx))
    mask, _ = cylinder_mask(nx, ny, nz, dx, R)
    grid = api.Grid3D(nx, ny, nz, dx, mask)
    m = api.Material(**mat)
    dm = np.zeros(mask.shape, bool)
    dv = np.full(mask.shape, Tinf)
    dm[:, :, nz - 1] = mask[:, :, nz - 1]
    neumann = None
    if kind == 'dirichlet':
        dm[:, :, 0] = mask[:, :, 0]
        dv[:, :, 0] = Ts
    else:
        neumann = {'z-': q0}
    hs = h_side * h_scale
    packs = api.precompute_coeff_packs_unified(grid, m, dir_mask=dm, dir_value=dv, neumann=neumann,
                                               robin_h={'x-': hs, 'x+': hs, 'y-': hs, 'y+': hs}, robin_Tinf=Tinf)
    step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
    dt_cap = cfl * dx * dx / alpha
    prm = api.Params(dt=1e-3, theta=theta)
    T = to_state(np.full(mask.shape, Tinf))
    t_cur, out = 0.0, []
    for tt in times:
        remain = float(tt - t_cur)
        nsub = max(1, int(math.ceil(remain / dt_cap))) if remain > 0 else 0
        prm.dt = max(remain / nsub if nsub else 0.0, 1e-15)
        for _ in range(nsub):
            T = step(T, grid, m, prm, packs, Tinf=Tinf)
        t_cur = tt
        out.append(to_host(T)[nx // 2, ny // 2, :].copy())
    return np.array(out), dx
