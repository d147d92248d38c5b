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


# ---- spiral deposition on an annular wall ------------------------------------------------------------------------------
# The analytic side of the reference's only test (tests/test_spiral_vs_analytic.py with spiral_analytic_solution.py:
# eigenvalue determinant :92-105, radial modes :128-150, deposition events :153-163, axial kernel :194-203, assembly
# :205-312), restated here as the checker.  A layer is laid in n_slices arc events per turn; an event at time t_e and
# angle phi_e contributes, at time t = t_e + u, radius r_p, angle phi and depth s below the current top,
#       dT * (dphi / 2 pi) * sum_m cos m(phi - phi_e) * sum_n P_mn R_mn(r_p) e^{-alpha kappa_mn^2 u} * int_0^h G(s, xi + off, u) dxi
# with R_mn = J_m(kappa r) + B Y_m(kappa r) the Robin eigenfunctions of the annulus a <= r <= b normalised in the r-weighted
# L2 norm, P_mn = int R_mn r dr, and G the 1-D heat kernel on a half line with a Robin end (image + erfc correction).
def _annulus_modes(m, a, b, gi, go, n_modes, r_probe):
    """eigenvalues kappa of the m-th angular order (sign changes of the 2 x 2 boundary determinant on a fine grid up to
    kappa = 400, refined by bracketing) with each mode's projection P = int R r dr and value at the probe radius"""
    from scipy.optimize import brentq
    from scipy.special import jv, jvp, yv, yvp

    def rows(kap):
        ra = (-kap * jvp(m, kap * a, 1) - gi * jv(m, kap * a), -kap * yvp(m, kap * a, 1) - gi * yv(m, kap * a))
        rb = (-kap * jvp(m, kap * b, 1) - go * jv(m, kap * b), -kap * yvp(m, kap * b, 1) - go * yv(m, kap * b))
        return ra, rb

    def det(kap):
        ra, rb = rows(kap)
        return ra[0] * rb[1] - ra[1] * rb[0]

    xs = np.linspace(1e-6, 400.0, 20000)
    v = det(xs)
    sg = np.sign(v)
    kaps = []
    for i in np.nonzero((sg[:-1] * sg[1:] < 0) & np.isfinite(v[:-1]) & np.isfinite(v[1:]))[0]:
        try:
            root = brentq(det, xs[i], xs[i + 1], maxiter=200)
        except ValueError:
            continue
        if not kaps or abs(root - kaps[-1]) > 1e-6:
            kaps.append(root)
        if len(kaps) >= n_modes:
            break
    rs = np.linspace(a, b, 1024)
    w = np.gradient(rs)
    P, Rp = [], []
    for kap in kaps:
        ra, _ = rows(kap)
        B = 0.0 if abs(ra[1]) < 1e-14 else -ra[0] / ra[1]
        shape = lambda r: jv(m, kap * r) + B * yv(m, kap * r)
        Rv = shape(rs)
        scale = 1.0 / math.sqrt(max(float(np.sum(Rv * Rv * rs * w)), 1e-30))
        P.append(float(np.sum(Rv * rs * w)) * scale)
        Rp.append(float(shape(r_probe)) * scale)
    return np.array(kaps), np.array(P), np.array(Rp)


def _robin_end_kernel(s, xi, u, alpha, beta):
    from scipy.special import erfc
    den = np.sqrt(4.0 * np.pi * alpha * u)
    direct = np.exp(-(s - xi) ** 2 / (4.0 * alpha * u)) / den
    image = np.exp(-(s + xi) ** 2 / (4.0 * alpha * u)) / den
    corr = beta * np.exp(beta * (s + xi) + alpha * beta * beta * u) * erfc((s + xi) / (2.0 * np.sqrt(alpha * u)) + beta * np.sqrt(alpha * u))
    return direct + image - corr


class SpiralAnnulus:
    """analytic (phi, z) temperature map at the mid-wall radius of an annular wall built by spiral deposition"""

    def __init__(self, mat, Tinf, Tdep, R_in, wall, h_in, h_out, h_end, z_back, layer_h, n_layers, tau, n_slices, nz, nphi,
                 m_ang=12, n_radial=6):
        self.alpha = mat['k'] / (mat['rho'] * mat['cp'])
        self.dT = Tdep - Tinf
        self.Tinf, self.beta = Tinf, h_end / mat['k']
        self.layer_h, self.n_layers, self.tau, self.n_slices, self.z_back = layer_h, n_layers, tau, n_slices, z_back
        a, b = R_in, R_in + wall
        self.modes = [_annulus_modes(m, a, b, h_in / mat['k'], h_out / mat['k'], n_radial, 0.5 * (a + b)) for m in range(m_ang + 1)]
        self.phi = np.linspace(0.0, 2.0 * np.pi, nphi, endpoint=False)
        self.z = np.linspace(-z_back, layer_h * n_layers, nz)
        dts = tau / n_slices
        self.events = sorted((l * tau + (p + 0.5) * dts, l, 2.0 * np.pi * (p + 0.5) / n_slices)
                             for l in range(n_layers) for p in range(n_slices))

    def map_at(self, t):
        """(nz, nphi) temperatures, NaN where the wall does not exist yet"""
        nz, nphi = self.z.size, self.phi.size
        full = min(self.n_layers, int(math.floor(t / self.tau)))
        top = full * self.layer_h
        frac = max(0.0, min(1.0, (t - full * self.tau) / self.tau)) if full < self.n_layers else 0.0
        progress = 2.0 * np.pi * frac
        exist = np.zeros((nz, nphi), bool)
        exist[(self.z <= top) & (self.z >= -self.z_back), :] = True
        if full < self.n_layers and frac > 0.0:
            band = (self.z >= top) & (self.z <= top + self.layer_h)
            exist[band, :] |= (self.phi < progress)[None, :]
        rows = np.nonzero(exist.any(axis=1))[0]
        out = np.full((nz, nphi), np.nan)
        if rows.size == 0:
            return out
        s = top - self.z[rows]
        xi = np.linspace(0.0, self.layer_h, 64)
        wxi = np.gradient(xi)
        dphi = 2.0 * np.pi / self.n_slices
        acc = np.zeros((rows.size, nphi))
        for (te, layer, phie) in self.events:
            if te > t:
                break
            u = t - te
            if u <= 0.0 or layer > full or (layer == full and phie > progress):
                continue
            off = (full - 1 - layer) * self.layer_h if layer < full else 0.0
            axial = (_robin_end_kernel(s[:, None], xi[None, :] + off, u, self.alpha, self.beta) * wxi).sum(axis=1)
            ang = np.zeros(nphi)
            for m, (kap, P, Rp) in enumerate(self.modes):
                if kap.size:
                    ang += float(np.sum(P * Rp * np.exp(-self.alpha * kap * kap * u))) * np.cos(m * (self.phi - phie))
            acc += (self.dT * dphi / (2.0 * np.pi)) * axial[:, None] * ang[None, :]
        out[rows, :] = np.where(exist[rows, :], self.Tinf + acc, np.nan)
        return out


def run_spiral_deposition(api, times, mat, Tinf, Tdep, R_in, wall, h_side, h_end, z_back, layer_h, n_layers, tau, nr, nphi,
                          to_state=lambda x: x, to_host=lambda x: np.asarray(x)):
    """The numeric side of the same test (tests/test_spiral_vs_analytic.py:17-120): one z-cell per layer, nphi cells per
    turn, dt = tau / nphi, BE; before each step the arc swept during it is activated at Tdep (whole radial columns), then
    adi_step_masked advances the field.  Returns the grid, the (nr, nphi, nz) fields and the active masks at `times`."""
    dr = wall / nr
    nz = int(round((z_back + layer_h * n_layers) / layer_h))
    grid = api.GridCyl(nr, nphi, nz, dr, 2.0 * math.pi / nphi, layer_h, R_in + wall, R_in=R_in)
    m = api.Material(mat['rho'], mat['cp'], mat['k'])
    wall_bc = api.RobinR(h_side, Tinf)
    zbc = api.ZBC(kind_bot='neumann0', kind_top='robin', h_top=h_end, T_inf_top=Tinf)
    iz0 = int(round(z_back / layer_h))
    T = np.full((nr, nphi, nz), float(Tinf))
    active = np.zeros((nr, nphi, nz), bool)
    active[:, :, :iz0] = True
    dt, omega = tau / nphi, 2.0 * math.pi / tau
    prm = api.Params(dt, 1.0, "be")
    layer, angle, t = 0, 0.0, 0.0
    fields, masks = [], []
    for t_goal in times:
        while t < t_goal - 1e-12:
            t_next = min(t + dt, t_goal)
            left = omega * (t_next - t)
            while left > 0.0 and layer < n_layers:
                seg = min(left, 2.0 * math.pi - angle)
                if seg > 0.0:
                    iz = iz0 + layer
                    if 0 <= iz < nz:
                        first = int(math.floor(angle / grid.dphi))
                        last = max(first, int(math.floor((angle + seg - 1e-12) / grid.dphi)))
                        for c in range(first, last + 1):
                            if not active[0, c % nphi, iz]:
                                active[:, c % nphi, iz] = True
                                T[:, c % nphi, iz] = Tdep
                    angle += seg
                    left -= seg
                if angle >= 2.0 * math.pi - 1e-15:
                    angle = 0.0
                    layer += 1
                    if iz0 + layer > nz - 1:
                        layer = n_layers
            prm.dt = t_next - t
            T = to_host(api.adi_step_masked(to_state(T), grid, m, prm, wall_bc, zbc, active, robin_inner=wall_bc,
                                            robin_void=wall_bc)).copy()
            t = t_next
        fields.append(T.copy())
        masks.append(active.copy())
    return grid, fields, masks
