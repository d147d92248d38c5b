"""oracle/cyl_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's cylindrical (r, phi, z) backward-Euler ADI
step, adi3d_cyl_phi_v3.py (valid path only: scheme="be", :338-350; the "douglas"
branch and the cyclic-Thomas helpers are broken/dead in the reference, SURVEY.md
D2/D3, and are deliberately not restated), plus the void-clamping wrapper
adi_step_masked of quick_spiral_deposition_gif_v5.py:31-70.

Third-party arithmetic: the periodic phi solve uses numpy.fft.rfft/irfft exactly
as the reference does (adi3d_cyl_phi_v3.py:325-328; numpy 2.2.6 in this image).

Parity status: PINNED by tests/test_oracle_golden.py (golden vectors from the
imported reference + KAT3 of SURVEY.md 8(c)).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np


class GridCyl:  # adi3d_cyl_phi_v3.py:33-43
    def __init__(self, nr, nphi, nz, dr, dphi, dz, R, R_in=0.0):
        # R_in: the annular grid quick_spiral_deposition_gif_v5.py:80 asks for (the reference's constructor raises TypeError,
        # SURVEY D1): r shifted by the inner radius, everything else as written.  R_in != 0 is pinned by
        # tests/golden/cyl_spiral_annulus.npz (the reference's own numeric loop run with a GridCyl subclass that supplies the
        # missing constructor, tests/golden/make_golden_spiral.py); R_in = 0 by the other cylindrical golden vectors.
        self.nr = int(nr); self.nphi = int(nphi); self.nz = int(nz)
        self.dr = float(dr); self.dphi = float(dphi); self.dz = float(dz)
        self.R = float(R)
        self.R_in = float(R_in)
        self.r = self.R_in + (np.arange(self.nr, dtype=np.float64) + 0.5) * self.dr
        self.r_imh = self.r - 0.5 * self.dr
        self.r_iph = self.r + 0.5 * self.dr
        self.r_outer_face = self.r_iph[-1]


class Material:  # :45-50
    def __init__(self, rho, cp, k):
        self.rho = float(rho); self.cp = float(cp); self.k = float(k)

    @property
    def alpha(self):
        return self.k / (self.rho * self.cp)


class Params:  # :52-54
    def __init__(self, dt, theta=0.5, scheme="be"):
        self.dt = float(dt); self.theta = float(theta); self.scheme = str(scheme).lower()


class RobinR:  # :56-58
    def __init__(self, h, T_inf):
        self.h = float(h); self.T_inf = float(T_inf)


class ZBC:  # :60-68
    def __init__(self, kind_bot='neumann0', kind_top='robin', h_bot=0.0, h_top=0.0,
                 T_inf_bot=20.0, T_inf_top=20.0, T_bot=20.0, T_top=20.0):
        self.kind_bot = kind_bot; self.kind_top = kind_top
        self.h_bot = float(h_bot); self.h_top = float(h_top)
        self.T_inf_bot = float(T_inf_bot); self.T_inf_top = float(T_inf_top)
        self.T_bot = float(T_bot); self.T_top = float(T_top)


def thomas_batch(a, b, c, d):
    """adi3d_cyl_phi_v3.py:71-87 (normalised c', d' form over rows of (M, n) arrays)"""
    m, n = d.shape
    x = np.empty_like(d); cp = np.empty_like(c); dp = np.empty_like(d)
    cp[:, 0] = c[:, 0] / b[:, 0]
    dp[:, 0] = d[:, 0] / b[:, 0]
    for i in range(1, n):
        denom = b[:, i] - a[:, i] * cp[:, i - 1]
        cp[:, i] = np.where(i < n - 1, c[:, i] / denom, 0.0)
        dp[:, i] = (d[:, i] - a[:, i] * dp[:, i - 1]) / denom
    x[:, n - 1] = dp[:, n - 1]
    for i in range(n - 2, -1, -1):
        x[:, i] = dp[:, i] - cp[:, i] * x[:, i + 1]
    return x


def r_coefficients(grid, mat, dt, theta, robin_r):
    """Per-radius (a_i, b_i, c_i) and the Robin RHS increment of build_coeff_r,
    adi3d_cyl_phi_v3.py:155-202.  Returns (a, b, c, rhs_add_last), vectors of length nr."""
    nr = grid.nr
    alpha, dr = mat.alpha, grid.dr
    r_i = np.maximum(grid.r, 1e-15)
    r_imh = np.maximum(grid.r_imh, 1e-15)
    r_iph = grid.r_iph
    fac = theta * alpha * dt
    a = np.zeros(nr); b = np.zeros(nr); c = np.zeros(nr)
    ai = -fac * (r_imh[1:-1] / (r_i[1:-1] * dr * dr))
    ci = -fac * (r_iph[1:-1] / (r_i[1:-1] * dr * dr))
    a[1:-1] = ai; b[1:-1] = 1.0 - (ai + ci); c[1:-1] = ci
    a[0] = 0.0
    c0 = -fac * (r_iph[0] / (r_i[0] * dr * dr))
    b[0] = 1.0 - c0; c[0] = c0
    h = float(robin_r.h)
    aN = -fac * (r_imh[-1] / (r_i[-1] * dr * dr))
    bN = 1.0 + fac * (r_imh[-1] / (r_i[-1] * dr * dr))
    add = 0.0
    if h != 0.0:
        bN += fac * (r_iph[-1] * (h / mat.k)) / (r_i[-1] * dr)
        add = fac * (r_iph[-1] * (h / mat.k)) / (r_i[-1] * dr) * robin_r.T_inf
    a[-1] = aN; b[-1] = bN; c[-1] = 0.0
    return a, b, c, add


def build_coeff_r(grid, mat, dt, theta, robin_r, rhs):
    """adi3d_cyl_phi_v3.py:155-202: rows = (phi, z) pairs, columns = r."""
    nr, nphi, nz = grid.nr, grid.nphi, grid.nz
    M = nphi * nz
    av, bv, cv, add = r_coefficients(grid, mat, dt, theta, robin_r)
    a = np.tile(av, (M, 1)); b = np.tile(bv, (M, 1)); c = np.tile(cv, (M, 1))
    rhs_r = np.moveaxis(rhs, 0, -1).reshape(M, nr).astype(np.float64, copy=True)
    if float(robin_r.h) != 0.0:
        rhs_r[:, -1] += add
    return a, b, c, rhs_r


def phi_solve_spectral(Tin, grid, mat, theta, dt):
    """adi3d_cyl_phi_v3.py:302-329"""
    nr, nphi, nz = Tin.shape
    if nphi == 1:
        return Tin.copy()
    r = grid.r.copy()
    dphi = grid.dphi
    alpha = mat.alpha
    fac = np.zeros(nr, dtype=np.float64)
    for ir in range(1, nr):
        fac[ir] = theta * alpha * dt / (r[ir] * r[ir] * dphi * dphi)
    k = np.arange(nphi // 2 + 1, dtype=np.float64)
    cosk = np.cos(2.0 * np.pi * k / float(nphi))
    lam = 1.0 + 2.0 * fac[:, None] * (1.0 - cosk[None, :])
    F = np.fft.rfft(Tin, axis=1)
    F /= lam[:, :, None]
    return np.fft.irfft(F, n=nphi, axis=1)


def build_coeff_z(grid, mat, dt, theta, zbc, rhs):
    """adi3d_cyl_phi_v3.py:255-298"""
    nr, nphi, nz = grid.nr, grid.nphi, grid.nz
    alpha, dz = mat.alpha, grid.dz
    M = nr * nphi
    a = np.zeros((M, nz)); b = np.zeros((M, nz)); c = np.zeros((M, nz))
    fac = theta * alpha * dt / (dz * dz)
    a[:, 1:-1] = -fac; b[:, 1:-1] = 1.0 + 2.0 * fac; c[:, 1:-1] = -fac
    d = rhs.reshape(M, nz).astype(np.float64, copy=True)
    if zbc.kind_bot == 'neumann0':
        a[:, 0] = 0.0; b[:, 0] = 1.0 + fac; c[:, 0] = -fac
    elif zbc.kind_bot == 'dirichlet':
        a[:, 0] = 0.0; b[:, 0] = 1.0; c[:, 0] = 0.0
        d[:, 0] = zbc.T_bot
    elif zbc.kind_bot == 'robin':
        beta = zbc.h_bot / mat.k
        a[:, 0] = 0.0; b[:, 0] = 1.0 + fac * (1.0 + beta * dz); c[:, 0] = -fac
        d[:, 0] += (theta * alpha * dt) * (beta / dz) * zbc.T_inf_bot
    else:
        raise ValueError("unknown zbc.kind_bot")
    if zbc.kind_top == 'neumann0':
        a[:, -1] = -fac; b[:, -1] = 1.0 + fac; c[:, -1] = 0.0
    elif zbc.kind_top == 'dirichlet':
        a[:, -1] = 0.0; b[:, -1] = 1.0; c[:, -1] = 0.0
        d[:, -1] = zbc.T_top
    elif zbc.kind_top == 'robin':
        beta = zbc.h_top / mat.k
        a[:, -1] = -fac; b[:, -1] = 1.0 + fac * (1.0 + beta * dz); c[:, -1] = 0.0
        d[:, -1] += (theta * alpha * dt) * (beta / dz) * zbc.T_inf_top
    else:
        raise ValueError("unknown zbc.kind_top")
    return a, b, c, d


def adi_step(Tn, grid, mat, prm, robin_r, zbc, S=None, theta=None, return_stages=False):
    """adi3d_cyl_phi_v3.py:332-350, backward-Euler branch only."""
    if prm.scheme != "be":
        raise NotImplementedError("oracle restates scheme='be' only (reference 'douglas' is broken, SURVEY D2)")
    dt = prm.dt
    R0 = Tn + (dt * (S / (mat.rho * mat.cp)) if S is not None else 0.0)
    nr, nphi, nz = grid.nr, grid.nphi, grid.nz
    a, b, c, d = build_coeff_r(grid, mat, dt, 1.0, robin_r, R0.copy())
    X = thomas_batch(a, b, c, d)
    TR = np.moveaxis(X.reshape(nphi, nz, nr), -1, 0)
    Tphi = phi_solve_spectral(TR, grid, mat, 1.0, dt)
    a, b, c, d = build_coeff_z(grid, mat, dt, 1.0, zbc, Tphi.copy())
    Tnp1 = thomas_batch(a, b, c, d).reshape(nr, nphi, nz)
    if return_stages:
        return Tnp1, dict(R0=np.array(R0), TR=np.array(TR), Tphi=np.array(Tphi), W=Tnp1)
    return Tnp1


def adi_step_masked(Tn, grid, mat, prm, robin_outer, zbc, active, robin_inner=None, robin_void=None):
    """quick_spiral_deposition_gif_v5.py:31-70"""
    robin_inner = robin_inner or robin_outer
    robin_void = robin_void or robin_outer
    T_work = np.array(Tn, copy=True)
    ambient_inner = float(robin_inner.T_inf)
    ambient_void = float(robin_void.T_inf)
    void_mask = ~active
    if np.any(void_mask):
        T_work[void_mask] = ambient_void
    Tnp1 = adi_step(T_work, grid, mat, prm, robin_outer, zbc)
    if np.any(void_mask):
        Tnp1[void_mask] = ambient_void
    axis_mask = (~active[0])
    if np.any(axis_mask):
        Tnp1[0, axis_mask] = ambient_inner
    return Tnp1
