"""Physics regression with asserted tolerances, independent of the oracle (the reference's
quick_compare_* scripts only plot their analytic comparisons).  Runs the same checks on the CPU oracle (always) and
on the HIP backend (GPU)."""
import math

import numpy as np
import pytest

STEEL = dict(rho=7800.0, cp=490.0, k=54.0)
ALPHA = STEEL['k'] / (STEEL['rho'] * STEEL['cp'])


def _erfc_case(api, to_state=lambda x: x, to_host=lambda x: np.asarray(x)):
    """Dirichlet step on the face z = 0 of an insulated bar (quick_compare_dirichlet_robin.py without side losses):
    T(z, t) = T0 + (Ts - T0) erfc(z / (2 sqrt(alpha t))) while the far end is not reached."""
    nx, ny, nz = 4, 4, 256
    dx = 2.5e-4
    mask = np.ones((nx, ny, nz), bool)
    dm = np.zeros_like(mask); dm[:, :, 0] = True; dm[:, :, -1] = True
    dv = np.zeros(mask.shape); dv[:, :, 0] = 1000.0; dv[:, :, -1] = 20.0
    grid = api.Grid3D(nx, ny, nz, dx, mask)
    mat = api.Material(**STEEL)
    dt = 0.5 * dx * dx / ALPHA
    prm = api.Params(dt, 0.5)
    packs = api.precompute_coeff_packs_unified(grid, mat, dir_mask=dm, dir_value=dv)
    T = to_state(np.full(mask.shape, 20.0))
    step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
    nsteps = 400
    for _ in range(nsteps):
        T = step(T, grid, mat, prm, packs, Tinf=20.0)
    T = to_host(T)
    t = nsteps * dt
    z = np.arange(nz) * dx            # the Dirichlet cell centre is the surface
    ana = 20.0 + 980.0 * np.array([math.erfc(v / (2.0 * math.sqrt(ALPHA * t))) for v in z])
    prof = T[2, 2, :]
    assert np.allclose(T, T[:1, :1, :], rtol=0, atol=1e-9)        # 1-D problem stays 1-D
    err = np.max(np.abs(prof - ana)) / 980.0
    assert err < 5e-3, err                                          # second-order scheme, 256 cells
    return err


def _linear_steady_state(api, to_state=lambda x: x, to_host=lambda x: np.asarray(x)):
    """a linear profile between two Dirichlet planes is an exact fixed point of the discrete scheme"""
    shape = (6, 5, 40)
    mask = np.ones(shape, bool)
    dm = np.zeros(shape, bool); dm[:, :, 0] = True; dm[:, :, -1] = True
    dv = np.zeros(shape); dv[:, :, 0] = 1000.0; dv[:, :, -1] = 20.0
    lin = np.broadcast_to(np.linspace(1000.0, 20.0, shape[2]), shape).copy()
    grid = api.Grid3D(*shape, 1e-3, mask)
    mat = api.Material(**STEEL)
    prm = api.Params(50.0 * 1e-6 / ALPHA, 0.5)
    packs = api.precompute_coeff_packs_unified(grid, mat, dir_mask=dm, dir_value=dv)
    step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
    T = to_state(lin)
    for _ in range(5):
        T = step(T, grid, mat, prm, packs, Tinf=20.0)
    assert np.max(np.abs(to_host(T) - lin)) < 1e-9


def _energy_balance(api, to_state=lambda x: x, to_host=lambda x: np.asarray(x)):
    """insulated body heated through one face by a Neumann flux q for time t gains exactly q*A*t of heat
    (sum over cells of rho cp dx^3 dT), the discrete scheme being conservative"""
    shape = (10, 12, 14)
    dx = 1e-3
    mask = np.ones(shape, bool)
    grid = api.Grid3D(*shape, dx, mask)
    mat = api.Material(**STEEL)
    prm = api.Params(30.0 * dx * dx / ALPHA, 0.5)
    q = 2.5e5
    packs = api.precompute_coeff_packs_unified(grid, mat, neumann={'z-': q})
    step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
    T = to_state(np.full(shape, 20.0))
    n = 6
    for _ in range(n):
        T = step(T, grid, mat, prm, packs, Tinf=20.0)
    gained = (to_host(T) - 20.0).sum() * STEEL['rho'] * STEEL['cp'] * dx ** 3
    expect = q * (shape[0] * shape[1] * dx * dx) * n * prm.dt
    # the factorised (ADI) operator adds O(dt^2) cross terms, so the balance holds to that order, not to rounding
    assert abs(gained - expect) / expect < 2e-2, (gained, expect)


@pytest.mark.parametrize('check', [_erfc_case, _linear_steady_state, _energy_balance])
def test_physics_on_oracle(check):
    from oracle import adi_oracle as orc
    check(orc)


@pytest.mark.gpu
@pytest.mark.parametrize('check', [_erfc_case, _linear_steady_state, _energy_balance])
def test_physics_on_hip(check):
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    check(hip, to_state=hip.to_device, to_host=lambda d: d.get())


# ---- the reference's own analytic cases (quick_compare_dirichlet_robin.py, quick_compare_neumann_robin.py) ------------
# finite-radius cylinder, side Robin, far end at ambient; closed forms in tests/analytic_series.py.  The scripts plot
# these comparisons at nxr = 64, nz = 160; here they are asserted -- on the CPU oracle at nxr = 16 (seconds) and on the
# HIP backend at the scripts' default grid.
def _cylinder_case(api, kind, nxr, nz, bound_dirichlet=3.0, bound_neumann=0.1, **kw):
    import analytic_series as an
    R, h, Tinf, Ts, q0 = 0.02, 500.0, 20.0, 1000.0, 2.0e5
    Bi = h * R / STEEL['k']
    times = np.linspace(0.01, 5.0, 6)                      # the scripts' frame times
    prof, dx = an.run_cylinder(api, kind, nxr, nz, times, STEEL, R=R, h_side=h, Tinf=Tinf, Ts=Ts, q0=q0, **kw)
    errs = []
    for i, tt in enumerate(times):
        if kind == 'dirichlet':
            # the Dirichlet cell centre is the heated surface: cell k sits at z = k dx
            ana = an.dirichlet_step_axis(np.arange(nz) * dx, tt, ALPHA, R, Bi, Ts, Tinf)
        else:
            ana = an.neumann_heating_axis((np.arange(nz) + 0.5) * dx, tt, ALPHA, R, Bi, q0, STEEL['k'], Tinf)
        errs.append(float(np.abs(prof[i] - ana).max()))
    if kind == 'dirichlet':
        # t >= 1 s: the thermal layer sqrt(alpha t) spans several cells (at t = 0.01 s it is a third of a cell at nxr = 16)
        assert max(errs[1:]) <= bound_dirichlet, errs          # degrees C of a 980 degree step
        assert prof.max() <= Ts + 1e-6 and prof.min() >= Tinf - 1e-6
    else:
        assert max(errs) <= bound_neumann, errs                 # degrees C of a ~33 degree rise
        assert prof[-1, 0] > Tinf + 25.0                        # q0 > 0 heats the body
    return errs


@pytest.mark.parametrize('kind', ['dirichlet', 'neumann'])
def test_cylinder_series_on_oracle(kind):
    from oracle import adi_oracle as orc
    _cylinder_case(orc, kind, 16, 40, bound_dirichlet=3.5, bound_neumann=0.25)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['dirichlet', 'neumann'])
def test_cylinder_series_on_hip_default_grid(kind):
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    _cylinder_case(hip, kind, 64, 160, bound_dirichlet=2.0, bound_neumann=0.1, to_state=hip.to_device, to_host=lambda d: d.get())


@pytest.mark.gpu
def test_perimeter_ratio_correction_improves_the_neumann_case():
    """quick_compare_layer_birth_robin_v3.py:95-112: the staircase cylinder has 4/pi times the true lateral area; scaling
    h_side by gamma = true / digital perimeter (exposed faces counted on the device) brings the late-time error of the
    Neumann case down by an order of magnitude"""
    import analytic_series as an
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    R, nxr = 0.02, 64
    dx = R / nxr
    _, m2 = an.cylinder_mask(2 * nxr, 2 * nxr, 1, dx, R)
    gamma = hip.perimeter_ratio(m2, dx, 2.0 * math.pi * R)
    assert abs(gamma - math.pi / 4.0) < 2e-3, gamma
    raw = _cylinder_case(hip, 'neumann', nxr, 160, to_state=hip.to_device, to_host=lambda d: d.get())
    cor = _cylinder_case(hip, 'neumann', nxr, 160, to_state=hip.to_device, to_host=lambda d: d.get(), h_scale=gamma)
    assert cor[-1] < 0.2 * raw[-1] and cor[-1] < 5e-3, (raw, cor)


# ---- the reference's own (and only) test: spiral deposition on an annular wall against the closed-form series ----------
# /root/reference/tests/test_spiral_vs_analytic.py:123-209.  It fails at the reference's HEAD with a TypeError
# (GridCyl(R_in=...)).  tests/golden/make_golden_spiral.py supplies the missing constructor (r shifted by R_in)
# and runs the reference's own numeric loop and analytic series: golden/cyl_spiral_annulus.npz.  With that the test still
# fails -- mean |err| 74-145 degrees, max 500-850 against its tolerances 60 / 120 (D10; the series keeps
# one radial mode per angular order and overshoots the deposition temperature) -- so what is asserted here is what can be:
#   * the oracle / the HIP backend, driven by the restated deposition loop, reproduce the reference's five fields
#     (GridCyl(R_in=...) + adi_step_masked with a growing active set): <= 1e-10 relative L-inf, masks identical;
#   * the restated series (tests/analytic_series.py) reproduces the reference's analytic maps: <= 1e-9;
#   * numeric vs analytic: the error levels the reference's code produces, as a regression bound (5 % margin).
def _spiral_case(api, **kw):
    import analytic_series as an
    from helpers import golden, rel_linf
    g = golden('cyl', 'spiral_annulus')
    k, rho, cp, Tinf, Tdep, R_in, wall, h_side, h_end, z_back, layer_h, n_layers, nphi, tau, nr = g['params']
    n_layers, nphi, nr = int(n_layers), int(nphi), int(nr)
    mat = dict(rho=rho, cp=cp, k=k)
    times = g['times']
    grid, fields, masks = an.run_spiral_deposition(api, times, mat, Tinf, Tdep, R_in, wall, h_side, h_end, z_back, layer_h,
                                                   n_layers, tau, nr, nphi, **kw)
    assert (grid.nr, grid.nphi, grid.nz) == g['fields'].shape[1:] and np.allclose(grid.r, g['r'], rtol=0, atol=1e-15)
    for i in range(len(times)):
        assert np.array_equal(masks[i], g['active'][i]), i
        assert rel_linf(fields[i], g['fields'][i]) <= 1e-10, (i, rel_linf(fields[i], g['fields'][i]))
    assert fields[-1].max() <= Tdep + 1e-9 and fields[-1].min() >= Tinf - 1e-9          # maximum principle
    return grid, fields, masks, g


def test_spiral_annulus_series_and_oracle_vs_reference():
    import analytic_series as an
    from oracle import cyl_oracle as cyl
    grid, fields, masks, g = _spiral_case(cyl)
    k, rho, cp, Tinf, Tdep, R_in, wall, h_side, h_end, z_back, layer_h, n_layers, nphi, tau, nr = g['params']
    ana = an.SpiralAnnulus(dict(rho=rho, cp=cp, k=k), Tinf, Tdep, R_in, wall, h_side, h_side, h_end, z_back, layer_h,
                           int(n_layers), tau, int(nphi), grid.nz, grid.nphi)
    ir = int(np.abs(grid.r - (R_in + 0.5 * wall)).argmin())
    for i, t in enumerate(g['times']):
        m = ana.map_at(float(t))
        assert np.array_equal(np.isfinite(m), np.isfinite(g['analytic'][i]))
        ok = np.isfinite(m)
        assert np.max(np.abs(m[ok] - g['analytic'][i][ok])) <= 1e-9                     # the restated series = the reference's
        ok = ok & masks[i][ir].T
        if ok.any():
            d = np.abs(fields[i][ir].T - m)[ok]
            mean_ref, max_ref = g['errors'][i]
            assert d.mean() <= 1.05 * mean_ref + 1e-9 and d.max() <= 1.05 * max_ref + 1e-9, (t, d.mean(), d.max())
    assert g['errors'][1:, 0].min() > 60.0          # D10: the reference's own tolerance is out of its own reach


@pytest.mark.gpu
def test_spiral_annulus_on_hip_vs_reference():
    import adi_thermal_fields_amd.adi3d_hip_cyl as hipcyl
    _spiral_case(hipcyl)
