"""CPU: pin the oracle (oracle/adi_oracle.c, oracle/cyl_oracle.py) to the golden vectors that
tests/golden/make_golden.py produced by importing the reference, and to the KAT spot values of
SURVEY.md 8(c).  Bit-exact where the evaluation order is restated exactly (all of it)."""
import numpy as np
import pytest

import cases
from helpers import golden, run_cart_case, run_cyl_case
from oracle import adi_oracle as orc
from oracle import cyl_oracle as cyl


def test_kat1_spot_values():
    out = run_cart_case(orc, cases.cart_case('kat1'))
    T = out['T_final']
    assert T[3, 4, 5] == 77.97105340045186
    assert T[0, 0, 0] == 52.57541495954793
    assert T[15, 15, 15] == 128.59241000169195
    assert abs(T.sum() - 177394.2622383918) <= 1e-12 * 177394.2622383918


def test_kat2_spot_values():
    c = cases.cart_case('kat2')
    assert int(c['mask'].sum()) == 2240
    T = run_cart_case(orc, c)['T_final']
    assert T[6, 6, 0] == 157.61408858371772
    assert T[6, 6, 5] == 20.70812408526635
    assert T[1, 6, 0] == 157.0033217250357
    assert T[0, 0, 0] == 20.0
    assert abs(T[c['mask']].sum() - 70957.53852852067) <= 1e-12 * 70957.53852852067


def test_kat3_spot_values():
    T = run_cyl_case(cyl, cases.cyl_case('kat3'))['T_final']
    for idx, v in (((0, 0, 0), 44.28317676851759), ((7, 3, 11), 121.76173891020403),
                   ((4, 8, 6), 35.517735621362995)):
        assert abs(T[idx] - v) <= 1e-12 * abs(v)
    assert abs(T.sum() - 61068.6750615946) <= 1e-12 * 61068.6750615946


@pytest.mark.parametrize('name', cases.CART_CASES)
def test_cart_oracle_bit_exact(name):
    c = cases.cart_case(name)
    g = golden('cart', name)
    out = run_cart_case(orc, c)
    packs = out['packs0']
    for ax, p in zip('xyz', packs):
        assert np.array_equal(p.coeff, g['coeff_' + ax]), 'coeff_' + ax
        assert np.array_equal(p.qflux, g['qflux_' + ax]), 'qflux_' + ax
    assert np.array_equal(packs[0].dir_mask, g['dir_mask'])
    assert np.array_equal(packs[0].dir_val, g['dir_val'])
    for f in cases.FACES:
        assert np.array_equal(orc.exposed_mask(c['mask'], f), g['exposed_' + f]), f
    for key in g.files:
        if key.startswith('T_'):
            assert np.array_equal(out[key], g[key]), key


def test_cart_oracle_stages_bit_exact():
    c = cases.cart_case(cases.CART_STAGE_CASE)
    g = golden('cart', cases.CART_STAGE_CASE)
    grid = orc.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = orc.Material(**c['mat']); prm = orc.Params(c['dt'], c['theta'])
    packs = orc.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    for ax, nm in enumerate(('Lx', 'Ly', 'Lz')):
        assert np.array_equal(orc.lap1D(c['T0'], c['mask'], c['dx'], ax), g[nm]), nm
    assert np.array_equal(orc.explicit_rhs(c['T0'], grid, mat, prm), g['R0'])
    W, st = orc.adi_step_numba_coeff(c['T0'], grid, mat, prm, packs, Tinf=c['Tinf'], return_stages=True)
    for nm in ('R0', 'U', 'V', 'W'):
        assert np.array_equal(st[nm], g[nm]), nm
    # each sweep alone, fed with the reference's own previous stage
    prev = {'U': 'R0', 'V': 'U', 'W': 'V'}
    for ax, nm in enumerate('UVW'):
        got = orc.sweep_axis(ax, g[prev[nm]], grid, mat, prm, packs[ax], Tinf=c['Tinf'])
        assert np.array_equal(got, g[nm]), nm
    assert np.array_equal(W, g['T_step1'])


def test_cart_oracle_config1_64_planes():
    """BASELINE.json configs[0]: 64^3 Dirichlet, 100 steps (reference ran ~5 min of CPython once)."""
    try:
        g = golden('cart', 'config1_64')
    except FileNotFoundError:
        pytest.skip('config1_64 golden not generated')
    c = cases.cart_case('config1_64')
    grid = orc.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = orc.Material(**c['mat']); prm = orc.Params(c['dt'], c['theta'])
    packs = orc.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    T = orc.adi_run(c['T0'], grid, mat, prm, packs, Tinf=c['Tinf'], nsteps=c['nsteps'])
    nx, ny, nz = c['shape']
    assert np.array_equal(T[nx // 2], g['plane_i'])
    assert np.array_equal(T[:, ny // 2], g['plane_j'])
    assert np.array_equal(T[:, :, nz // 4], g['plane_k'])
    assert T.sum() == float(g['T_sum'])


def test_oracle_omp_variant_identical():
    c = cases.cart_case('holes_mixed')
    grid = orc.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = orc.Material(**c['mat']); prm = orc.Params(c['dt'], c['theta'])
    packs = orc.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    a = orc.adi_run(c['T0'], grid, mat, prm, packs, Tinf=c['Tinf'], nsteps=3)
    b = orc.adi_run(c['T0'], grid, mat, prm, packs, Tinf=c['Tinf'], nsteps=3, omp=True)
    assert np.array_equal(a, b)


def test_thomas_vs_dense():
    rng = np.random.default_rng(0)
    n = 37
    a = -rng.uniform(0, 1, n); c = -rng.uniform(0, 1, n); a[0] = 0; c[-1] = 0
    b = 1.0 + np.abs(a) + np.abs(c) + rng.uniform(0, 1, n)
    d = rng.uniform(-5, 5, n)
    A = np.diag(b) + np.diag(a[1:], -1) + np.diag(c[:-1], 1)
    x = orc.thomas_solve(a, b, c, d)
    assert np.allclose(x, np.linalg.solve(A, d), rtol=1e-13, atol=1e-13)


def test_bad_face():
    with pytest.raises(ValueError):
        orc.exposed_mask(np.ones((2, 2, 2), bool), 'w+')


@pytest.mark.parametrize('name', cases.CYL_CASES)
def test_cyl_oracle_bit_exact(name):
    c = cases.cyl_case(name)
    g = golden('cyl', name)
    out = run_cyl_case(cyl, c)
    assert np.array_equal(out['T_step1'], g['T_step1'])
    assert np.array_equal(out['T_final'], g['T_final'])


def test_cyl_bad_zbc_kind():
    c = cases.cyl_case('kat3')
    c['zbc'] = dict(kind_bot='bogus', kind_top='robin')
    with pytest.raises(ValueError):
        run_cyl_case(cyl, c)


@pytest.mark.parametrize('name', ['kat2', 'edge_shapes'])
def test_surface_impulse_oracle(name):
    c = cases.cart_case(name)
    g = golden('cart', name)
    grid = orc.Grid3D(*c['shape'], c['dx'], c['mask']); mat = orc.Material(**c['mat'])
    for f in cases.FACES:
        T = np.array(c['T0'], dtype=np.float64)
        orc.apply_surface_impulse_Q(T, grid, mat, 3.5e4, face=f)
        assert np.array_equal(T, g['impulse_' + f]), f
