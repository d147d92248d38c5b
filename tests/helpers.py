"""Helpers shared by the parity tests (reference-free)."""
import os

import numpy as np

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(kind, name):
    return np.load(os.path.join(GOLDEN, '%s_%s.npz' % (kind, name)), allow_pickle=False)


def rel_linf(a, b):
    """max|a-b| / max|b|  (the tolerance metric of BASELINE.json / SURVEY.md 8(d))."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b)) if b.size else 1.0
    if den == 0.0:
        den = 1.0
    return (np.max(np.abs(a - b)) / den) if b.size else 0.0


def run_cart_case(api, c, step_fn=None, return_layers=False):
    """Drive a Cartesian case through any module exposing the reference's operator surface
    (oracle.adi_oracle or the HIP host module).  Mirrors tests/golden/make_golden.py:run_cart."""
    nx, ny, nz = c['shape']
    grid = api.Grid3D(nx, ny, nz, c['dx'], c['mask'])
    mat = api.Material(**c['mat'])
    prm = api.Params(c['dt'], c['theta'])
    step = step_fn or getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff

    def packs_for():
        return api.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                                  neumann=c['neumann'], robin_h=c['robin_h'])
    packs = packs_for()
    out = dict(packs0=packs)
    T = np.array(c['T0'], dtype=np.float64)
    if c['births'] is None:
        for s in range(c['nsteps']):
            T = step(T, grid, mat, prm, packs, Tinf=c['Tinf'])
            if s == 0:
                out['T_step1'] = np.array(T)
    else:
        mask_act = c['mask'].copy()
        for li, (z0, z1) in enumerate([(None, None)] + list(c['births'])):
            if z0 is not None:
                newborn = c['full_mask'].copy(); newborn[:, :, :z0] = False; newborn[:, :, z1:] = False
                newborn &= ~mask_act
                T = np.array(T)
                T[newborn] = c['Ts']
                mask_act |= newborn
                grid.mask = mask_act
                packs = packs_for()
            for s in range(c['nsteps']):
                T = step(T, grid, mat, prm, packs, Tinf=c['Tinf'])
            out['T_layer%d' % li] = np.array(T)
    out['T_final'] = np.array(T)
    return out


def run_cyl_case(api, c, masked_fn=None):
    nr, nphi, nz = c['shape']
    grid = api.GridCyl(nr, nphi, nz, c['dr'], c['dphi'], c['dz'], c['R'])
    mat = api.Material(**c['mat'])
    prm = api.Params(c['dt'], 1.0, "be")
    rr = api.RobinR(*c['robin_r'])
    zbc = api.ZBC(**c['zbc'])
    T = np.array(c['T0'], dtype=np.float64)
    out = {}
    for s in range(c['nsteps']):
        if c['active'] is not None:
            ri = api.RobinR(*c['robin_inner']); rv = api.RobinR(*c['robin_void'])
            T = (masked_fn or api.adi_step_masked)(T, grid, mat, prm, rr, zbc, c['active'],
                                                  robin_inner=ri, robin_void=rv)
        else:
            T = api.adi_step(T, grid, mat, prm, rr, zbc, S=c['S'])
        if s == 0:
            out['T_step1'] = np.array(T)
    out['T_final'] = np.array(T)
    return out
