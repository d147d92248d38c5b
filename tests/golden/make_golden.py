#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container, where /root/reference is mounted read-only
(the reference never travels to the GPU box; the committed .npz files do).

    python tests/golden/make_golden.py            # fast cases (seconds..minutes of CPython)
    python tests/golden/make_golden.py --slow     # + config1_64 (64^3 x 100 steps, minutes)

Each <case>.npz holds the case's expected outputs exactly as the reference computed them
(fp64, bit-for-bit) plus the inputs' SHA-256 so a drift of tests/cases.py is caught.
Reference entry points used:
  adi3d_numba_coeff.py : Grid3D, Material, Params, precompute_coeff_packs_unified,
                         exposed_mask, lap1D_x/y/z, sweep_axis0/1/2, adi_step_numba_coeff
                         (its own pure-Python fallback, NUMBA=False in this image)
  adi3d_cyl_phi_v3.py  : GridCyl, Material, Params, RobinR, ZBC, adi_step (scheme="be")
  quick_spiral_deposition_gif_v5.py : adi_step_masked
"""
import argparse
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, '/root/reference')

import cases  # noqa: E402


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        if a is None:
            h.update(b'none')
        elif np.isscalar(a):
            h.update(np.float64(a).tobytes())
        else:
            a = np.ascontiguousarray(a)
            h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def cart_input_hash(c):
    parts = [c['mask'], c['T0'], c['dir_mask'], c['dir_value'], c['dx'], c['dt'], c['theta'], c['Tinf']]
    for spec in (c['neumann'], c['robin_h']):
        if isinstance(spec, dict):
            for f in cases.FACES:
                parts.append(spec.get(f, None) if f in spec else None)
        else:
            parts.append(spec)
    return sha(*parts)


def run_cart(name, stages=False, planes_only=False):
    import adi3d_numba_coeff as ref
    c = cases.cart_case(name)
    nx, ny, nz = c['shape']
    grid = ref.Grid3D(nx, ny, nz, c['dx'], c['mask'])
    mat = ref.Material(**c['mat'])
    prm = ref.Params(c['dt'], c['theta'])

    def packs_for():
        return ref.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                                  neumann=c['neumann'], robin_h=c['robin_h'])
    packs = packs_for()
    out = dict(input_sha=np.array(cart_input_hash(c)))
    for ax, p in zip('xyz', packs):
        out['coeff_' + ax] = p.coeff; out['qflux_' + ax] = p.qflux
    out['dir_mask'] = packs[0].dir_mask; out['dir_val'] = packs[0].dir_val
    for f in cases.FACES:
        out['exposed_' + f] = ref.exposed_mask(grid.mask, f)
    T = np.array(c['T0'], dtype=np.float64)
    if stages:
        kappa = mat.k / (mat.rho * mat.cp); gam = kappa * prm.dt / (grid.dx ** 2)
        out['Lx'] = ref.lap1D_x(T, grid.mask, grid.dx)
        out['Ly'] = ref.lap1D_y(T, grid.mask, grid.dx)
        out['Lz'] = ref.lap1D_z(T, grid.mask, grid.dx)
        R0 = T + prm.dt * kappa * (1.0 - prm.theta) * (out['Lx'] + out['Ly'] + out['Lz'])
        out['R0'] = R0
        sw = (ref.sweep_axis0, ref.sweep_axis1, ref.sweep_axis2)
        cur = R0
        for ax, nm in enumerate('UVW'):
            p = packs[ax]
            cur = sw[ax](T, cur, grid.mask, p.coeff, p.dir_mask, p.dir_val, p.qflux, prm.theta, gam, prm.dt,
                         kappa, c['Tinf'])
            out[nm] = cur
    if name == 'edge_shapes' or name == 'kat2':   # apply_surface_impulse_Q (adi3d_numba_coeff.py:304-320), in place
        for f in cases.FACES:
            Ti = np.array(c['T0'], dtype=np.float64)
            ref.apply_surface_impulse_Q(Ti, grid, mat, 3.5e4, face=f)
            out['impulse_' + f] = Ti
    if c['births'] is None:
        for s in range(c['nsteps']):
            T = ref.adi_step_numba_coeff(T, grid, mat, prm, packs, Tinf=c['Tinf'])
            if s == 0:
                out['T_step1'] = T.copy()
    else:  # layer birth: the loop of waam_from_stl_v7_mm.py:515-550 / activate_layer :487-495
        mask_act = c['mask'].copy()
        for li, (z0, z1) in enumerate([(None, None)] + list(c['births'])):
            if z0 is not None:
                newborn = c['full_mask'].copy(); newborn[:, :, :z0] = False; newborn[:, :, z1:] = False
                newborn &= ~mask_act
                T[newborn] = c['Ts']
                mask_act |= newborn
                grid.mask = mask_act
                packs = packs_for()
            for s in range(c['nsteps']):
                T = ref.adi_step_numba_coeff(T, grid, mat, prm, packs, Tinf=c['Tinf'])
            out['T_layer%d' % li] = T.copy()
    if planes_only:
        out['T_sum'] = np.array(T.sum()); out['T_sha'] = np.array(sha(T))
        out['plane_i'] = T[nx // 2].copy(); out['plane_j'] = T[:, ny // 2].copy(); out['plane_k'] = T[:, :, nz // 4].copy()
    else:
        out['T_final'] = T
    return out


def cyl_input_hash(c):
    return sha(c['T0'], c['S'], c['active'], c['dt'], c['dr'], c['dz'], c['dphi'])


def run_cyl(name):
    import adi3d_cyl_phi_v3 as ref
    c = cases.cyl_case(name)
    nr, nphi, nz = c['shape']
    grid = ref.GridCyl(nr, nphi, nz, c['dr'], c['dphi'], c['dz'], c['R'])
    mat = ref.Material(**c['mat'])
    prm = ref.Params(c['dt'], 1.0, "be")
    rr = ref.RobinR(*c['robin_r'])
    zbc = ref.ZBC(**c['zbc'])
    T = np.array(c['T0'], dtype=np.float64)
    out = dict(input_sha=np.array(cyl_input_hash(c)))
    if c['active'] is not None:
        from quick_spiral_deposition_gif_v5 import adi_step_masked
        ri = ref.RobinR(*c['robin_inner']); rv = ref.RobinR(*c['robin_void'])
    for s in range(c['nsteps']):
        if c['active'] is not None:
            T = adi_step_masked(T, grid, mat, prm, rr, zbc, c['active'], robin_inner=ri, robin_void=rv)
        else:
            T = ref.adi_step(T, grid, mat, prm, rr, zbc, S=c['S'])
        if s == 0:
            out['T_step1'] = T.copy()
    out['T_final'] = T
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--slow', action='store_true')
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    todo = []
    for n in cases.CART_CASES:
        todo.append(('cart_' + n, lambda n=n: run_cart(n, stages=(n == cases.CART_STAGE_CASE))))
    for n in cases.CYL_CASES:
        todo.append(('cyl_' + n, lambda n=n: run_cyl(n)))
    if a.slow:
        for n in cases.CART_SLOW_CASES:
            todo.append(('cart_' + n, lambda n=n: run_cart(n, planes_only=True)))
    for fn, job in todo:
        if a.only and a.only not in fn:
            continue
        out = job()
        np.savez_compressed(os.path.join(HERE, fn + '.npz'), **out)
        print('wrote', fn, {k: getattr(v, 'shape', None) for k, v in list(out.items())[:3]}, flush=True)


if __name__ == '__main__':
    main()
