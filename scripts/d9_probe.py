#!/usr/bin/env python3
"""Reference defect D9, reproduced with the CPU oracle alone (no GPU, no reference import): BASELINE.json configs[4] as
SURVEY.md 8(d) writes it -- synthetic head 256 x 256 x 320, layers of 2 planes born at Ts = 1000 into T = 20, Robin h = 40
on every face, theta = 0.5, sub-steps of dt_cap = cfl dx^2 / alpha with cfl = 2000 -- driven through the event loop of
waam_from_stl_v7_mm.py:515-550 (adi_thermal_fields_amd.waam.run_layer_birth) on oracle/adi_oracle_omp.c, the pinned
restatement of adi3d_numba_coeff.py:290-302.  Prints min / max of the in-mask field after every ADI step, the first
step that leaves [Tinf, Ts] by more than 1e-6 and the growth of the overshoot per step from there on.

    python scripts/d9_probe.py [--shape 256 256 320] [--steps 45] [--theta 0.5] > profiles/r03_d9_probe.txt

The maximum principle says a consistent monotone scheme keeps T inside [20, 1000] here; theta = 1 does (run with
--theta 1.0), theta = 0.5 at cfl = 2000 does not: Crank-Nicolson damps nothing at this step size (amplification -> -1) and
the mask, hence the operator, changes at every birth."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adi_thermal_fields_amd import waam          # noqa: E402  (host loop only; no GPU is touched)
from oracle import adi_oracle as orc             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', type=int, nargs=3, default=[256, 256, 320])
ap.add_argument('--steps', type=int, default=45)
ap.add_argument('--theta', type=float, default=0.5)
ap.add_argument('--cfl', type=float, default=2000.0)
a = ap.parse_args()

STEEL = (7800.0, 490.0, 54.0)
Tinf, Ts = 20.0, 1000.0
mask = waam.synthetic_head_mask(*a.shape)
layers = waam.plan_layers(mask, 2)
dx = 1e-3
times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
log = []


class Done(Exception):
    pass


class Probe:
    Grid3D, Material, Params = orc.Grid3D, orc.Material, orc.Params
    precompute_coeff_packs_unified = staticmethod(orc.precompute_coeff_packs_unified)

    @staticmethod
    def adi_step_numba_coeff(T, grid, mat, params, packs, Tinf=0.0):
        W = orc.adi_run(T, grid, mat, params, packs, Tinf=Tinf, nsteps=1, omp=True)
        m = np.asarray(grid.mask)
        log.append((float(W[m].min()), float(W[m].max()), int(m.sum()), float(params.dt)))
        if len(log) >= a.steps:
            raise Done()
        return W


try:
    waam.run_layer_birth(Probe, mask, dx, STEEL, 40.0, Tinf, Ts, a.theta, a.cfl, layers, times, [times[-1]])
except Done:
    pass

alpha = STEEL[2] / (STEEL[0] * STEEL[1])
print('# D9 probe: oracle/adi_oracle_omp.c, head %dx%dx%d (%d of %d cells in the mask), %d layers of 2 planes, theta %g, '
      'cfl %g (dt_cap %.4f s), Ts %g into Tinf %g, Robin h 40' % (*a.shape, int(mask.sum()), mask.size, len(layers), a.theta,
                                                                  a.cfl, a.cfl * dx * dx / alpha, Ts, Tinf))
print('# step   active cells   dt [s]      min T          max T          overshoot beyond [%g, %g]' % (Tinf, Ts))
first, prev = None, None
for i, (lo, hi, nact, dt) in enumerate(log, 1):
    over = max(Tinf - lo, hi - Ts, 0.0)
    note = ''
    if over > 1e-6:
        if first is None:
            first = i
            note = '   <- first step outside the range'
        elif prev and prev > 1e-6:
            note = '   x %.3g per step' % (over / prev)
    prev = over
    print('%6d %14d %10.4f %14.6g %14.6g %14.6g%s' % (i, nact, dt, lo, hi, over, note))
print('# first step outside [%g, %g] by more than 1e-6: %s' % (Tinf, Ts, first))
