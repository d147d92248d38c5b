#!/usr/bin/env python3
"""Golden vectors of the reference's only test scenario -- spiral deposition on an annular wall,
/root/reference/tests/test_spiral_vs_analytic.py:123-209 -- produced by IMPORTING THE REFERENCE in the build container
(it never travels; tests/golden/cyl_spiral_annulus.npz does).

    python tests/golden/make_golden_spiral.py

The test fails at the reference's HEAD before it computes anything: build_grid_annular passes R_in= to GridCyl, whose
constructor has no such parameter (quick_spiral_deposition_gif_v5.py:80 vs adi3d_cyl_phi_v3.py:34).  The one
thing added here is the constructor the call site asks for -- a subclass of the reference's GridCyl that accepts R_in and
shifts the cell radii by it; every formula that then runs (build_coeff_r, thomas_batch, phi_solve_spectral, build_coeff_z,
adi_step, adi_step_masked, the deposition loop _run_numeric_simulation of the test file, and the analytic series of
spiral_analytic_solution.py) is the reference's own code.  Stored: the five (nr, nphi, nz) fields and active masks of the
numeric side, the five (nz, nphi) analytic maps, and the mean / max |numeric - analytic| per time over the existing cells.
Those errors are 74-145 / 500-850 degrees: the reference's tolerances (60 / 120, :192-193) are not met by the reference
itself once it runs (D10 in DESIGN.md section 6)."""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/tests')

import adi3d_cyl_phi_v3 as core  # noqa: E402
import quick_spiral_deposition_gif_v5 as drv  # noqa: E402
import spiral_analytic_solution as ana  # noqa: E402


class AnnularGridCyl(core.GridCyl):
    """GridCyl(..., R_in=...) as quick_spiral_deposition_gif_v5.py:80 calls it"""

    def __init__(self, nr, nphi, nz, dr, dphi, dz, R, R_in=0.0):
        super().__init__(nr, nphi, nz, dr, dphi, dz, R)
        self.R_in = float(R_in)
        self.r = self.R_in + (np.arange(self.nr, dtype=np.float64) + 0.5) * self.dr
        self.r_imh = self.r - 0.5 * self.dr
        self.r_iph = self.r + 0.5 * self.dr
        self.r_outer_face = self.r_iph[-1]


drv.GridCyl = AnnularGridCyl
import test_spiral_vs_analytic as ref_test  # noqa: E402  (binds build_grid_annular from the patched module)

# the parameters of test_spiral_numeric_matches_analytic, :124-161
k, rho, cp, T_inf, T_dep = 54.0, 7800.0, 490.0, 20.0, 900.0
R_in, wall, h_side, h_end = 0.03, 0.002, 400.0, 500.0
z_back, layer_h, n_layers, nphi, tau = 0.02, 0.004, 2, 36, 2.0
times = np.linspace(0.0, tau * n_layers, 5)
cfg_numeric = dict(R_out=R_in + wall, wall_thickness=wall, height=layer_h * n_layers, z_back=z_back, nr=6, nphi=nphi,
                   dz_override=layer_h, rho=rho, cp=cp, k=k, h_side=h_side, h_end=h_end, T_inf=T_inf, T_deposit=T_dep,
                   h_void=h_side, layer_cells=1, n_layers=n_layers, loops_per_layer=1, dt=tau / nphi,
                   omega=2.0 * math.pi / tau)
grid, snaps, actives = ref_test._run_numeric_simulation(times, cfg_numeric)
cfg = ana.SpiralAnalyticConfig(k=k, rho=rho, cp=cp, T_inf=T_inf, T_deposit=T_dep, inner_radius=R_in, wall_thickness=wall,
                               h_inner=h_side, h_outer=h_side, h_end=h_end, base_height=z_back, layer_height=layer_h,
                               n_layers=n_layers, tau_dep=tau, n_phi_depo=nphi, z_back=z_back, z_front=layer_h * n_layers,
                               Nz=grid.nz, Nphi=grid.nphi, M_ang=12, Nr_modes=6)
cache = ana.build_cache(cfg)
maps = [ana.temperature_phi_z_at_time(cfg, cache, float(t))[0] for t in times]
ir = int(np.abs(grid.r - cfg.probe_r).argmin())
errs = []
for T, act, m in zip(snaps, actives, maps):
    ok = np.isfinite(m) & act[ir].T
    d = np.abs(T[ir].T - m)[ok]
    errs.append((float(d.mean()), float(d.max())) if ok.any() else (0.0, 0.0))
np.savez_compressed(os.path.join(HERE, 'cyl_spiral_annulus.npz'), times=times, fields=np.array(snaps), active=np.array(actives),
                    analytic=np.array(maps), errors=np.array(errs), r=grid.r,
                    params=np.array([k, rho, cp, T_inf, T_dep, R_in, wall, h_side, h_end, z_back, layer_h, n_layers, nphi, tau, 6]))
print('shape', np.array(snaps).shape, 'errors (mean, max) per time:', np.round(np.array(errs), 2).tolist())
