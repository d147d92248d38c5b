"""exact reference call semantics (NumPy in, NumPy out per step) at 512^3: the PCIe-inclusive rate DESIGN.md quotes"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
n = 512; dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
T = np.random.default_rng(0).uniform(20, 1000, (n, n, n))
T = adi.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
t0 = time.perf_counter(); K = 5
for _ in range(K):
    T = adi.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
dt = (time.perf_counter() - t0) / K
print('NumPy in/out: %.1f ms per step = %.1f steps/s (1 GiB up + 1 GiB down per step)' % (dt * 1e3, 1.0 / dt))
