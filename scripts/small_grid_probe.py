"""time per step of the nsub loop on small grids: plain launches (one Python call per stage) vs StagedStepper.run
(HIP-graph replay).  python scripts/small_grid_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

for n in (32, 64, 128, 256):
    mat = adi.Material(7800.0, 490.0, 54.0); dx = 1e-3
    prm = adi.Params(2.0 * dx * dx / (mat.k / (mat.rho * mat.cp)), 0.5)
    grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    T = adi.to_device(np.random.default_rng(0).uniform(20, 1000, (n, n, n)))
    st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
    K = 200
    for _ in range(5):
        T = adi.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        T = adi.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
    torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / K
    st.run(T, 4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T2 = st.run(T, K)
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / K
    print('%4d^3: plain %.1f us/step, graph %.1f us/step  (x%.2f)' % (n, t_plain * 1e6, t_graph * 1e6, t_plain / t_graph), flush=True)
