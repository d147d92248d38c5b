"""the three general-pack (42 B/cell, dense) sweeps at n^3: ms and fraction of the 8 TB/s peak"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
L = grid.layout
T = L.empty(); T.copy_(torch.rand((n, n, n), dtype=torch.float64, device='cuda') * 980 + 20)
out = L.empty()
for ax in (0, 1, 2):
    ms = []
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for it in range(13):
        e0.record(); st.sweep_into(ax, T, out, variant=_lib.SWEEP_GENERAL, dense=True); e1.record(); e1.synchronize()
        if it >= 3: ms.append(e0.elapsed_time(e1))
    m = float(np.mean(ms))
    print('axis %d general dense: %.4f ms  %.0f GB/s  frac %.3f' % (ax, m, 42 * n ** 3 / m / 1e6, 42 * n ** 3 / m / 1e6 / 8000), flush=True)
