"""explicit + axis-0 sweep as two kernels vs the fused kernel (adi_explicit_sweep0) at n^3, lean Robin workload.
    python scripts/fused_probe.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda')
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
dx = 5e-4
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
L = grid.layout
T = L.empty(); T.copy_(torch.rand((n, n, n), dtype=torch.float64, device=dev) * 980 + 20)
A = L.empty(); B = L.empty(); C = L.empty()
kappa = alpha


def two():
    adi.check(adi.lib.adi_explicit_rhs(adi._p(T), adi._p(grid.d_flags), n, n, n, grid.sx, dx, prm.dt, kappa, prm.theta,
                                       adi._p(A), adi._stream()))
    adi._sweep_into(0, A, B, grid, mat, prm, packs[0], 20.0)


def one():
    adi._explicit_sweep0_into(T, C, grid, mat, prm, packs[0], 20.0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t2 = timeit(two); t1 = timeit(one)
print('n=%d: two kernels %.4f ms, fused %.4f ms (%.0f GB/s on 17 B/cell), identical=%s maxdiff=%.3e' % (
    n, t2, t1, 17.03 * n ** 3 / t1 / 1e6, bool(torch.equal(B, C)), float((B - C).abs().max())), flush=True)
