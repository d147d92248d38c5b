"""Padded extents (Layout / adi_recommended_dims): units the FAST kernels still queue to the GENERAL ones and the time per
sweep on all-solid LOGICAL boxes of ragged size.  python scripts/pad_probe.py [NXxNYxNZ ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

dx = 1e-3
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)


def ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(250, 250, 250), (257, 257, 257), (300, 300, 300)]
for shape in shapes:
    grid = adi.Grid3D(*shape, dx, np.ones(shape, bool)); prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    L = grid.layout
    A = L.to_layout(np.full(shape, 300.0), torch.float64); B = L.empty()
    _, work, wb = grid.scratch(2)
    out = []
    for ax in range(3):
        run = (lambda: adi._explicit_sweep0_into(A, B, grid, mat, prm, packs[0], 20.0)) if ax == 0 and adi.fused_supported(grid) \
            else (lambda: adi._sweep_into(ax, A, B, grid, mat, prm, packs[ax], 20.0))
        run(); torch.cuda.synchronize()
        q = int(work[:4].view(torch.int32)[0])
        out.append('axis %d: %d queued, %.3f ms' % (ax, q, ms(run)))
    print(shape, '->', L.pd[:3], '; '.join(out), flush=True)
