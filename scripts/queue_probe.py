"""share of FAST units that are still queued to the GENERAL kernels, per axis, on the synthetic WAAM head
(256 x 256 x 320), on an ellipsoid with an inner void and on thin- and thick-walled tubes, with the time per sweep.  python scripts/queue_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import waam

shape = (256, 256, 320); dx = 1e-3
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
g3 = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
ell = ((g3[0] / 0.46) ** 2 + (g3[1] / 0.42) ** 2 + (g3[2] / 0.47) ** 2 <= 1.0) & \
      ~(((g3[0] - 0.1) / 0.12) ** 2 + (g3[1] / 0.15) ** 2 + ((g3[2] + 0.05) / 0.1) ** 2 <= 1.0)
rr = np.sqrt(g3[0] ** 2 + g3[1] ** 2) * 256


def tube(wall):
    """tube along z, outer radius 100 voxels"""
    return (rr <= 100.0) & (rr >= 100.0 - wall)


def ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, mask in (('head', waam.synthetic_head_mask(*shape)), ('ellipsoid with void', ell),
                   ('tube, 4-voxel wall', tube(4)), ('tube, 12-voxel wall', tube(12)), ('tube, 24-voxel wall', tube(24))):
    grid = adi.Grid3D(*shape, dx, mask); prm = adi.Params(200.0 * dx * dx / alpha, 1.0)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=40.0)
    L = grid.layout
    A = L.empty(); A.copy_(torch.from_numpy(np.where(mask, 500.0, 20.0))); B = L.empty()
    _, work, wb = grid.scratch(2)
    out = []
    for ax in range(3):
        if ax == 0:
            adi._explicit_sweep0_into(A, B, grid, mat, prm, packs[0], 20.0)
        else:
            adi._sweep_into(ax, A, B, grid, mat, prm, packs[ax], 20.0)
        torch.cuda.synchronize()
        if ax == 0:
            t = ms(lambda: adi._explicit_sweep0_into(A, B, grid, mat, prm, packs[0], 20.0))
        else:
            t = ms(lambda: adi._sweep_into(ax, A, B, grid, mat, prm, packs[ax], 20.0))
        q = int(work[:4].view(torch.int32)[0])
        n_units = {0: 256 * 320 // 16, 1: 256 * 320 // 16, 2: 256 * 256 // 2}[ax]
        out.append('axis %d: %d of %d (%.1f%%) %.3f ms' % (ax, q, n_units, 100.0 * q / n_units, t))
    print('%-20s' % name, '; '.join(out), flush=True)
