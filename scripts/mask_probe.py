"""per-stage step time at n^3 (default 512) for the all-solid box with the hint, the same box WITHOUT the hint (the build of
the strided FAST kernels that carries the surface-segment lanes, every lane uniform) and an ellipsoid (semi-axes 0.47 / 0.49 /
0.48 of the box: 74 % of the axis-0 tiles are crossed by the surface): what a curved solid pays per stage.
    [ADI_HIP_LIB=scripts/_build/libadi_X.so] python scripts/mask_probe.py [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
c = [((np.arange(n) + 0.5) / n - 0.5) / a for a in (0.47, 0.49, 0.48)]
ell = (c[0][:, None, None] ** 2 + c[1][None, :, None] ** 2 + c[2][None, None, :] ** 2) <= 1.0
T0 = np.random.default_rng(1).uniform(20.0, 1000.0, (n, n, n))
print('library:', os.environ.get('ADI_HIP_LIB', 'in-tree'), flush=True)
for name, mask, hint in (('box + hint', np.ones((n, n, n), bool), True), ('box, no hint', np.ones((n, n, n), bool), False),
                         ('ellipsoid', ell, False)):
    grid = adi.Grid3D(n, n, n, dx, mask)
    if not hint:
        grid.all_solid = False
    prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    if not hint:
        grid.all_solid = False
    T = adi.to_device(T0)
    st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
    nst = len(st.stage_names)
    for _ in range(6):
        T = st.step(T)
    K = 30
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
    for s in range(K):
        T = st.step(T, events=ev[s])
    torch.cuda.synchronize()
    ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
    print('%-13s in-mask %5.1f%%  step %.3f ms  %s' % (name, 100.0 * mask.mean(), ms.sum(),
                                                        dict(zip(st.stage_names, np.round(ms, 3)))), flush=True)
    del T, st, packs, grid
    torch.cuda.empty_cache()
