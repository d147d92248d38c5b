"""perf_map.py restricted to the short and medium line lengths (config-2-sized grids): Gcell/s of each stage kernel.
    [ADI_HIP_LIB=...] python scripts/perf_map_small.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp); dx = 5e-4
print('library:', os.environ.get('ADI_HIP_LIB', 'in-tree'))
print('%-18s %-6s %s' % ('shape', 'fused', 'Gcell/s per stage'))
for n in (128, 160, 192, 224, 256):
    for ax in range(3):
        shape = [256, 256, 256]
        shape[ax] = n
        shape = tuple(shape)
        grid = adi.Grid3D(*shape, dx, np.ones(shape, bool))
        prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
        packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
        st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
        T = adi.to_device(np.full(shape, 300.0))
        nst = len(st.stage_names)
        for _ in range(4):
            T = st.step(T)
        K = 12
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
        for s in range(K):
            T = st.step(T, events=ev[s])
        torch.cuda.synchronize()
        ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
        N = shape[0] * shape[1] * shape[2]
        print('%-18s %-6s %s   step %.4f ms' % (shape, st.fused, '  '.join('%s %.0f' % (nm.replace('sweep_', '').replace('explicit', 'ex'), N / m / 1e6)
                                                           for nm, m in zip(st.stage_names, ms)), ms.sum()), flush=True)
        del grid, packs, st, T
