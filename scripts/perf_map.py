"""Gcell/s of each stage kernel of the step over line lengths (all-solid lean workload, ~16-32 M cells), to spot
tiling cliffs.  python scripts/perf_map.py [NXxNYxNZ ...]   (no arguments: one axis at a time from 40 to 1024 rows)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp); dx = 5e-4
def default_shapes():
    for n in (40, 64, 96, 128, 160, 192, 200, 256, 320, 384, 448, 512, 640, 768, 1024):
        for ax in range(3):
            shape = [256, 256, 256]
            shape[ax] = n
            if n > 512:
                shape[(ax + 1) % 3] = 128
            yield tuple(shape)


shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or list(default_shapes())
print('%-18s %-6s %s' % ('shape', 'fused', 'Gcell/s per stage'))
for shape in shapes:
    if True:
        grid = adi.Grid3D(*shape, dx, np.ones(shape, bool))
        prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
        packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
        st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
        T = adi.to_device(np.full(shape, 300.0))
        nst = len(st.stage_names)
        for _ in range(4):          # (the no-fallback promise is learnt on the third step: one host synchronisation)
            T = st.step(T)
        K = 8
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
        for s in range(K):
            T = st.step(T, events=ev[s])
        torch.cuda.synchronize()
        ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
        N = shape[0] * shape[1] * shape[2]
        print('%-18s %-6s %6.3f ms  %s' % (shape, st.fused, ms.sum(), '  '.join('%s %.0f' % (nm.replace('sweep_', '').replace('explicit', 'ex'), N / m / 1e6)
                                                           for nm, m in zip(st.stage_names, ms))), flush=True)
        del grid, packs, st, T
