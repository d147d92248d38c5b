"""streaming (nt) against plain output stores over cubic grids: where the fields stop fitting the 256 MB Infinity Cache.
    [ADI_HIP_LIB=scripts/_build/libadi_nont.so] python scripts/nt_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp); dx = 5e-4
print('library:', os.environ.get('ADI_HIP_LIB', 'in-tree'))
for shape in ((128,) * 3, (192,) * 3, (256,) * 3, (288, 288, 288), (320,) * 3, (352, 352, 352), (384,) * 3, (448,) * 3, (512,) * 3, (256, 256, 320), (64, 512, 512), (128, 512, 512)):
    grid = adi.Grid3D(*shape, dx, np.ones(shape, bool))
    prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
    T = adi.to_device(np.full(shape, 300.0))
    nst = len(st.stage_names)
    for _ in range(5):
        T = st.step(T)
    K = 16
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
    for s in range(K):
        T = st.step(T, events=ev[s])
    torch.cuda.synchronize()
    ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
    N = shape[0] * shape[1] * shape[2]
    print('%-16s field %6.0f MB  step %.4f ms  %s' % (shape, N * 8 / 1e6, ms.sum(), '  '.join('%s %.0f' % (nm.replace('sweep_', '').replace('explicit', 'ex'), N / m / 1e6) for nm, m in zip(st.stage_names, ms))), flush=True)
    del grid, packs, st, T
    torch.cuda.empty_cache()
