"""step time on the synthetic WAAM head (BASELINE configs[4] geometry, all layers born) against the all-solid box of the
same size: how much of the realistic workload runs through the GENERAL (surface) kernels.
    python scripts/head_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import waam

shape = (256, 256, 320)
dx = 1e-3
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
for name, mask in (('all-solid box', np.ones(shape, bool)), ('synthetic head', waam.synthetic_head_mask(*shape))):
    grid = adi.Grid3D(*shape, dx, mask)
    prm = adi.Params(200.0 * dx * dx / alpha, 1.0)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=40.0)
    T = adi.to_device(np.where(mask, 500.0, 20.0))
    st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
    nst = len(st.stage_names)
    for _ in range(3):
        T = st.step(T)
    K = 20
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
    for s in range(K):
        T = st.step(T, events=ev[s])
    torch.cuda.synchronize()
    ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
    print('%-15s in-mask %.1f%%  step %.3f ms  %s  bytes/cell %s' % (
        name, 100.0 * mask.mean(), ms.sum(), dict(zip(st.stage_names, np.round(ms, 3))),
        [round(p.bytes_per_cell, 2) for p in packs]), flush=True)
