"""Cartesian step with the axis-1 and axis-2 sweeps run IN PLACE on the fused kernel's output (every sweep kernel reads only
the rows it writes) against the shipped ping-pong through two scratch fields, at 256^3 (134 MB per field: one field fits the
256 MB Infinity Cache, two do not) and 512^3 (1 GiB).  Needs a library built with -DADI_ALLOW_INPLACE.
    ADI_HIP_LIB=scripts/_build/libadi_cartip.so python scripts/inplace_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
for n in (128, 192, 256, 512):
    grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
    prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    L = grid.layout
    T = L.empty(); T.copy_(torch.rand((n, n, n), dtype=torch.float64, device='cuda') * 980 + 20)
    A, B, O, P = L.empty(), L.empty(), L.empty(), L.empty()

    def shipped(t, out):
        adi._explicit_sweep0_into(t, B, grid, mat, prm, packs[0], 20.0)
        adi._sweep_into(1, B, A, grid, mat, prm, packs[1], 20.0)
        adi._sweep_into(2, A, out, grid, mat, prm, packs[2], 20.0)

    def inplace(t, out):
        adi._explicit_sweep0_into(t, out, grid, mat, prm, packs[0], 20.0)
        adi._sweep_into(1, out, out, grid, mat, prm, packs[1], 20.0)
        adi._sweep_into(2, out, out, grid, mat, prm, packs[2], 20.0)

    shipped(T, O); inplace(T, P)
    print('n=%d  identical: %s' % (n, bool(torch.equal(O, P))), flush=True)
    for name, fn in (('ping-pong (shipped)', shipped), ('axis 1 / 2 in place', inplace)):
        x, y = O, P
        for _ in range(6):
            fn(x, y); x, y = y, x
        K = 40
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K)]
        torch.cuda.synchronize()
        for s in range(K):
            ev[s][0].record(); fn(x, y); ev[s][1].record(); x, y = y, x
        torch.cuda.synchronize()
        ms = np.array([e[0].elapsed_time(e[1]) for e in ev])
        print('  %-22s %.4f ms/step (median %.4f)' % (name, ms.mean(), np.median(ms)), flush=True)
    del T, A, B, O, P, packs, grid
    torch.cuda.empty_cache()
