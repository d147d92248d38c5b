#!/usr/bin/env python3
"""BASELINE.json configs[3]: cylindrical (r, phi, z) 128 x 256 x 512 BE step on one GPU -- per-step time and GB/s
against the 48 B/cell/step algorithmic traffic, plus parity vs the NumPy oracle on a shrink."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_cyl as cyl


def main():
    nr, nphi, nz = 128, 256, 512
    g = cyl.GridCyl(nr, nphi, nz, 2.5e-4, 2 * np.pi / nphi, 2.5e-4, 0.032)
    mat = cyl.Material(7800.0, 490.0, 54.0)
    prm = cyl.Params(0.05, 1.0, "be")
    rr = cyl.RobinR(400.0, 20.0)
    zbc = cyl.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0)
    T0 = np.full((nr, nphi, nz), 20.0); T0[:, :, -16:] = 1000.0
    T = cyl.to_device(T0)
    for _ in range(3):
        T = cyl.adi_step(T, g, mat, prm, rr, zbc)
    torch.cuda.synchronize()
    K = 50
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        T = cyl.adi_step(T, g, mat, prm, rr, zbc)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / K
    N = nr * nphi * nz
    print('cylindrical 128x256x512 BE: %.3f ms/step, %.1f steps/s, %.0f GB/s of 48 B/cell' % (ms, 1e3 / ms, 48 * N / ms / 1e6))
    # the same loop replayed from a HIP graph (StagedCylStepper.run)
    st = cyl.StagedCylStepper(g, mat, prm, rr, zbc)
    for use_graph in (False, True):
        st.run(T, 4, graph=use_graph)
        torch.cuda.synchronize()
        e0.record()
        Tg = st.run(T, 200, graph=use_graph)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print('  run(nsteps=200, graph=%s): %.3f ms/step, %.1f steps/s, %.0f GB/s of 48 B/cell' % (use_graph, ms, 1e3 / ms, 48 * N / ms / 1e6))


if __name__ == '__main__':
    main()
