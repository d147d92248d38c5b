#!/usr/bin/env python3
"""time pass A (adi_sweep_condense along axis 0) on sub-boxes of K planes of a 512^3 slab"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import dist_slab
from scripts.dist_probe import FakeComm

n = 512
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
dx = 5e-4
prm = adi.Params(10.0 * dx * dx / alpha, 0.5)
st = dist_slab.SlabStepper(np.ones((n, n, n), bool), dx, mat, prm, 20.0, robin_h=500.0, comm=FakeComm(8, 3))
E = st.engine
A = dist_slab._interior(st._tmp[0])
A.copy_(torch.rand((n, n, n), dtype=torch.float64, device='cuda') * 900 + 20)
gam = alpha * prm.dt / dx ** 2
nl = n * n
cond = E.vec(6 * nl)


def tm(fn, k=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(k):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)


for K in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 96, 128, 256, 512]:
    L = E.layout(K, n, n, st.Lint.sx)
    pk = tuple(None if t is None else t[:K] for t in st.packs_int[0])
    a = tm(lambda: E.condense(0, st.variant, L, A[:K], st.flags_int[:K], pk, prm.theta, gam, prm.dt, 20.0, cond))
    o = n - K
    pk2 = tuple(None if t is None else t[o:] for t in st.packs_int[0])
    b = tm(lambda: E.condense(0, st.variant, L, A[o:], st.flags_int[o:], pk2, prm.theta, gam, prm.dt, 20.0, cond))
    c = tm(lambda: E.explicit(st.Lext, st._ext_bufs[0], st.flags_ext, dx, prm.dt, alpha, prm.theta, st._tmp[1], 1, K + 1))
    print('K=%4d  condense first-window %.3f ms  last-window %.3f ms  (%.0f GB/s of 9 B/cell)   explicit K planes %.3f ms'
          % (K, a, b, 9.0 * K * nl / (a * 1e-3) / 1e9, c))
