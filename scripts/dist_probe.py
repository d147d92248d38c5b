#!/usr/bin/env python3
"""Per-rank compute cost of the slab-decomposed step on ONE GPU: the communicator is faked (the rank talks to
copies of itself), so the timing contains every kernel a rank of an N-GPU run executes (halo-extended explicit
stage, pass A, interface solve, pass B, local sweeps) but no wire time.  Upper bound on weak-scaling efficiency
= single-domain step time / this time."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import dist_slab


FakeComm = dist_slab.LoopbackComm


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    mat = adi.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    dx = 5e-4
    import os
    prm = adi.Params(float(os.environ.get('PROBE_CFL', '200')) * dx * dx / alpha, 0.5)
    dev = torch.device('cuda')
    T = adi.DeviceField(torch.rand((n, n, n), dtype=torch.float64, device=dev) * 980 + 20)
    st = dist_slab.SlabStepper(np.ones((n, n, n), bool), dx, mat, prm, 20.0, robin_h=500.0,
                               comm=FakeComm(world, world // 2))
    import os
    st._force_exact = bool(int(os.environ.get('PROBE_EXACT', '0')))
    pre = bool(int(os.environ.get('PROBE_PREFETCH', '1')))
    st._allow_fused = bool(int(os.environ.get('PROBE_FUSED', '1')))
    st._keep_r0 = bool(int(os.environ.get('PROBE_KEEP_R0', '1')))
    st._allow_dots = bool(int(os.environ.get('PROBE_DOTS', '1')))
    for _ in range(3):
        T = st.step(T, prefetch_halo=pre)
    torch.cuda.synchronize()
    print('axis-0 interface form:', st.axis0_mode, 'K =', st._a0['K'], 'chunks =', len(st._a0['chunks']), 'fused =', st._a0['fused'], 'dots =', st._a0['dots'])
    K = 10
    nst = len(st.stage_names)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
    t0 = time.perf_counter()
    for s in range(K):
        T = st.step(T, events=ev[s], prefetch_halo=pre)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]).mean(axis=0)
    print('per-rank step %.3f ms (wall); stages %s' % (dt, dict(zip(st.stage_names, np.round(ms, 3)))))


if __name__ == '__main__':
    main()


def parts():
    """time pass A / interface / pass B separately on the full slab (no chunking)"""
    n = 512; world = 8
    mat = adi.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    dx = 5e-4
    prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    st = dist_slab.SlabStepper(np.ones((n, n, n), bool), dx, mat, prm, 20.0, robin_h=500.0, comm=FakeComm(world, 3))
    E = st.engine
    A = dist_slab._interior(st._tmp[0]); B = dist_slab._interior(st._tmp[1])
    A.copy_(torch.rand((n, n, n), dtype=torch.float64, device='cuda') * 900 + 20)
    kappa = mat.k / (mat.rho * mat.cp); gam = kappa * prm.dt / dx ** 2
    nl = n * n
    cond = E.vec(6 * nl); call = E.vec(6 * nl * world); xlo = E.vec(nl); xhi = E.vec(nl)

    def tm(fn, k=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(k):
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    a = tm(lambda: E.condense(0, st.variant, st.Lint, A, st.flags_int, st.packs_int[0], prm.theta, gam, prm.dt, 20.0, cond))
    call.view(world, -1).copy_(cond.view(1, -1).expand(world, -1))
    b = tm(lambda: E.interface(call, world, 3, nl, xlo, xhi))
    c = tm(lambda: E.sweep(0, st.variant, st.Lint, A, st.flags_int, st.packs_int[0], prm.theta, gam, prm.dt, 20.0, B, xlo, xhi))
    print('pass A %.3f ms, interface %.3f ms, pass B %.3f ms' % (a, b, c))


if __name__ == '__main__' and len(sys.argv) > 3:
    parts()
