"""Does chunking the in-plane sweeps (axes 1 and 2) over groups of planes keep the intermediate field in the 256 MiB
Infinity Cache?  For each chunk of planes: sweep1 chunk -> small scratch, sweep2 scratch -> out chunk.  Launches are
replayed from a HIP graph so the host does not bound the measurement.
    python scripts/mall_probe.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import dist_slab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda')
mat = adi.Material(7800.0, 490.0, 54.0)
alpha = mat.k / (mat.rho * mat.cp)
dx = 5e-4
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
L = grid.layout
E = dist_slab.HipEngine()
gam = alpha * prm.dt / dx ** 2
v = packs[0].variant
fl = grid.d_flags
pk = [(p.d_coeff, p.d_dir_mask, p.d_dir_val, p.d_qflux) for p in packs]
B = L.empty(); B.copy_(torch.rand((n, n, n), dtype=torch.float64, device=dev) * 980 + 20)
A = L.empty(); O = L.empty(); O2 = L.empty()
E._workspace(L)


def cut(t, p0, p1):
    return None if t is None else t[p0:p1]


def full():
    E.sweep(1, v, L, B, fl, pk[1], prm.theta, gam, prm.dt, 20.0, A)
    E.sweep(2, v, L, A, fl, pk[2], prm.theta, gam, prm.dt, 20.0, O)


def chunked(c, scratch_small=True):
    def run():
        for p0 in range(0, n, c):
            p1 = min(n, p0 + c)
            Lc = E.layout(p1 - p0, n, n, L.sx)
            S = A[0:p1 - p0] if scratch_small else A[p0:p1]
            E.sweep(1, v, Lc, B[p0:p1], fl[p0:p1], tuple(cut(t, p0, p1) for t in pk[1]), prm.theta, gam, prm.dt, 20.0, S)
            E.sweep(2, v, Lc, S, fl[p0:p1], tuple(cut(t, p0, p1) for t in pk[2]), prm.theta, gam, prm.dt, 20.0, O2[p0:p1])
    return run


def timeit(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(3):
            g.replay()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_full = timeit(full)
print('full sweeps 1+2: %.4f ms' % t_full, flush=True)
for c in (4, 8, 16, 32, 64, 128, 256):
    for small in (True, False):
        t = timeit(chunked(c, small))
        ok = bool(torch.equal(O, O2))
        print('chunk %3d planes (%5.0f MB) scratch_%s: %.4f ms  x%.2f  same=%s' % (
            c, c * L.sx * 8 / 1e6, 'small' if small else 'full ', t, t_full / t, ok), flush=True)
