"""One-off soak of the slab decomposition: seeded random grids, masks (all-solid -> the deferred forms; holes / curved ->
the two-pass forms), boundary sets, time steps, world sizes 2..5 and even / uneven slab thicknesses, several in-process
ranks on one GPU (tests/test_dist_slab_gpu.py::_run_slabs) against the single-domain HIP step.
    python scripts/slab_soak.py [N=200] [first_seed=0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_dist_slab_gpu as D
from helpers import rel_linf, run_cart_case
import adi_thermal_fields_amd.adi3d_hip_coeff as hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
alpha = 54.0 / (7800.0 * 490.0)
worst, bad, modes_seen = 0.0, [], {}
t0 = time.time()
for seed in range(S0, S0 + N):
    rng = np.random.default_rng(77000 + seed)
    world = int(rng.integers(2, 6))
    sizes = [int(rng.choice([8, 16, 24, 32, 62, 64, 66, 128])) for _ in range(world)]
    if rng.random() < 0.3:
        sizes[int(rng.integers(0, world))] += int(rng.choice([1, 3]))          # an odd slab: generic condensation kernels
    nx = sum(sizes)
    ny = int(rng.choice([4, 8, 16, 24, 70, 100])); nz = int(rng.choice([16, 32, 40, 64, 50, 90]))     # (70, 100, 40, 50, 90: padded planes)
    shape = (nx, ny, nz)
    kind = str(rng.choice(['solid', 'solid', 'holes', 'ellipsoid']))
    if kind == 'solid':
        mask = np.ones(shape, bool)
    elif kind == 'holes':
        mask = rng.random(shape) > 0.05
    else:
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = (g[0] / 0.48) ** 2 + (g[1] / 0.47) ** 2 + (g[2] / 0.49) ** 2 <= 1.0
    cfl = float(rng.choice([0.05, 1.0, 3.0, 50.0, 300.0, 2000.0]))
    dx = 1e-3
    bc = str(rng.choice(['scalar', 'array', 'neumann']))
    robin_h = 300.0 if bc != 'array' else rng.uniform(50.0, 700.0, shape)
    neumann = {'x-': 2e5, 'x+': 1e5, 'y+': 5e4} if bc == 'neumann' else None
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1200.0, shape),
             dir_mask=None, dir_value=None, neumann=neumann, robin_h=robin_h, Tinf=20.0, theta=float(rng.choice([0.5, 1.0])),
             dt=cfl * dx * dx / alpha, nsteps=3, births=None)
    opts = dict(prefetch=bool(rng.random() < 0.7), allow_fused=bool(rng.random() < 0.8), allow_dots=bool(rng.random() < 0.7),
                allow_deferred=bool(rng.random() < 0.85), allow_deferred_exact=bool(rng.random() < 0.8),
                allow_deferred_lines=bool(rng.random() < 0.6))
    modes = set()
    opts['cost_rule'] = bool(rng.random() < 0.5)
    got = D._run_slabs(c, world, sizes, 3, opts, modes)
    want = run_cart_case(hip, c)['T_final']
    err = rel_linf(got, want)
    worst = max(worst, err)
    key = (kind, tuple(sorted(modes)))
    modes_seen[key] = modes_seen.get(key, 0) + 1
    if not err <= 1e-12 or len(modes) != 1:
        bad.append((seed, shape, sizes, kind, cfl, bc, opts, sorted(modes), err))
    if (seed - S0) % 25 == 24:
        print('%d / %d, worst so far %.3e, %.0f s' % (seed - S0 + 1, N, worst, time.time() - t0), flush=True)
print('slab soak: %d cases, worst rel L-inf vs one domain %.3e, failures %s' % (N, worst, bad))
print('  (mask kind, interface form) counts:', dict(sorted(modes_seen.items())))
sys.exit(1 if bad else 0)
