"""One rank of a slab decomposition (middle rank of 8, loopback communicator) on a slab whose sharded-axis lines are NOT uniform
(a cylinder along the sharded axis with an off-centre void, i.e. voids and surface crossings in the lines): step time with the
deferred form with per-line homogeneous solutions ('deferred_lines', ABI v17) and with the two-pass forms it replaces.
python scripts/deferred_lines_probe.py [NXLxNYxNZ ...] [cfl=200 ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab

dx = 1e-3
mat = hip.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
cfls = [float(a[4:]) for a in sys.argv[1:] if a.startswith('cfl=')] or [200.0]
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:] if not a.startswith('cfl=')] or [(512, 512, 512), (512, 256, 320)]
for shape in shapes:
    g = np.meshgrid((np.arange(shape[0]) + 0.5) / shape[0] - 0.5, (np.arange(shape[1]) + 0.5) / shape[1] - 0.5,
                    (np.arange(shape[2]) + 0.5) / shape[2] - 0.5, indexing='ij')
    mask = ((g[1] / 0.47) ** 2 + (g[2] / 0.45) ** 2 <= 1.0) & ~((g[0] / 0.2) ** 2 + ((g[1] - 0.1) / 0.15) ** 2 + (g[2] / 0.2) ** 2 <= 1.0)
    del g
    for cfl, on in [(c, o) for c in cfls for o in (True, False)]:
        st = dist_slab.SlabStepper(mask, dx, mat, hip.Params(cfl * dx * dx / alpha, 0.5), 20.0, robin_h=500.0,
                                   comm=dist_slab.LoopbackComm(8, 4))
        st._allow_deferred_lines = on
        T = torch.full(shape, 300.0, dtype=torch.float64, device='cuda')
        for _ in range(30):
            T = st.step(T, prefetch_halo=True)
        torch.cuda.synchronize()
        names = st.stage_names
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(20)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(20):
            T = st.step(T, events=ev[k], prefetch_halo=True)
        e1.record(); torch.cuda.synchronize()
        stg = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(len(names))] for k in range(20)]).mean(axis=0)
        print(shape, 'cfl %g' % cfl, 'in mask %.0f %%' % (100.0 * mask.mean()), 'deferred_lines allowed' if on else 'two-pass forms',
              '-> %s: %.3f ms per step (%s)' % (st.axis0_mode, e0.elapsed_time(e1) / 20, ' '.join('%.3f' % v for v in stg)),
              'K = %s' % (st._a0 or {}).get('K'), flush=True)
        del st, T
