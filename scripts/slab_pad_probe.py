"""One rank of a slab decomposition (middle rank of 4, loopback communicator) on ragged planes: step time with the planes
padded (SlabStepper default, HipEngine.plane_dims) and with the caller's planes as they are.
python scripts/slab_pad_probe.py [NXLxNYxNZ ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab

dx = 1e-3
mat = hip.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(128, 250, 250), (128, 300, 300), (64, 237, 402)]
for shape in shapes:
    g = np.meshgrid(np.zeros(shape[0]), (np.arange(shape[1]) + 0.5) / shape[1] - 0.5, (np.arange(shape[2]) + 0.5) / shape[2] - 0.5, indexing='ij')
    for kind, mask in (('box', np.ones(shape, bool)), ('cylinder', (g[1] / 0.47) ** 2 + (g[2] / 0.45) ** 2 <= 1.0)):
        out = []
        for pad in (True, False):
            E = dist_slab.HipEngine()
            if not pad:
                E.plane_dims = lambda ny, nz: (ny, nz)
            st = dist_slab.SlabStepper(mask, dx, mat, hip.Params(200.0 * dx * dx / alpha, 0.5), 20.0, robin_h=500.0,
                                       comm=dist_slab.LoopbackComm(4, 1), engine=E)
            T = torch.full(shape, 300.0, dtype=torch.float64, device='cuda')
            for _ in range(40):                    # (plans, promises and first allocations settle in the first tens of steps)
                T = st.step(T, prefetch_halo=True)
            torch.cuda.synchronize()
            # per-step times between events: the MEDIAN is reported -- one cyclic garbage collection of the Python heap (35 - 60 ms,
            # scripts/slab_outlier_probe.py) inside a 30-step loop once put 2.400 ms where the steps take 1.2
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
            evs[0].record()
            for k in range(30):
                T = st.step(T, prefetch_halo=True)
                evs[k + 1].record()
            torch.cuda.synchronize()
            per = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(30)])
            names = st.stage_names
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(10)]
            for k in range(10):
                T = st.step(T, events=ev[k], prefetch_halo=True)
            torch.cuda.synchronize()
            stg = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(len(names))] for k in range(10)]).mean(axis=0)
            p = st._a0 or {}
            out.append('%s planes %s: median %.3f ms, mean %.3f, max %.3f (%s%s%s; %s)' % ('padded' if pad else 'caller\'s', (st.ny, st.nz), np.median(per), per.mean(), per.max(),
                                                              st.axis0_mode, ', dots' if p.get('dots') else '', ', fused' if p.get('fused') else '',
                                                              ' '.join('%.3f' % v for v in stg)))
            del st, T
        print(shape, kind, '; '.join(out), flush=True)
