"""per-step time (host clock around a synchronised step) of the first steps of a slab stepper: where do one-off stalls sit?
python scripts/slab_stall_probe.py [NXLxNYxNZ] [box|cylinder]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '256x500x500').split('x'))
kind = sys.argv[2] if len(sys.argv) > 2 else 'cylinder'
dx = 1e-3
mat = hip.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
g = np.meshgrid(np.zeros(shape[0]), (np.arange(shape[1]) + 0.5) / shape[1] - 0.5, (np.arange(shape[2]) + 0.5) / shape[2] - 0.5, indexing='ij')
mask = np.ones(shape, bool) if kind == 'box' else (g[1] / 0.47) ** 2 + (g[2] / 0.45) ** 2 <= 1.0
st = dist_slab.SlabStepper(mask, dx, mat, hip.Params(200.0 * dx * dx / alpha, 0.5), 20.0, robin_h=500.0, comm=dist_slab.LoopbackComm(4, 1))
T = torch.full(shape, 300.0, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
ts = []
for s in range(60):
    t0 = time.perf_counter()
    T = st.step(T, prefetch_halo=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(shape, kind, st.axis0_mode, 'reserved MiB', torch.cuda.memory_reserved() >> 20)
print(' '.join('%.2f' % t for t in ts))
