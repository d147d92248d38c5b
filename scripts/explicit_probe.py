#!/usr/bin/env python3
"""explicit-stage timing at 512^3 (min of 20, HIP events)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mat = adi.Material(7800.0, 490.0, 54.0)
dx = 5e-4
alpha = mat.k / (mat.rho * mat.cp)
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
T = grid.layout.to_layout(torch.rand((n, n, n), dtype=torch.float64, device='cuda') * 900 + 20, torch.float64)
T = adi.DeviceField(T)
for _ in range(5):
    adi.adi_explicit_rhs(T, grid, mat, prm)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record(); adi.adi_explicit_rhs(T, grid, mat, prm); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
print('explicit min %.4f ms  mean %.4f ms  -> %.0f GB/s of 17 B/cell' % (min(ts), np.mean(ts), 17 * n ** 3 / min(ts) / 1e6))
