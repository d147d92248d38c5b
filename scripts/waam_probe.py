"""wall time of the layer-birth loop (BASELINE configs[4] geometry: synthetic head 256 x 256 x 320, 160 births of 2 planes,
cfl 2000, theta = 1, pack rebuild per birth) on one GPU, device loop vs host-side mask handling.
    python scripts/waam_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import waam

shape = (256, 256, 320)
mask = waam.synthetic_head_mask(*shape)
layers = waam.plan_layers(mask, 2)
dx = 1e-3
times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
outs = [0.0, times[-1]]
STEEL = (7800.0, 490.0, 54.0)
res = {}
for name, kw in (('device loop', {}), ('host masks', dict(device_loop=False))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        T, n = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 1.0, 2000.0, layers, times, outs, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = T
    print('%-12s %d births, %d steps: %.2f s (%.1f ms per birth+segment)' % (name, len(layers), n, dt, 1e3 * dt / len(layers)), flush=True)
print('identical:', bool(np.array_equal(res['device loop'], res['host masks'])))
