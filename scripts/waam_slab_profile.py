"""Where the host time of the per-rank layer-birth loop goes (rank 1 of 4 of the 256 x 256 x 320 head takes
0.22 s against 0.14 s for the whole head on one GPU -- a slab step is a dozen launches of a few microseconds each, so the loop is
bound by the host).  Every method of the engine, the communicator and the stepper is timed on the host over one run of
waam.run_layer_birth_slab (rank 1 of 4, loopback communicator); prints the loop time and the calls sorted by total host time.
python scripts/waam_slab_profile.py"""
import os, sys, time, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab, waam

STEEL = (7800.0, 490.0, 54.0)
shape = (256, 256, 320)
mask = waam.synthetic_head_mask(*shape)
layers = waam.plan_layers(mask, 2)
dx = 1e-3
times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
outs = [times[-1]]
world, rank = 4, 1
i0, i1 = rank * 64, (rank + 1) * 64
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
depth = [0]


def wrap_cls(cls, names=None):
    for nm in (names or [n for n in dir(cls) if not n.startswith('__') and callable(getattr(cls, n))]):
        f = getattr(cls, nm)
        if isinstance(f, (staticmethod, classmethod)) or not callable(f):
            continue

        def make(f, label):
            def g(*a, **k):
                t0 = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    tot[label] += time.perf_counter() - t0; cnt[label] += 1
            return g
        try:
            setattr(cls, nm, make(f, cls.__name__ + '.' + nm))
        except Exception:
            pass


for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Tl, n2 = waam.run_layer_birth_slab(dist_slab.LoopbackComm(world, rank), i0, i1, mask, dx, hip.Material(*STEEL), hip.Params,
                                       40.0, 20.0, 1000.0, 1.0, 2000.0, layers, times, outs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('unprofiled run %d: %d births, %d steps: %.3f s' % (rep, len(layers), n2, t1 - t0), flush=True)
wrap_cls(dist_slab.HipEngine, [n for n in dir(dist_slab.HipEngine) if not n.startswith('__') and n not in ('check',)])
wrap_cls(dist_slab.LoopbackComm, ['exchange_planes', 'all_gather'])
wrap_cls(dist_slab.SlabStepper, ['step', '_step', '_plan_axis0', '_plan_deferred_lines', '_load_state', '_streams', 'set_mask_device', 'set_mask',
                                 '_fused_supported', '_axis0_pipeline', '_axis0_finish', 'self_check'])
torch.cuda.synchronize(); t0 = time.perf_counter()
Tl, n2 = waam.run_layer_birth_slab(dist_slab.LoopbackComm(world, rank), i0, i1, mask, dx, hip.Material(*STEEL), hip.Params,
                                   40.0, 20.0, 1000.0, 1.0, 2000.0, layers, times, outs)
torch.cuda.synchronize(); t1 = time.perf_counter()
print('profiled run: %d births, %d steps: %.3f s (host timers on)' % (len(layers), n2, t1 - t0))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:28]:
    print('  %-44s %8.1f ms  %6d calls  %7.1f us/call' % (k, v * 1e3, cnt[k], v / cnt[k] * 1e6))
