"""Where do the one-off ~36 ms steps of the `exact` slab form come from (padded-plane cylinder 2.400 ms per
step over a 30-step loop against a stage sum of 1.18 ms)?  One middle rank of 4 (loopback communicator), 256 x 500 x 500
cylinder on padded planes, 40 warm-up + 90 steps; every step is bracketed by host timestamps (with a device synchronisation)
and every call into the engine / the communicator is timed on the host; allocator counters are sampled per step.  Prints the
median step, every step above 3x the median with its per-call host breakdown, and the allocator deltas.
python scripts/slab_outlier_probe.py [NXLxNYxNZ] [box|cylinder]"""
import os, sys, time, collections
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
E = dist_slab.HipEngine()
comm = dist_slab.LoopbackComm(4, 1)
calls = collections.defaultdict(float)


def wrap(obj, name):
    f = getattr(obj, name)

    def g_(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            calls[type(obj).__name__ + '.' + name] += (time.perf_counter() - t0) * 1e3
    setattr(obj, name, g_)


for nm in dir(E):
    if not nm.startswith('__') and callable(getattr(E, nm)) and nm not in ('check',):
        wrap(E, nm)
for nm in ('exchange_planes', 'all_gather'):
    wrap(comm, nm)
st = dist_slab.SlabStepper(mask, dx, mat, hip.Params(200.0 * dx * dx / alpha, 0.5), 20.0, robin_h=500.0, comm=comm, engine=E)
for nm in ('_plan_axis0', '_load_state', '_streams'):
    wrap(st, nm)
T = torch.full(shape, 300.0, dtype=torch.float64, device='cuda')
rows = []
keys = ('num_alloc_retries', 'num_device_alloc', 'num_device_free', 'reserved_bytes.all.current', 'allocated_bytes.all.peak')
prev = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
for s in range(130):
    calls.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = st.step(T, prefetch_halo=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms_ = torch.cuda.memory_stats()
    cur = {k: ms_.get(k, 0) for k in keys}
    rows.append(dict(step=s, host_ms=(t1 - t0) * 1e3, total_ms=(t2 - t0) * 1e3, calls=dict(calls),
                     mem={k: cur[k] - prev[k] for k in keys}, plan_steps=st._plan_steps, mode=st.axis0_mode))
    prev = cur
tot = np.array([r['total_ms'] for r in rows])
late = tot[40:]
print('%s %s on planes %s: form %s; steps 40..129: median %.3f ms, mean %.3f ms, max %.3f ms' %
      (shape, kind, (st.ny, st.nz), st.axis0_mode, np.median(late), late.mean(), late.max()), flush=True)
med = float(np.median(late))
for r in rows:
    if r['total_ms'] > 3 * med and r['step'] >= 3:
        top = sorted(r['calls'].items(), key=lambda kv: -kv[1])[:5]
        print('  step %3d: total %.2f ms (host part %.2f); plan_steps %s; alloc deltas %s; slowest host calls: %s' %
              (r['step'], r['total_ms'], r['host_ms'], r['plan_steps'], {k: v for k, v in r['mem'].items() if v},
               ', '.join('%s %.2f' % kv for kv in top)), flush=True)
print('  first three steps: ' + ', '.join('%.1f ms' % r['total_ms'] for r in rows[:3]))
