"""Do the stepper's two scratch fields want a small skew against 2 MiB alignment?  (offset_probe2.py: a 512-byte skew between a sweep's
input and output measured 2 - 3 % better on the axis-1 / axis-2 sweeps, within 1 - 2 sigma.)  The same StagedStepper, its scratch fields
either as allocated (each on a 2 MiB boundary) or as views skewed by 64 / 128 elements, ALTERNATED; per-stage event times.
    python scripts/skew_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

n = 512
dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
L = grid.layout
T = adi.DeviceField(torch.rand(L.shape, dtype=torch.float64, device='cuda') * 980 + 20)
for _ in range(5):
    T = st.step(T)
fields, work, wb = grid.scratch(2)
print('scratch bases mod 2 MiB: %#x %#x' % (fields[0].data_ptr() % (1 << 21), fields[1].data_ptr() % (1 << 21)))
numel = L.numel_padded
raw = [torch.empty(numel + 4096, dtype=torch.float64, device='cuda') for _ in range(2)]
variants = {'aligned': fields}
for name, (s0, s1) in {'skew 64/128': (64, 128), 'skew 64/64': (64, 64), 'skew 0/64': (0, 64), 'skew 256/512': (256, 512)}.items():
    variants[name] = [raw[0][s0:s0 + numel].as_strided(L.shape, L.strides), raw[1][s1:s1 + numel].as_strided(L.shape, L.strides)]
nst = len(st.stage_names)
res = {k: [] for k in variants}
for rnd in range(8):
    for name, fl in variants.items():
        grid._scratch = (fl, work, wb)
        for _ in range(2):
            T = st.step(T)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(10)]
        for s in range(10):
            T = st.step(T, events=ev[s])
        torch.cuda.synchronize()
        res[name].append(np.array([[e[i].elapsed_time(e[i + 1]) for i in range(nst)] for e in ev]).mean(axis=0))
print('%-14s %s   step' % ('scratch', '   '.join('%-22s' % s_ for s_ in st.stage_names)))
for name, v in res.items():
    a = np.array(v)
    print('%-14s %s   %.4f' % (name, '   '.join('%.4f +- %.4f       ' % (a[:, i].mean(), a[:, i].std()) for i in range(nst)), a.sum(axis=1).mean()))
