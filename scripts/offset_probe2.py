"""Follow-up to offset_probe.py: a few candidate placements of the output field relative to the input, ALTERNATED many times
(the first probe made one pass; differences of 2 - 5 % need alternation to be believed).  d = elements between the end of the input
field and the start of the output field; the fields are 1025 MiB, so d = 131072 (1 MiB) puts the output at a multiple of 2 MiB from
the input -- what two separate allocations of the caching allocator give.    python scripts/offset_probe2.py"""
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
numel = L.numel_padded
big = torch.empty(2 * numel + (1 << 22), dtype=torch.float64, device='cuda')
tin = big[:numel].as_strided(L.shape, L.strides)
tin.copy_(torch.rand(L.shape, dtype=torch.float64, device='cuda') * 980 + 20)
for _ in range(4):
    st.step(adi.DeviceField(tin))
cands = [131072, 131072 + 16, 131072 + 64, 131072 + 1024, 16, 64, 1024, 262144 + 32768]
res = {d: {'fused': [], 1: [], 2: []} for d in cands}
for rnd in range(10):
    for d in cands:
        tout = big[numel + d: 2 * numel + d].as_strided(L.shape, L.strides)
        for which in ('fused', 1, 2):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            tt = []
            for it in range(6):
                e0.record()
                if which == 'fused':
                    adi._explicit_sweep0_into(tin, tout, grid, mat, prm, packs[0], 20.0)
                else:
                    st.sweep_into(which, tin, tout)
                e1.record(); e1.synchronize()
                if it >= 1:
                    tt.append(e0.elapsed_time(e1))
            res[d][which].append(float(np.mean(tt)))
print('%-10s %-18s %-18s %-18s  (mean +- std over 10 alternated rounds, ms)' % ('d', 'fused', 'axis1', 'axis2'))
for d in cands:
    print('%-10d %s' % (d, '   '.join('%.4f +- %.4f' % (np.mean(res[d][w]), np.std(res[d][w])) for w in ('fused', 1, 2))), '  (out - in) mod 2 MiB = %#x' % (((numel + d) * 8) % (1 << 21)))
