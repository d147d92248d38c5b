"""Does the RELATIVE placement of a sweep's input and output field matter?  (Same box, same build, bench.py's axis-1 sweep took
0.413 ms in one process and 0.443 in the next: the allocator had handed out other addresses.)  One big buffer; the input field sits
at its start, the output field at `numel + d` elements for a list of d; the flags stay where they are.  Times the three stage
kernels of the lean 512^3 step for each d (median of 12).    python scripts/offset_probe.py [n=512]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 5e-4
mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp)
prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
grid = adi.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
L = grid.layout
numel = L.numel_padded
slack = 1 << 24
big = torch.empty(2 * numel + 2 * slack, dtype=torch.float64, device='cuda')
print('base address %#x (mod 2 MiB: %#x), field %d elements = %.1f MiB, plane stride %d' % (big.data_ptr(), big.data_ptr() % (1 << 21), numel, numel * 8 / 2 ** 20, L.sx))
tin = big[:numel].as_strided(L.shape, L.strides)
tin.copy_(torch.rand(L.shape, dtype=torch.float64, device='cuda') * 980 + 20)
for _ in range(4):
    T = st.step(adi.DeviceField(tin))      # learn the no-fallback promises
offs = [0, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, (1 << 20) + 4096, 3 << 19, 1 << 21, 1 << 22, 1 << 23]
print('%-10s %s' % ('d (elem)', 'fused ex+axis0   axis1    axis2   [ms, median of 12]'))
for d in offs:
    tout = big[numel + d: 2 * numel + d].as_strided(L.shape, L.strides)
    res = []
    for which in ('fused', 1, 2):
        tt = []
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for it in range(15):
            e0.record()
            if which == 'fused':
                adi._explicit_sweep0_into(tin, tout, grid, mat, prm, packs[0], 20.0)
            else:
                st.sweep_into(which, tin, tout)
            e1.record(); e1.synchronize()
            if it >= 3:
                tt.append(e0.elapsed_time(e1))
        res.append(float(np.median(tt)))
    print('%-10d %.4f           %.4f   %.4f     (out - in = %#x bytes)' % (d, res[0], res[1], res[2], (numel + d) * 8), flush=True)
