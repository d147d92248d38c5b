"""Plane-pitch sweep on elongated boxes (the fused kernel at 139-149 Gcell/s on (320..512) x 256 x 256
against 207-209 at 256^3 and 205 at 512^3).  For each shape the plane stride is forced to ny*nz + pad (elements) and the three
stage kernels are timed.   python scripts/pitch_probe.py [NXxNYxNZ ...] [--pads 0,64,256,...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as adi

mat = adi.Material(7800.0, 490.0, 54.0); alpha = mat.k / (mat.rho * mat.cp); dx = 5e-4
pads = [0, 32, 64, 128, 256, 512, 768, 1024, 2048, 4096, 8192 + 256, 16384 + 256]
shapes = []
GEN = '--gen' in sys.argv            # also time the dense general-pack (42 B/cell) sweeps along axes 0 and 1
for a_ in sys.argv[1:]:
    if a_ == '--gen':
        continue
    if a_.startswith('--pads='):
        pads = [int(v) for v in a_.split('=')[1].split(',')]
    else:
        shapes.append(tuple(int(v) for v in a_.split('x')))
shapes = shapes or [(512, 256, 256), (320, 256, 256), (256, 256, 256)]
_orig_init = adi.Layout.__init__
PAD = [None]


def _init(self, nx, ny, nz, sx=None, phys=None):
    _orig_init(self, nx, ny, nz, sx, phys)
    if sx is None and PAD[0] is not None:
        self.sx = self.py * self.pz + PAD[0]


adi.Layout.__init__ = _init
print('%-16s %-7s %-6s %s' % ('shape', 'pad', 'fused', 'ms total, Gcell/s per stage'))
for shape in shapes:
    for pad in [None] + pads:
        PAD[0] = pad
        grid = adi.Grid3D(*shape, dx, np.ones(shape, bool))
        prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
        packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
        st = adi.StagedStepper(grid, mat, prm, packs, 20.0)
        T = adi.to_device(np.full(shape, 300.0))
        nst = len(st.stage_names)
        for _ in range(4):
            T = st.step(T)
        K = 10
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
        for s in range(K):
            T = st.step(T, events=ev[s])
        torch.cuda.synchronize()
        ms = np.median(np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(K)]), axis=0)
        N = shape[0] * shape[1] * shape[2]
        gen = ''
        if GEN:
            from adi_thermal_fields_amd import _lib
            L = grid.layout
            tin = L.to_layout(T, torch.float64); out = L.empty()
            for ax in (0, 1):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                tt = []
                for it in range(9):
                    e0.record(); st.sweep_into(ax, tin, out, variant=_lib.SWEEP_GENERAL, dense=True); e1.record(); e1.synchronize()
                    if it >= 3:
                        tt.append(e0.elapsed_time(e1))
                gen += '  gen42 axis%d %.4f ms' % (ax, float(np.median(tt)))
            del tin, out
        print('%-16s %-7s %-6s %6.3f ms  %s' % ('x'.join(map(str, shape)), 'dflt:%d' % (grid.layout.sx - grid.layout.py * grid.layout.pz) if pad is None else pad,
                                               st.fused, ms.sum(), '  '.join('%s %.0f' % (nm.replace('sweep_', '').replace('explicit', 'ex'), N / m / 1e6)
                                                                             for nm, m in zip(st.stage_names, ms)) + gen), flush=True)
        del grid, packs, st, T
        torch.cuda.empty_cache()
