#!/usr/bin/env python3
"""Per-stage timing probe (HIP events via torch on the launch stream): explicit stage, each sweep axis,
general-pack and lean variants, full step.  Prints GB/s against the algorithmic byte counts."""
import sys, time, json
import numpy as np
import torch
sys.path.insert(0, '.')
import adi_thermal_fields_amd.adi3d_hip_coeff as adi
from adi_thermal_fields_amd import _lib


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    shape = (n, n, n)
    if len(sys.argv) > 3:
        shape = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    N = shape[0] * shape[1] * shape[2]
    dev = torch.device('cuda')
    mask = np.ones(shape, bool)
    grid = adi.Grid3D(shape[0], shape[1], shape[2], 5e-4, mask)
    mat = adi.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    prm = adi.Params(200.0 * grid.dx ** 2 / alpha, 0.5)
    t0 = time.time()
    packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    torch.cuda.synchronize()
    print('pack build s', time.time() - t0)
    g = torch.Generator(device=dev); g.manual_seed(0)
    T = adi.DeviceField(grid.layout.to_layout(torch.rand(shape, dtype=torch.float64, device=dev, generator=g) * 980.0 + 20.0, torch.float64))
    res = {}
    ms, mn = timeit(lambda: adi.adi_explicit_rhs(T, grid, mat, prm))
    res['explicit'] = (ms, mn, 17 * N / mn / 1e6)
    for ax in range(3):
        for v, nm, dn in ((_lib.SWEEP_GENERAL, 'general', True), (None, 'lean_dense', True), (None, 'lean', False)):
            ms, mn = timeit(lambda: adi.adi_sweep_axis(ax, T, grid, mat, prm, packs[ax], Tinf=20.0, variant=v, dense=dn))
            bpc = 42 if v == 0 else (_lib.SWEEP_BYTES_PER_CELL[packs[ax].variant] if dn else packs[ax].bytes_per_cell)
            res['sweep%d_%s' % (ax, nm)] = (ms, mn, bpc * N / mn / 1e6)
    ms, mn = timeit(lambda: adi.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0))
    res['step_lean'] = (ms, mn, (17 + sum(p.bytes_per_cell for p in packs)) * N / mn / 1e6)
    for k, (ms, mn, gbs) in res.items():
        print('%-16s median %8.3f ms  min %8.3f ms  %8.1f GB/s (min)' % (k, ms, mn, gbs))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
