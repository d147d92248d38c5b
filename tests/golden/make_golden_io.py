#!/usr/bin/env python3
"""Generate tests/golden/io_morph.npz and io_vtk_*.txt by importing the REFERENCE (build container only; the
reference never travels): voxel morphology of waam_from_stl_v7_mm.py:73-188 and the two ASCII VTK writers
(vtk_writer.py, waam_from_stl_v7_mm.py:191-216) on small seeded inputs.
    python tests/golden/make_golden_io.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.argv = [sys.argv[0]]
import vtk_writer                      # noqa: E402
import waam_from_stl_v7_mm as w        # noqa: E402


def cases():
    rng = np.random.default_rng(42)
    out = {}
    out['random30'] = rng.random((9, 11, 13)) < 0.3
    out['random70'] = rng.random((12, 7, 10)) < 0.7
    s = np.zeros((12, 12, 12), bool); s[2:10, 2:10, 2:10] = True; s[3:9, 3:9, 3:9] = False
    out['hollow_cube'] = s
    X, Y, Z = np.meshgrid(*(np.arange(n) - (n - 1) / 2 for n in (17, 15, 16)), indexing='ij')
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    sph = (r < 6.5) & (r > 4.5)
    sph[8, 7, :] = False                  # a hole through the shell along z
    out['leaky_sphere'] = sph
    out['empty'] = np.zeros((5, 6, 7), bool)
    out['thin'] = rng.random((1, 6, 9)) < 0.5
    return out


def main():
    g = {}
    for name, m in cases().items():
        g[name + '__in'] = m
        g[name + '__dilate6'] = w.dilate6(m)
        g[name + '__erode6'] = w.erode6(m)
        g[name + '__closing6_2'] = w.closing6(m, iters=2)
        g[name + '__flood_outside'] = w.flood_fill_outside(m)
        for mode in ('off', 'flood', 'close_flood', 'auto'):
            g[name + '__solidify_' + mode] = w.solidify_mask(m, mode=mode, close_iters=2, verbose=False)
    np.savez_compressed(os.path.join(HERE, 'io_morph.npz'), **g)
    rng = np.random.default_rng(7)
    T = rng.uniform(-5.0, 1500.0, (4, 3, 5)); T[0, 0, 0] = 0.0; T[1, 2, 3] = 1e-7; T[3, 1, 4] = 123456789.0
    M = rng.random((4, 3, 5)) < 0.5
    np.savez_compressed(os.path.join(HERE, 'io_vtk_inputs.npz'), T=T, M=M)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'a.vtk')
        vtk_writer.write_vtk_structured_points(p, T, 5e-4, origin=(0.001, -0.002, 0.0), field_name="Temp", mask=M)
        open(os.path.join(HERE, 'io_vtk_points.txt'), 'w', encoding='utf-8').write(open(p, encoding='utf-8').read())
        vtk_writer.write_vtk_structured_points(p, T, 5e-4)
        open(os.path.join(HERE, 'io_vtk_points_nomask.txt'), 'w', encoding='utf-8').write(open(p, encoding='utf-8').read())
        w.write_vtk_structured_points(p, T, 0.5, origin_mm=(1.0, -2.0, 0.25), field_name="Temperature", mask=M)
        open(os.path.join(HERE, 'io_vtk_waam.txt'), 'w', encoding='utf-8').write(open(p, encoding='utf-8').read())
    print('wrote', len(g), 'morphology arrays and 3 VTK texts')


if __name__ == '__main__':
    main()
