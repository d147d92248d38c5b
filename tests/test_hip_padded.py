"""Padded extents: a ragged (nx, ny, nz) grid is allocated as the physical box adi_recommended_dims() picks, the extra
cells are off-mask, the kernels run on the physical box and the caller sees the logical one (Layout, adi3d_hip_coeff.py).
The fuzz and parity suites cover these layouts implicitly (any long axis that is not a friendly length); this file pins
the properties of the mechanism itself."""
import ctypes

import numpy as np
import pytest

from helpers import rel_linf, run_cart_case

STEEL = dict(rho=7800.0, cp=490.0, k=54.0)
ALPHA = STEEL['k'] / (STEEL['rho'] * STEEL['cp'])


def _dims(nx, ny, nz):
    from adi_thermal_fields_amd import _lib
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert _lib.lib.adi_recommended_dims(nx, ny, nz, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    return a.value, b.value, c.value


def test_recommended_dims_properties():
    """host-only arithmetic (no GPU): never smaller than the logical box, bounded growth, friendly boxes untouched, small
    boxes untouched, nz a multiple of 16 wherever a strided FAST kernel could run"""
    for s in [(512, 512, 512), (256, 256, 256), (128, 256, 512), (64, 64, 64), (320, 384, 448), (640, 512, 512), (6, 5, 7),
              (48, 48, 48), (40, 33, 16), (1, 4, 6)]:
        assert _dims(*s) == s, s
    rng = np.random.default_rng(5)
    for _ in range(300):
        s = tuple(int(v) for v in rng.integers(1, 700, 3))
        p = _dims(*s)
        assert all(q >= n and q <= n + n // 8 + 16 for n, q in zip(s, p)), (s, p)
        if s[2] >= 64 and (s[0] >= 64 or s[1] >= 64):
            assert p[2] % 16 == 0, (s, p)
        assert p[2] == s[2] or p[2] % 16 == 0, (s, p)
        for n, q in zip(s[:2], p[:2]):
            assert q == n or (n >= 64 and q % 8 == 0), (s, p)
    assert _dims(250, 250, 250) == (256, 256, 256) and _dims(255, 256, 257)[0] == 256
    assert _dims(300, 300, 300) == (320, 320, 320)                       # exact fits (10 / 20 rows x 32 / 16 segments) over 304
    from adi_thermal_fields_amd import _lib
    assert _lib.lib.adi_recommended_dims(0, 4, 4, None, None, None) != 0


@pytest.mark.gpu
def test_layout_round_trip_and_storage():
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(11)
    for shape in [(130, 66, 50), (70, 5, 250), (65, 64, 64), (257, 3, 48)]:
        L = hip.Layout(*shape)
        px, py, pz, sx = L.pd
        assert L.padded and (px, py, pz) == _dims(*shape) and sx >= py * pz and L.shape == shape
        a = rng.uniform(-5.0, 5.0, shape)
        t = L.to_layout(a, torch.float64)
        assert L.is_native(t) and tuple(t.shape) == shape and t.untyped_storage().nbytes() == px * sx * 8
        assert np.array_equal(L.to_host(t), a)
        flat = t.as_strided((px * sx,), (1,))
        assert float(flat.sum()) == pytest.approx(float(a.sum()), rel=1e-12)       # everything outside the logical box is zero
        m = rng.random(shape) < 0.5
        dm = L.to_layout(m, torch.uint8)
        assert np.array_equal(L.to_host(dm).astype(bool), m) and int(dm.as_strided((px * sx,), (1,)).sum()) == int(m.sum())
        f = hip.to_device(a)
        assert L.is_native(f.t) and np.array_equal(f.get(), a) and np.array_equal(np.asarray(f), a)
        g = f.copy()
        assert L.is_native(g.t) and g.t.data_ptr() != f.t.data_ptr() and np.array_equal(g.get(), a)
        f[3, 2, 1] = 7.5
        assert f[3, 2, 1] == 7.5 and g[3, 2, 1] == a[3, 2, 1]
        assert f.max() == max(7.5, a.max()) and f.shape == shape
        # a dense tensor of the logical shape is not native: it is copied into the layout, not reinterpreted
        dense = torch.from_numpy(a).to('cuda')
        assert not L.is_native(dense) and np.array_equal(L.to_host(L.to_layout(dense, torch.float64)), a)
    assert not hip.Layout(64, 64, 64).padded and not hip.Layout(130, 66, 50, sx=66 * 50).padded


@pytest.mark.gpu
@pytest.mark.parametrize('shape,kind', [((130, 66, 50), 'solid'), ((130, 66, 50), 'holes'), ((70, 90, 250), 'solid'),
                                        ((257, 20, 33), 'ellipsoid'), ((100, 100, 100), 'holes')])
def test_ragged_grid_vs_oracle(shape, kind):
    """ragged boxes (every axis on padded extents) against the oracle on the logical box; the padded cells stay zero and
    off-mask cells inside the box stay untouched; device-resident stepping and the graph loop give the same bits"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape))
    if kind == 'solid':
        mask = np.ones(shape, bool)
    elif kind == 'holes':
        mask = rng.random(shape) > 0.15
    else:
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
    dx = 1e-3
    c = dict(shape=shape, dx=dx, mat=dict(STEEL), mask=mask, T0=rng.uniform(20.0, 1500.0, shape), dir_mask=None, dir_value=None,
             neumann={'z+': 3e4}, robin_h={'x-': 300.0, 'x+': 50.0, 'y-': 0.0, 'y+': 700.0, 'z-': 100.0}, Tinf=25.0, theta=0.5,
             dt=150.0 * dx * dx / ALPHA, nsteps=3, births=None)
    got = run_cart_case(hip, c)
    want = run_cart_case(orc, c)
    for key in ('T_step1', 'T_final'):
        assert rel_linf(got[key], want[key]) <= 1e-10, (shape, kind, key, rel_linf(got[key], want[key]))
    assert np.array_equal(got['T_final'][~mask], c['T0'][~mask])
    grid = hip.Grid3D(*shape, dx, mask)
    assert grid.layout.padded and not grid.all_solid and grid.shape == shape and grid.mask.shape == shape
    packs = hip.precompute_coeff_packs_unified(grid, hip.Material(**STEEL), neumann=c['neumann'], robin_h=c['robin_h'])
    assert packs[2].qflux.shape == shape and packs[0].coeff.shape == shape
    st = hip.StagedStepper(grid, hip.Material(**STEEL), hip.Params(c['dt'], 0.5), packs, 25.0)
    T = hip.to_device(c['T0'])
    for _ in range(3):
        T = st.step(T)
    assert np.array_equal(T.get(), got['T_final'])
    px, py, pz, sx = grid.layout.pd
    phys = T.t.as_strided((px, py, pz), (sx, pz, 1)).cpu().numpy()
    inside = np.zeros((px, py, pz), bool); inside[:shape[0], :shape[1], :shape[2]] = True
    assert np.all(phys[~inside] == 0.0)                                  # identity rows: the padding is never written with anything else
    assert np.array_equal(st.run(hip.to_device(c['T0']), 3, graph=True).get(), got['T_final'])


@pytest.mark.gpu
def test_ragged_grid_exposure_counts_and_exposed_mask():
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(3)
    shape = (70, 66, 37)
    mask = rng.random(shape) > 0.3
    want = np.zeros(shape[2], np.int64)
    for f in ('x-', 'x+', 'y-', 'y+'):
        want += orc.exposed_mask(mask, f).sum(axis=(0, 1))
    assert np.array_equal(hip.exposed_faces_per_layer(mask), want)
    g = hip.Grid3D(*shape, 1e-3, mask)
    assert g.layout.padded and np.array_equal(hip.exposed_faces_per_layer(g), want)
    for f in ('x+', 'z+', 'y-'):
        assert np.array_equal(hip.exposed_mask(mask, f), orc.exposed_mask(mask, f))


@pytest.mark.gpu
@pytest.mark.parametrize('nz', [144, 208, 400, 432, 496, 528, 1008])
def test_contiguous_lines_with_a_segment_count_that_is_not_a_power_of_two(nz):
    """16 rows per lane, 9 ... 63 segments: the contiguous FAST kernel reads such lines coalesced too (coal_load_r: the lanes beyond
    the last segment are padding; taken for all-solid boxes, the others keep lane-owned chunks).  Two and four lines per wave
    and lines that span the whole wave; solid, voids, a curved solid, fluxes -- the axis-2 sweep alone against the oracle's, and
    a full step"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    shape = (6, 8, nz)
    assert hip.recommended_dims(*shape) == shape
    alpha = 54.0 / (7800.0 * 490.0)
    dx = 1e-3
    for kind in ('solid', 'holes', 'ellipsoid', 'solid_q'):
        rng = np.random.default_rng(nz + len(kind))
        mask = np.ones(shape, bool)
        if kind == 'holes':
            mask = rng.random(shape) > 0.02
        if kind == 'ellipsoid':
            g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
            mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
        neu = {'z+': 1e5, 'z-': -2e4, 'y+': 3e4} if kind == 'solid_q' else None
        c = dict(shape=shape, dx=dx, mat=dict(STEEL), mask=mask, T0=rng.uniform(20.0, 1500.0, shape), dir_mask=None,
                 dir_value=None, neumann=neu, robin_h={'z-': 350.0, 'z+': 20.0, 'y-': 100.0, 'x+': 500.0}, Tinf=20.0, theta=0.5,
                 dt=150.0 * dx * dx / alpha, nsteps=2, births=None)
        assert rel_linf(run_cart_case(hip, c)['T_final'], run_cart_case(orc, c)['T_final']) <= 1e-10, (nz, kind)
        for api, key in ((hip, 'h'), (orc, 'o')):
            grid = api.Grid3D(*shape, dx, mask)
            mat = api.Material(**STEEL); prm = api.Params(c['dt'], 0.5)
            packs = api.precompute_coeff_packs_unified(grid, mat, neumann=neu, robin_h=c['robin_h'])
            if api is hip:
                got = hip.adi_sweep_axis(2, c['T0'], grid, mat, prm, packs[2], Tinf=20.0)
            else:
                want = orc.sweep_axis(2, np.array(c['T0']), grid, mat, prm, packs[2], 20.0)
        assert rel_linf(got, want) <= 1e-12, (nz, kind, rel_linf(got, want))


@pytest.mark.gpu
def test_frame_output_of_a_padded_field(tmp_path):
    """a DeviceField on padded extents through the frame writers: the logical box, byte for byte"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import frame_io as fio
    rng = np.random.default_rng(9)
    shape = (70, 66, 37)
    T = rng.uniform(20.0, 900.0, shape)
    dev = hip.to_device(T)
    assert hip.Layout.of(dev.t).padded
    assert fio.pack_frame_f32be(dev) == T.reshape(-1, order='F').astype('>f4').tobytes()
    a, b = str(tmp_path / 'a.vtk'), str(tmp_path / 'b.vtk')
    fio.write_vtk_structured_points(a, T, 5e-4, field_name='Temp')
    fio.write_vtk_structured_points(b, dev, 5e-4, field_name='Temp')
    assert open(a, 'rb').read() == open(b, 'rb').read()
    p = str(tmp_path / 'f.npy')
    fio.write_npy(p, dev)
    assert np.array_equal(np.load(p), T)
