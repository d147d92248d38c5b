"""GPU: device voxel morphology and frame output against the reference's golden outputs, the
oracle and scipy; bit-exact (byte / boolean work)."""
import os

import numpy as np
import pytest

from oracle import morph_oracle as mo

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['random30', 'random70', 'hollow_cube', 'leaky_sphere', 'empty', 'thin']


@pytest.mark.parametrize('name', CASES)
def test_morphology_vs_reference_golden(name):
    from adi_thermal_fields_amd import voxel_morph as vm
    g = np.load(os.path.join(G, 'io_morph.npz'))
    m = g[name + '__in']
    assert np.array_equal(vm.dilate6(m), g[name + '__dilate6'])
    assert np.array_equal(vm.erode6(m), g[name + '__erode6'])
    assert np.array_equal(vm.closing6(m, iters=2), g[name + '__closing6_2'])
    assert np.array_equal(vm.flood_fill_outside(m, reference_defect=True), g[name + '__flood_outside'])
    for mode in ('off', 'flood', 'close_flood', 'auto'):
        assert np.array_equal(vm.solidify_mask(m, mode, 2, reference_defect=True), g[name + '__solidify_' + mode]), mode
        assert np.array_equal(vm.solidify_mask(m, mode, 2), mo.solidify_mask(m, mode, 2)), mode     # intended semantics
    assert np.array_equal(vm.flood_fill_outside(m), mo.flood_fill_outside(m))


def test_flood_fill_large_vs_scipy():
    """the synthetic head (104 x 104 x 128 box) as a 2-voxel shell with pin-holes that closing seals: device solidify ==
    scipy.ndimage.binary_fill_holes of the closed shell; ragged sizes; a maze-like mask (many rounds of line scans)"""
    from scipy import ndimage
    from adi_thermal_fields_amd import voxel_morph as vm, waam
    st = ndimage.generate_binary_structure(3, 1)
    head = np.pad(waam.synthetic_head_mask(96, 96, 120), 4)      # clear of the box faces (erode6 always clears those)
    shell = head & ~ndimage.binary_erosion(head, structure=st, iterations=2)
    rng = np.random.default_rng(3)
    shell &= rng.random(shell.shape) > 0.002                     # pin-holes: closing6 seals them
    closed = mo.closing6(shell, 2)
    want = ndimage.binary_fill_holes(closed, structure=st)
    got = vm.solidify_mask(shell, 'close_flood', 2)
    assert np.array_equal(got, want)
    assert got.sum() > 5 * shell.sum()                           # it did fill the inside
    assert np.array_equal(vm.solidify_mask(shell, 'auto', 2), want)          # shell test -> close_flood
    assert np.array_equal(vm.solidify_mask(head, 'auto', 2), head)           # solid -> unchanged
    for shape in ((37, 5, 61), (1, 1, 9), (64, 64, 64)):
        m = rng.random(shape) < 0.45                             # near the percolation threshold: tortuous air paths
        lab, _ = ndimage.label(np.pad(~m, 1, constant_values=True), structure=st)
        assert np.array_equal(vm.flood_fill_outside(m), (lab == lab[0, 0, 0])[1:-1, 1:-1, 1:-1]), shape
        assert np.array_equal(vm.dilate6(m), mo.dilate6(m)) and np.array_equal(vm.erode6(m), mo.erode6(m))


def test_vtk_writers(tmp_path):
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import frame_io as fio
    d = np.load(os.path.join(G, 'io_vtk_inputs.npz'))
    T, M = d['T'], d['M']
    rd = lambda n: open(os.path.join(G, n), encoding='utf-8').read()
    p = str(tmp_path / 'a.vtk')
    # ASCII mode: the reference's text, byte for byte (device-resident field in, too)
    fio.write_vtk_structured_points(p, T, 5e-4, origin=(0.001, -0.002, 0.0), field_name="Temp", mask=M, binary=False)
    assert open(p, encoding='utf-8').read() == rd('io_vtk_points.txt')
    fio.write_vtk_structured_points(p, hip.to_device(T), 5e-4, binary=False)
    assert open(p, encoding='utf-8').read() == rd('io_vtk_points_nomask.txt')
    fio.write_vtk_structured_points_mm(p, T, 0.5, origin_mm=(1.0, -2.0, 0.25), mask=M, binary=False)
    assert open(p, encoding='utf-8').read() == rd('io_vtk_waam.txt')
    # BINARY mode: same header lines (ASCII -> BINARY), payload = big-endian float32 in the same point order
    fio.write_vtk_structured_points(p, hip.to_device(T), 5e-4, origin=(0.001, -0.002, 0.0), field_name="Temp", mask=M)
    raw = open(p, 'rb').read()
    ref_head = rd('io_vtk_points.txt').split('SCALARS')[0].replace('ASCII', 'BINARY').encode()
    assert raw.startswith(ref_head + b"SCALARS Temp float 1\nLOOKUP_TABLE default\n")
    off = len(ref_head) + len(b"SCALARS Temp float 1\nLOOKUP_TABLE default\n")
    n = T.size
    assert raw[off:off + 4 * n] == T.reshape(-1, order='F').astype('>f4').tobytes()
    off2 = off + 4 * n + 1 + len(b"SCALARS mask float 1\nLOOKUP_TABLE default\n")
    assert raw[off2:off2 + 4 * n] == M.astype(np.float32).reshape(-1, order='F').astype('>f4').tobytes()
    assert len(raw) == off2 + 4 * n + 1
    # a padded-plane field at a size with ragged tiles, and .npy
    rng = np.random.default_rng(1)
    T2 = rng.uniform(-1e3, 1e3, (37, 64, 50))
    g2 = hip.Grid3D(37, 64, 50, 1e-3, np.ones((37, 64, 50), bool))
    dev = hip.DeviceField(g2.layout.to_layout(T2, __import__('torch').float64))
    assert fio.pack_frame_f32be(dev) == T2.reshape(-1, order='F').astype('>f4').tobytes()
    q = str(tmp_path / 'f.npy')
    fio.write_npy(q, dev)
    assert np.array_equal(np.load(q), T2)


def test_exposed_face_count_and_perimeter_ratio():
    """device exposed-face count (adi_count_exposed_faces) against the cell-by-cell count of
    quick_compare_layer_birth_robin_v3.py:97-108: bit-exact integers on random, disk and edge-touching sections; the
    per-layer form against the same count applied to every plane of a 3-D mask, for the lateral and for all six faces"""
    import math
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import perimeter_oracle as ref
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(4)
    secs = [rng.random((13, 9)) > 0.4, np.ones((5, 7), bool), np.zeros((4, 4), bool), rng.random((1, 17)) > 0.3]
    xs = np.arange(40) + 0.5 - 20
    secs.append(np.sqrt(xs[:, None] ** 2 + xs[None, :] ** 2) <= 17.3)
    for m in secs:
        assert hip.count_exposed_faces(m) == ref.count_exposed_faces(m)
    with pytest.raises(ValueError):
        hip.perimeter_ratio(np.zeros((4, 4), bool), 1e-3, 1.0)
    g = hip.perimeter_ratio(secs[-1], 1.0, 2.0 * math.pi * 17.3)
    assert abs(g - math.pi / 4.0) < 0.02
    m3 = rng.random((11, 10, 23)) > 0.35
    lat = hip.exposed_faces_per_layer(m3)
    assert lat.dtype == np.int64 and lat.shape == (23,)
    assert [int(v) for v in lat] == [ref.count_exposed_faces(m3[:, :, k]) for k in range(23)]
    allf = hip.exposed_faces_per_layer(m3, faces=('x-', 'x+', 'y-', 'y+', 'z-', 'z+'))
    want = sum(orc.exposed_mask(m3, f).sum(axis=(0, 1)) for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+'))
    assert np.array_equal(allf, want)
    with pytest.raises(ValueError):
        hip.exposed_faces_per_layer(m3, faces=('w+',))
