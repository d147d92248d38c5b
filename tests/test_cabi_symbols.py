"""CPU: the C-ABI library loads and exports every symbol include/adi_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'adi_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(adi_[A-Za-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    from adi_thermal_fields_amd import _lib      # ImportError here = library not built
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), 'libadi_hip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'ctypes binding missing for %s' % n
    for n in _lib.SIGNATURES:
        assert n in names, '%s bound but not declared in include/adi_hip.h' % n
    assert _lib.lib.adi_abi_version() == 18


def test_argument_errors_without_gpu():
    """argument validation happens before any HIP call, so it is testable on a CPU-only box"""
    import ctypes
    import pytest
    from adi_thermal_fields_amd import _lib
    with pytest.raises(ValueError, match='bad face'):
        _lib.check(_lib.lib.adi_exposed_mask(ctypes.c_void_p(8), 2, 2, 2, 0, 9, ctypes.c_void_p(8), None))
    with pytest.raises(ValueError):
        _lib.check(_lib.lib.adi_sweep(5, 0, None, None, None, None, None, None, 1, 1, 1, 0, 0, 0.5, 1.0, 1.0, 0.0,
                                      None, None, None, None, None, 0, None))
    with pytest.raises(ValueError, match='adi_deferred_lines_apply'):          # K > nx
        _lib.check(_lib.lib.adi_deferred_lines_apply(ctypes.c_void_p(8), 4, 64, 64, ctypes.c_void_p(8), 3, ctypes.c_void_p(8), 5,
                                                     ctypes.c_void_p(8), 0, None, None))
    with pytest.raises(ValueError, match='lower boundary without its weights'):
        _lib.check(_lib.lib.adi_interface_deferred_lines(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None,
                                                         None, None, 16, ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None,
                                                         None))
    with pytest.raises(ValueError, match='unknown zbc.kind_bot'):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.adi_cyl_plan_create(4, 4, 4, 0, 1e-3, 0.1, 1e-3, 1.0, 1.0, 1.0, 0.1, 0.0, 0.0, 7, 0,
                                                0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ctypes.byref(h)))


def test_face_constants_are_the_pack_values_bit_for_bit():
    """adi_face_constants (host arithmetic, no GPU): the per-face scalars a sweep is given instead of loading coefficients
    must be the numbers precompute_coeff_packs_unified stores -- checked against the pinned oracle's packs on a mask whose
    cells are exposed on the minus face, the plus face, both, or neither, for every axis"""
    import ctypes
    import numpy as np
    from adi_thermal_fields_amd import _lib
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(5)
    shape = (7, 6, 8)
    mask = rng.random(shape) > 0.35
    dx, rho, cp = 1.3e-3, 7800.0, 490.0
    h = {'x-': 410.0, 'x+': 37.5, 'y-': 0.0, 'y+': 999.0, 'z-': 12.25, 'z+': 500.0}
    q = {'x+': 2e5, 'z-': -3.5e4}
    grid = orc.Grid3D(*shape, dx, mask)
    packs = orc.precompute_coeff_packs_unified(grid, orc.Material(rho, cp, 54.0), neumann=q, robin_h=h)
    faces = ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')
    I6, D6 = ctypes.c_int * 6, ctypes.c_double * 6
    hm, hs = I6(*[1] * 6), D6(*[h[f] for f in faces])
    qm, qs = I6(*[1 if f in q else 0 for f in faces]), D6(*[q.get(f, 0.0) for f in faces])
    consts, valid = (ctypes.c_double * 12)(), (ctypes.c_int * 3)()
    _lib.check(_lib.lib.adi_face_constants(dx, rho, cp, hm, hs, qm, qs, consts, valid))
    assert list(valid) == [1, 1, 1]
    m = np.pad(mask, 1)
    for a in range(3):
        sl = lambda d: tuple(slice(1 + (d if i == a else 0), 1 + (d if i == a else 0) + shape[i]) for i in range(3))
        lo, hi = m[sl(-1)], m[sl(+1)]                      # the minus / plus neighbour along axis a is in the mask
        cm, cpl, qmn, qpl = consts[4 * a:4 * a + 4]
        want_c = np.where(mask, (0.0 + np.where(~lo, cm, 0.0)) + np.where(~hi, cpl, 0.0), 0.0)
        want_q = np.where(mask, (0.0 + np.where(~lo, qmn, 0.0)) + np.where(~hi, qpl, 0.0), 0.0)
        assert np.array_equal(want_c, packs[a].coeff), a
        assert np.array_equal(want_q, packs[a].qflux), a
    # a per-voxel field on one face of an axis: no constants for that axis
    hm2 = I6(1, 2, 1, 1, 1, 1)
    _lib.check(_lib.lib.adi_face_constants(dx, rho, cp, hm2, hs, qm, qs, consts, valid))
    assert list(valid) == [0, 1, 1]
