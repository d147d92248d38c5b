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
    assert _lib.lib.adi_abi_version() == 14


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
    with pytest.raises(ValueError, match='unknown zbc.kind_bot'):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.adi_cyl_plan_create(4, 4, 4, 0, 1e-3, 0.1, 1e-3, 1.0, 1.0, 1.0, 0.1, 0.0, 0.0, 7, 0,
                                                0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ctypes.byref(h)))
