"""GPU: the context half of the C ABI (library-owned device memory, host arrays) through raw ctypes,
as a non-Python consumer of include/adi_hip.h would drive it."""
import ctypes

import numpy as np
import pytest

import cases
from helpers import golden, rel_linf

pytestmark = pytest.mark.gpu


def test_ctx_roundtrip_kat2():
    from adi_thermal_fields_amd import _lib
    lib = _lib.lib
    c = cases.cart_case('kat2')
    g = golden('cart', 'kat2')
    nx, ny, nz = c['shape']
    ctx = ctypes.c_void_p()
    _lib.check(lib.adi_ctx_create(nx, ny, nz, c['dx'], 0, ctypes.byref(ctx)))
    try:
        mask = np.ascontiguousarray(c['mask']).view(np.uint8)
        _lib.check(lib.adi_ctx_set_mask(ctx, mask.ctypes.data))
        # step before build_coeffs is a state error, not a crash
        T0 = np.ascontiguousarray(c['T0'], dtype=np.float64)
        _lib.check(lib.adi_ctx_upload_T(ctx, T0.ctypes.data))
        with pytest.raises(_lib.AdiError):
            _lib.check(lib.adi_ctx_step(ctx, 7800.0, 490.0, 54.0, c['dt'], c['theta'], c['Tinf'], 1))
        hm = (ctypes.c_int * 6)(1, 1, 1, 1, 0, 0)
        hs = (ctypes.c_double * 6)(500.0, 500.0, 500.0, 500.0, 0.0, 0.0)
        qm = (ctypes.c_int * 6)(0, 0, 0, 0, 1, 0)
        qs = (ctypes.c_double * 6)(0, 0, 0, 0, 2e6, 0)
        dm = np.ascontiguousarray(c['dir_mask']).view(np.uint8)
        dv = np.full(c['shape'], 20.0)
        _lib.check(lib.adi_ctx_build_coeffs(ctx, 7800.0, 490.0, hm, hs, None, qm, qs, None,
                                            dm.ctypes.data, dv.ctypes.data))
        co = np.empty(c['shape']); qz = np.empty(c['shape'])
        _lib.check(lib.adi_ctx_download_pack(ctx, 2, co.ctypes.data, qz.ctypes.data))
        assert np.array_equal(co, g['coeff_z']) and np.array_equal(qz, g['qflux_z'])
        _lib.check(lib.adi_ctx_step(ctx, 7800.0, 490.0, 54.0, c['dt'], c['theta'], c['Tinf'], c['nsteps']))
        ms = ctypes.c_float(0)
        _lib.check(lib.adi_ctx_last_step_ms(ctx, ctypes.byref(ms)))
        assert ms.value > 0
        T = np.empty(c['shape'])
        _lib.check(lib.adi_ctx_download_T(ctx, T.ctypes.data))
        assert rel_linf(T, g['T_final']) <= 1e-10
    finally:
        lib.adi_ctx_destroy(ctx)


def test_device_info():
    from adi_thermal_fields_amd import _lib
    n = ctypes.c_int(0)
    _lib.check(_lib.lib.adi_device_count(ctypes.byref(n)))
    assert n.value >= 1
    name = ctypes.create_string_buffer(256)
    cu = ctypes.c_int(0); hbm = ctypes.c_size_t(0); lds = ctypes.c_size_t(0)
    _lib.check(_lib.lib.adi_device_info(0, name, ctypes.byref(cu), ctypes.byref(hbm), ctypes.byref(lds)))
    assert b'gfx950' in name.value and cu.value == 256
