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


@pytest.mark.parametrize('kind', ['solid', 'holes'])
def test_ctx_learns_the_no_fallback_promise(kind):
    """adi_ctx_step counts the units each sweep queues on the first step after a mask / pack change (adi_step_queued) and,
    when there are none, runs the later steps without the queue reset and the fallback launch.  Same numbers either way:
    several calls on a solid box (nothing queued: promise taken) and on a box with voids (units queued: never promised),
    a mask change in between, all against the oracle."""
    from adi_thermal_fields_amd import _lib
    from oracle import adi_oracle as orc
    lib = _lib.lib
    shape = (256, 16, 32)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    dt = 60.0 * dx * dx / alpha
    rng = np.random.default_rng(11)
    T0 = rng.uniform(20.0, 900.0, shape)
    masks = [np.ones(shape, bool), rng.random(shape) > 0.05] if kind == 'solid' else [rng.random(shape) > 0.05, np.ones(shape, bool)]
    ctx = ctypes.c_void_p()
    _lib.check(lib.adi_ctx_create(*shape, dx, 0, ctypes.byref(ctx)))
    try:
        hm = (ctypes.c_int * 6)(1, 1, 1, 1, 1, 1)
        hs = (ctypes.c_double * 6)(*([300.0] * 6))
        qm = (ctypes.c_int * 6)(0, 0, 0, 0, 0, 0)
        qs = (ctypes.c_double * 6)(0, 0, 0, 0, 0, 0)
        T = np.ascontiguousarray(T0)
        want = np.array(T0)
        for mask in masks:
            m8 = np.ascontiguousarray(mask).view(np.uint8)
            _lib.check(lib.adi_ctx_set_mask(ctx, m8.ctypes.data))
            _lib.check(lib.adi_ctx_build_coeffs(ctx, 7800.0, 490.0, hm, hs, None, qm, qs, None, None, None))
            _lib.check(lib.adi_ctx_upload_T(ctx, T.ctypes.data))
            for nsteps in (1, 3, 2):                    # the first call learns, the others use what it learnt
                _lib.check(lib.adi_ctx_step(ctx, 7800.0, 490.0, 54.0, dt, 0.5, 20.0, nsteps))
            got = np.empty(shape)
            _lib.check(lib.adi_ctx_download_T(ctx, got.ctypes.data))
            og = orc.Grid3D(*shape, dx, mask)
            om = orc.Material(7800.0, 490.0, 54.0); op = orc.Params(dt, 0.5)
            pk = orc.precompute_coeff_packs_unified(og, om, robin_h=300.0)
            for _ in range(6):
                want = orc.adi_step_numba_coeff(want, og, om, op, pk, Tinf=20.0)
            assert rel_linf(got, want) <= 1e-10
            T = np.ascontiguousarray(got)
    finally:
        lib.adi_ctx_destroy(ctx)


@pytest.mark.parametrize('shape', [(32, 256, 256), (256, 32, 64), (1040, 16, 16)])
def test_step_queued_reports_zero_for_sweeps_that_run_no_fast_kernel(shape):
    """adi_step_queued: a sweep that runs no FAST kernel (lines shorter than 64 rows; lines beyond 1024 rows, where the
    workspace holds c' / d' doubles) never touches the queue word, so its count must be reported as 0 -- not whatever the
    workspace held -- or adi_ctx_step would refuse the no-fallback promise for the whole mask / pack epoch on garbage.
    The workspace is filled with 0xff before the call; an all-solid box queues nothing anywhere."""
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import _lib
    lib = _lib.lib
    nx, ny, nz = shape
    dx = 1e-3
    grid = hip.Grid3D(nx, ny, nz, dx, np.ones(shape, bool))
    mat = hip.Material(7800.0, 490.0, 54.0)
    packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0)
    L = grid.layout
    T = hip.to_device(np.random.default_rng(3).uniform(20.0, 900.0, shape)).t
    out, ta, tb = L.empty(), L.empty(), L.empty()
    _, work, wb = grid.scratch(2)
    work.fill_(255)
    q = (ctypes.c_uint * 3)(7, 7, 7)
    coeff = _lib.ptr_array([p.d_coeff.data_ptr() for p in packs])
    qflux = _lib.ptr_array([None, None, None])
    alpha = 54.0 / (7800.0 * 490.0)
    _lib.check(lib.adi_step_queued(hip._p(T), hip._p(out), hip._p(ta), hip._p(tb), hip._p(grid.d_flags), coeff, None, None, qflux,
                                   _lib.SWEEP_LEAN, 1, nx, ny, nz, grid.sx, dx, 7800.0, 490.0, 54.0, 50.0 * dx * dx / alpha, 0.5, 20.0,
                                   None, hip._p(work), wb, hip._stream(), ctypes.cast(q, ctypes.c_void_p)))
    torch.cuda.synchronize()
    assert list(q) == [0, 0, 0], list(q)
    want = hip.adi_step_hip_coeff(hip.DeviceField(T), grid, mat, hip.Params(50.0 * dx * dx / alpha, 0.5), packs, Tinf=20.0).t
    assert torch.equal(out, want)
