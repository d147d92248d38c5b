"""GPU: the HIP path (through the ctypes C ABI) against the golden vectors of the reference and the
oracle.  fp64 tolerance of BASELINE.json: <= 1e-10 relative L-inf (max|dT| / max|T|); the bit-exact
stages (coefficient build, explicit stage) are asserted bit-exact."""
import numpy as np
import pytest

import cases
from helpers import golden, rel_linf, run_cart_case, run_cyl_case

pytestmark = pytest.mark.gpu

TOL = 1e-10   # BASELINE.json north_star: <= 1e-10 relative L-inf vs the Numba reference


@pytest.fixture(scope='module')
def hip():
    import adi_thermal_fields_amd.adi3d_hip_coeff as m
    return m


@pytest.fixture(scope='module')
def hipcyl():
    import adi_thermal_fields_amd.adi3d_hip_cyl as m
    return m


@pytest.mark.parametrize('name', cases.CART_CASES)
def test_cart_vs_golden(hip, name):
    c = cases.cart_case(name)
    g = golden('cart', name)
    out = run_cart_case(hip, c)
    packs = out['packs0']
    for ax, p in zip('xyz', packs):
        assert np.array_equal(p.coeff, g['coeff_' + ax]), 'coeff_' + ax     # bit-exact (same order, IEEE div)
        assert np.array_equal(p.qflux, g['qflux_' + ax]), 'qflux_' + ax
    assert np.array_equal(packs[0].dir_mask, g['dir_mask'])
    assert np.array_equal(packs[0].dir_val, g['dir_val'])
    for f in cases.FACES:
        assert np.array_equal(hip.exposed_mask(c['mask'], f), g['exposed_' + f]), f
    for key in g.files:
        if key.startswith('T_'):
            assert rel_linf(out[key], g[key]) <= TOL, (key, rel_linf(out[key], g[key]))
            # off-mask cells are never touched by a step
    if c['births'] is None:
        off = ~c['mask']
        assert np.array_equal(out['T_final'][off], c['T0'][off])


@pytest.mark.parametrize('name', ['holes_mixed', 'kat2', 'long_line_70'])
def test_hand_built_packs_vs_golden(hip, name):
    """AxisCoeffPack(coeff, dir_mask, dir_val, qflux) built by the caller from host arrays (adi3d_numba_coeff.py:29-36:
    the constructor copies its arguments) instead of precompute_coeff_packs_unified: read densely by the sweeps
    (nothing is known about where they are zero), same fields as the reference's packs give"""
    c = cases.cart_case(name)
    g = golden('cart', name)
    nx, ny, nz = c['shape']
    grid = hip.Grid3D(nx, ny, nz, c['dx'], c['mask'])
    mat = hip.Material(**c['mat'])
    prm = hip.Params(c['dt'], c['theta'])
    srcs = [np.array(g['coeff_' + ax]) for ax in 'xyz']
    packs = tuple(hip.AxisCoeffPack(srcs[i], g['dir_mask'], g['dir_val'], g['qflux_' + ax]) for i, ax in enumerate('xyz'))
    assert not any(p.sparse_ok for p in packs)
    srcs[0][...] = -1.0                                      # the pack holds a copy
    assert np.array_equal(packs[0].coeff, g['coeff_x'])
    assert np.array_equal(packs[2].qflux, g['qflux_z']) and np.array_equal(packs[1].dir_mask, g['dir_mask'])
    p_noq = hip.AxisCoeffPack(g['coeff_x'], g['dir_mask'], g['dir_val'])          # qflux=None -> zeros (:36)
    assert np.array_equal(p_noq.qflux, np.zeros(c['shape']))
    T = np.array(c['T0'], dtype=np.float64)
    for s in range(c['nsteps']):
        T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=c['Tinf'])
        if s == 0:
            assert rel_linf(T, g['T_step1']) <= TOL, rel_linf(T, g['T_step1'])
    assert rel_linf(T, g['T_final']) <= TOL, rel_linf(T, g['T_final'])


def test_float32_state_is_upcast(hip):
    """waam_from_stl_v7_mm.py:408-413 (--precision float32) hands the step a float32 field: it is up-cast on the way
    in, the step runs in fp64 and returns a float64 array -- the same bits as stepping the up-cast copy"""
    c = cases.cart_case('holes_mixed')
    nx, ny, nz = c['shape']
    grid = hip.Grid3D(nx, ny, nz, c['dx'], c['mask'])
    mat = hip.Material(**c['mat'])
    prm = hip.Params(c['dt'], c['theta'])
    packs = hip.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    T32 = np.asarray(c['T0'], dtype=np.float32)
    a = hip.adi_step_hip_coeff(T32, grid, mat, prm, packs, Tinf=c['Tinf'])
    b = hip.adi_step_hip_coeff(T32.astype(np.float64), grid, mat, prm, packs, Tinf=c['Tinf'])
    assert a.dtype == np.float64 and np.array_equal(a, b)
    assert T32.dtype == np.float32                                        # the caller's array is untouched


def test_stale_packs_after_mask_rebind_follow_the_reference(hip):
    """`grid.mask = new` WITHOUT rebuilding the packs: the reference keeps using the old coefficient arrays with the
    live mask (sweep_axis* read coeff at every in-mask cell, adi3d_numba_coeff.py:150-162).  The device packs are then
    read densely (their sparsity pattern belongs to the old mask) and the result is the oracle's."""
    from oracle import adi_oracle as orc
    c = cases.cart_case('long_line_70')
    nx, ny, nz = c['shape']
    rng = np.random.default_rng(5)
    new_mask = c['mask'].copy()
    new_mask[rng.random(c['shape']) > 0.97] = False                       # cells that were exposed become interior and
    new_mask[:, :, nz // 2] = True                                        # vice versa; some holes are filled
    outs = []
    for api in (hip, orc):
        grid = api.Grid3D(nx, ny, nz, c['dx'], c['mask'])
        mat = api.Material(**c['mat'])
        prm = api.Params(c['dt'], c['theta'])
        packs = api.precompute_coeff_packs_unified(grid, mat, neumann=c['neumann'], robin_h=c['robin_h'])
        grid.mask = new_mask
        step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
        T = np.array(c['T0'], dtype=np.float64)
        for _ in range(2):
            T = step(T, grid, mat, prm, packs, Tinf=c['Tinf'])
        outs.append(T)
    assert rel_linf(outs[0], outs[1]) <= TOL, rel_linf(outs[0], outs[1])


def test_packs_of_one_grid_on_another_grid_follow_the_reference(hip):
    """Packs built on grid A, stepped on a same-shape grid B with a DIFFERENT mask: the reference reads coeff at every
    in-mask cell of B (adi3d_numba_coeff.py:150-162), so A's sparsity pattern must not be applied to B's flags, and the
    no-fallback promise A's packs learnt on A must not travel either (under a false promise the FAST kernel would leave
    tiles unwritten).  Mask versions come from one process-wide counter, so no two grids ever share one."""
    from oracle import adi_oracle as orc
    shape = (256, 16, 64)
    rng = np.random.default_rng(17)
    mask_a = np.ones(shape, bool)
    mask_b = rng.random(shape) > 0.06                                     # voids everywhere: surface units get queued
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    T0 = rng.uniform(20.0, 900.0, shape)
    outs = []
    for api in (hip, orc):
        ga = api.Grid3D(*shape, dx, mask_a)
        gb = api.Grid3D(*shape, dx, mask_b)
        mat = api.Material(7800.0, 490.0, 54.0)
        prm = api.Params(150.0 * dx * dx / alpha, 0.5)
        packs = api.precompute_coeff_packs_unified(ga, mat, robin_h=350.0)
        step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
        if api is hip:
            assert ga.mask_version != gb.mask_version
            W = hip.to_device(T0)
            for _ in range(4):                                            # A learns the promise (all-solid: nothing queued)
                W = step(W, ga, mat, prm, packs, Tinf=20.0)
            assert all(v is True for p in packs for v in p._nofb.values())
        T = np.array(T0)
        for _ in range(2):
            T = step(T, gb, mat, prm, packs, Tinf=20.0)
        outs.append(T)
    assert np.array_equal(outs[0][~mask_b], T0[~mask_b])
    assert rel_linf(outs[0], outs[1]) <= TOL, rel_linf(outs[0], outs[1])


def test_in_place_mask_change_is_seen_by_the_numpy_step(hip):
    """The reference reads grid.mask at every step (adi3d_numba_coeff.py:294-301): a mask mutated IN PLACE, with no
    `grid.mask = ...` and no pack rebuild, changes the next step.  Called like the reference (NumPy in, NumPy out) the HIP
    step re-uploads and compares the host mask, so it follows; the stale packs are read densely, as the reference does."""
    from oracle import adi_oracle as orc
    c = cases.cart_case('long_line_70')
    rng = np.random.default_rng(9)
    flip = rng.random(c['shape']) > 0.97
    outs = []
    for api in (hip, orc):
        grid = api.Grid3D(*c['shape'], c['dx'], c['mask'])
        mat = api.Material(**c['mat'])
        prm = api.Params(c['dt'], c['theta'])
        packs = api.precompute_coeff_packs_unified(grid, mat, neumann=c['neumann'], robin_h=c['robin_h'])
        step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
        T = np.array(c['T0'], dtype=np.float64)
        T = step(T, grid, mat, prm, packs, Tinf=c['Tinf'])
        grid.mask[flip] = ~grid.mask[flip]                    # in place: no assignment, no rebuild
        for _ in range(2):
            T = step(T, grid, mat, prm, packs, Tinf=c['Tinf'])
        outs.append(T)
    assert rel_linf(outs[0], outs[1]) <= TOL, rel_linf(outs[0], outs[1])


def test_second_pack_set_on_an_unchanged_grid_keeps_the_first_fresh(hip):
    """precompute_coeff_packs_unified re-uploads the mask every time (the documented synchronisation point); when the
    mask has not changed the flags and the mask version stand, so packs built earlier (a Dirichlet-vs-Robin comparison,
    packs rebuilt with another h) keep their sparse reads; an in-place change of the host mask is still picked up."""
    c = cases.cart_case('long_line_70')
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = hip.Material(**c['mat'])
    p1 = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0)
    v1 = grid.mask_version
    p2 = hip.precompute_coeff_packs_unified(grid, mat, robin_h=900.0)
    assert grid.mask_version == v1
    assert all(hip._sparse_arg(grid, p, False) & 1 for p in p1 + p2)
    m = grid.mask
    m[1, 1, 1] = not m[1, 1, 1]                                           # mutated in place, then the packs are rebuilt
    p3 = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0)
    assert grid.mask_version != v1
    assert not any(hip._sparse_arg(grid, p, False) & 1 for p in p1) and all(hip._sparse_arg(grid, p, False) & 1 for p in p3)


@pytest.mark.parametrize('variant,dense', [(0, True), (0, False), (None, False)])
def test_cart_stages_vs_golden(hip, variant, dense):
    """every stage alone, fed with the reference's own previous stage; variant 0 + dense forces the
    general-pack kernel that reads every array in full"""
    c = cases.cart_case(cases.CART_STAGE_CASE)
    g = golden('cart', cases.CART_STAGE_CASE)
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = hip.Material(**c['mat']); prm = hip.Params(c['dt'], c['theta'])
    packs = hip.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    R0 = hip.adi_explicit_rhs(c['T0'], grid, mat, prm)
    assert np.array_equal(R0, g['R0'])          # explicit stage is bit-exact (contraction off, same order)
    prev = {'U': 'R0', 'V': 'U', 'W': 'V'}
    for ax, nm in enumerate('UVW'):
        got = hip.adi_sweep_axis(ax, g[prev[nm]], grid, mat, prm, packs[ax], Tinf=c['Tinf'], variant=variant, dense=dense)
        assert rel_linf(got, g[nm]) <= TOL, (nm, rel_linf(got, g[nm]))


def test_cart_config1_64(hip):
    """BASELINE.json configs[0]: 64^3 Dirichlet, 100 steps, vs planes of the reference run."""
    try:
        g = golden('cart', 'config1_64')
    except FileNotFoundError:
        pytest.skip('config1_64 golden not generated')
    c = cases.cart_case('config1_64')
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = hip.Material(**c['mat']); prm = hip.Params(c['dt'], c['theta'])
    packs = hip.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    T = hip.to_device(c['T0'])
    for _ in range(c['nsteps']):
        T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=c['Tinf'])
    T = T.get()
    nx, ny, nz = c['shape']
    for got, key in ((T[nx // 2], 'plane_i'), (T[:, ny // 2], 'plane_j'), (T[:, :, nz // 4], 'plane_k')):
        assert rel_linf(got, g[key]) <= TOL, (key, rel_linf(got, g[key]))
    assert abs(T.sum() - float(g['T_sum'])) <= 1e-10 * abs(float(g['T_sum']))


def test_bad_face(hip):
    with pytest.raises(ValueError):
        hip.exposed_mask(np.ones((2, 2, 2), bool), 'w+')


def test_input_not_modified_and_new_array(hip):
    c = cases.cart_case('kat1')
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = hip.Material(**c['mat']); prm = hip.Params(c['dt'], c['theta'])
    packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=c['robin_h'])
    T0 = c['T0'].copy()
    T1 = hip.adi_step_hip_coeff(T0, grid, mat, prm, packs, Tinf=20.0)
    assert np.array_equal(T0, c['T0']) and T1 is not T0 and isinstance(T1, np.ndarray)
    D0 = hip.to_device(T0)
    D1 = hip.adi_step_numba_coeff(D0, grid, mat, prm, packs, Tinf=20.0)   # the reference's name works too
    assert isinstance(D1, hip.DeviceField) and np.array_equal(D0.get(), T0)
    assert np.array_equal(D1.get(), T1)      # same kernels, same bits, host or device resident


@pytest.mark.parametrize('name', cases.CYL_CASES)
def test_cyl_vs_golden(hipcyl, name):
    c = cases.cyl_case(name)
    g = golden('cyl', name)
    out = run_cyl_case(hipcyl, c)
    for key in ('T_step1', 'T_final'):
        assert rel_linf(out[key], g[key]) <= TOL, (key, rel_linf(out[key], g[key]))


@pytest.mark.parametrize('shape', [(32, 64, 128), (12, 8, 40), (16, 1, 64)])
def test_cyl_graph_replayed_loop_matches_plain_steps(hipcyl, shape):
    """StagedCylStepper: step() with per-sweep launches (adi_cyl_sweep), run() with plain launches and run() replayed from
    a HIP graph are the same kernels on the same buffers -- bit-identical -- and equal adi_step (one adi_cyl_step call)."""
    nr, nphi, nz = shape
    g = hipcyl.GridCyl(nr, nphi, nz, 2.5e-4, 2 * np.pi / max(nphi, 1), 2.5e-4, nr * 2.5e-4)
    mat = hipcyl.Material(7800.0, 490.0, 54.0)
    prm = hipcyl.Params(0.05, 1.0, "be")
    rr = hipcyl.RobinR(400.0, 20.0)
    zbc = hipcyl.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0)
    T0 = np.random.default_rng(nr + nz).uniform(20.0, 1000.0, shape)
    st = hipcyl.StagedCylStepper(g, mat, prm, rr, zbc)
    T = hipcyl.to_device(T0)
    for _ in range(5):
        T = st.step(T)
    want = T.get()
    ref = hipcyl.to_device(T0)
    for _ in range(5):
        ref = hipcyl.adi_step(ref, g, mat, prm, rr, zbc)
    assert np.array_equal(want, ref.get())
    assert np.array_equal(st.run(hipcyl.to_device(T0), 5, graph=False).get(), want)
    assert np.array_equal(st.run(hipcyl.to_device(T0), 5, graph=True).get(), want)
    assert np.array_equal(st.run(hipcyl.to_device(T0), 5, graph=True).get(), want)      # the captured graph, replayed again


def test_cyl_bad_kind(hipcyl):
    c = cases.cyl_case('kat3')
    c['zbc'] = dict(kind_bot='bogus', kind_top='robin')
    with pytest.raises(ValueError):
        run_cyl_case(hipcyl, c)


@pytest.mark.parametrize('shape', [(512, 6, 24), (6, 24, 512), (10, 512, 16), (256, 8, 256),
                                   # segment counts that are not powers of two: padding lanes / segments in the FAST kernels
                                   (320, 6, 32), (6, 20, 320), (10, 384, 16), (192, 4, 96), (5, 3, 200), (200, 3, 16)])
def test_long_lines_fast_and_general_units(hip, shape):
    """lines of 256/512 rows (M = 4/8, 64 lanes per line): mostly solid, so the FAST kernels take most units, with
    holes, a Dirichlet plane, a Neumann face and Robin everywhere forcing queued GENERAL units in the same sweep"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape))
    mask = np.ones(shape, bool)
    for _ in range(12):          # a few voids
        i, j, k = (rng.integers(0, n) for n in shape)
        mask[max(0, i - 3):i + 3, max(0, j - 1):j + 2, max(0, k - 3):k + 3] = False
    dm = np.zeros(shape, bool); dm[:, 0, :] = mask[:, 0, :]
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=dm, dir_value=150.0,
             neumann={'z+': 3e5, 'x-': rng.uniform(0, 2e5, shape)}, robin_h=400.0, Tinf=20.0, theta=0.5,
             dt=250.0 * dx * dx / alpha, nsteps=2, births=None)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    c2 = dict(c, dir_mask=None, dir_value=None, neumann=None)      # lean variant
    got = run_cart_case(hip, c2)['T_final']
    want = run_cart_case(orc, c2)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)


def test_random_shapes_and_bcs(hip):
    """ragged / odd / prime sizes, random masks and BC mixes: every kernel variant and tiling choice
    (scalar-load instantiations, padded segments, FAST + queued GENERAL units, generic fallbacks)"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(2026)
    sizes = [1, 2, 3, 5, 7, 8, 13, 16, 17, 31, 32, 33, 48, 64, 65, 67, 96, 127, 128, 129, 130, 200, 256, 257]
    worst = 0.0
    for trial in range(28):
        shape = tuple(int(rng.choice(sizes)) for _ in range(3))
        while shape[0] * shape[1] * shape[2] > 600000:
            shape = tuple(int(rng.choice(sizes)) for _ in range(3))
        fill = rng.choice([1.0, 0.97, 0.7])
        mask = rng.random(shape) < fill
        dx = 1e-3
        alpha = 54.0 / (7800.0 * 490.0)
        kind = trial % 4
        dm = (rng.random(shape) > 0.97) & mask if kind in (0, 2) else None
        neumann = {'y+': 2e5, 'z-': rng.uniform(0, 3e5, shape)} if kind in (0, 1) else None
        robin = [None, 300.0, {'x-': 100.0, 'z+': rng.uniform(0, 900, shape)}, rng.uniform(0, 500, shape)][trial % 4]
        c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
                 T0=rng.uniform(20.0, 1200.0, shape), dir_mask=dm, dir_value=(rng.uniform(50, 90, shape) if dm is not None else None),
                 neumann=neumann, robin_h=robin, Tinf=25.0, theta=float(rng.choice([0.5, 1.0])),
                 dt=float(rng.choice([0.7, 40.0, 900.0])) * dx * dx / alpha, nsteps=2, births=None)
        got = run_cart_case(hip, c)['T_final']
        want = run_cart_case(orc, c)['T_final']
        err = rel_linf(got, want)
        worst = max(worst, err)
        assert err <= TOL, (shape, kind, err)
    print('worst rel_linf over random cases: %.3e' % worst)


@pytest.mark.parametrize('name', ['kat2', 'edge_shapes'])
def test_surface_impulse(hip, name):
    """apply_surface_impulse_Q (adi3d_numba_coeff.py:304-320), host array and device field, in place"""
    c = cases.cart_case(name)
    g = golden('cart', name)
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask']); mat = hip.Material(**c['mat'])
    for f in cases.FACES:
        T = np.array(c['T0'], dtype=np.float64)
        hip.apply_surface_impulse_Q(T, grid, mat, 3.5e4, face=f)
        assert np.array_equal(T, g['impulse_' + f]), f
        D = hip.to_device(c['T0'])
        hip.apply_surface_impulse_Q(D, grid, mat, 3.5e4, face=f)
        assert np.array_equal(D.get(), g['impulse_' + f]), f
    with pytest.raises(ValueError):
        hip.apply_surface_impulse_Q(np.zeros(c['shape']), grid, mat, 1.0, face='q')


def test_device_field_ndarray_surface(hip):
    """the slice of the ndarray surface the reference's drivers use on their temperature field"""
    rng = np.random.default_rng(1)
    A = rng.uniform(0, 100, (6, 5, 8))
    D = hip.to_device(A)
    assert D.shape == (6, 5, 8) and D.ndim == 3 and D.size == 240
    assert np.array_equal(np.asarray(D), A) and np.array_equal(D.get(), A)
    assert np.array_equal(D[2, 3, :], A[2, 3, :]) and D[1, 2, 3] == A[1, 2, 3]
    assert D.min() == A.min() and D.max() == A.max() and abs(D.sum() - A.sum()) < 1e-9
    idx = np.where(A > 90.0)
    D[idx] = 7.0; A[idx] = 7.0                       # T[idx] = Ts        (waam_from_stl_v7_mm.py:489-493)
    D[1:3, 0:1, 2:5] = 3.0; A[1:3, 0:1, 2:5] = 3.0   # T[x0:x1, yi:yi+1, z0:z1] = T_track_init (single_track_on_plate.py:166)
    E = D.copy(); E[...] = D                         # T[...] = step(...)  (single_track_on_plate.py:175)
    assert np.array_equal(E.get(), A)
    m = A > 50.0
    D[m] = 1.0; A[m] = 1.0
    assert np.array_equal(D.get(), A)


@pytest.mark.parametrize('shape', [(768, 3, 16), (3, 1024, 8), (4, 6, 1024), (1500, 2, 8), (2, 1300, 4), (3, 2, 2050)])
def test_very_long_lines(hip, shape):
    """513..1024 rows: 16 rows per thread; beyond 1024: the generic one-thread-per-line kernel with HBM scratch"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(shape[0] + shape[2])
    mask = rng.random(shape) > 0.03
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None,
             neumann={'x+': 1e5}, robin_h={'x-': 300.0, 'y-': 200.0, 'z+': 500.0, 'z-': 100.0}, Tinf=20.0, theta=0.5,
             dt=400.0 * dx * dx / alpha, nsteps=2, births=None)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    c2 = dict(c, neumann=None, mask=np.ones(shape, bool))        # solid block: FAST kernels with 16 rows per thread
    got = run_cart_case(hip, c2)['T_final']
    want = run_cart_case(orc, c2)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)


@pytest.mark.parametrize('shape,fill', [((512, 6, 40), 1.0), ((512, 5, 32), 0.97), ((256, 3, 64), 1.0), ((64, 33, 48), 0.9), ((257, 4, 18), 1.0),
                                        ((320, 4, 32), 1.0), ((384, 3, 16), 0.98), ((96, 5, 48), 1.0),
                                        ((16, 16, 16), 0.8), ((128, 9, 130), 1.0), ((3, 7, 5), 0.7), ((1, 4, 6), 1.0)])
@pytest.mark.parametrize('bc', ['lean', 'general'])
def test_fused_explicit_sweep0_is_bit_identical(hip, shape, fill, bc):
    """adi_explicit_sweep0 (explicit stage evaluated inside the loads of the axis-0 sweep, FAST and queued GENERAL
    tiles, sparse and dense packs) against adi_explicit_rhs + adi_sweep(axis 0): the same arithmetic in the same
    order, so the two must agree bit for bit; and against the oracle's U at the parity bar."""
    from adi_thermal_fields_amd import _lib
    rng = np.random.default_rng(sum(shape) + len(bc))
    mask = rng.random(shape) < fill
    mask[shape[0] // 2:, :, :] |= rng.random((shape[0] - shape[0] // 2,) + shape[1:]) < 0.995   # mostly solid half
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    grid = hip.Grid3D(*shape, dx, mask)
    # (fused_supported() declines boxes the FAST fused kernel cannot tile -- e.g. nz % 16 != 0 -- because the step is
    # faster unfused there; the entry point itself still serves them, through the GENERAL fused kernel)
    # (... and what counts is the physical box the layout gives the kernels: 512 x 6 x 40 runs as 512 x 6 x 48)
    px, _, pz, _ = grid.layout.pd
    assert hip.fused_supported(grid) == (px < 64 or (pz % 16 == 0 and (px % 8 == 0 if px <= 256 else px % 16 == 0)))
    mat = hip.Material(7800.0, 490.0, 54.0)
    prm = hip.Params(150.0 * dx * dx / alpha, 0.5)
    kw = dict(robin_h=350.0)
    if bc == 'general':
        dm = np.zeros(shape, bool); dm[:, 0, :] = mask[:, 0, :]
        kw.update(dir_mask=dm, dir_value=77.0, neumann={'x+': 2e5, 'z-': rng.uniform(0, 1e5, shape)})
    packs = hip.precompute_coeff_packs_unified(grid, mat, **kw)
    T0 = rng.uniform(20.0, 1500.0, shape)
    R0 = hip.adi_explicit_rhs(T0, grid, mat, prm)
    for variant, dense in ((None, False), (_lib.SWEEP_GENERAL, True)):
        two = hip.adi_sweep_axis(0, R0, grid, mat, prm, packs[0], Tinf=25.0, variant=variant, dense=dense)
        one = hip.adi_explicit_sweep_axis0(T0, grid, mat, prm, packs[0], Tinf=25.0, variant=variant, dense=dense)
        # lines of 320 / 384 rows: the unfused FAST kernel cuts them into 16 segments of 20 / 24 rows (exact fit,
        # adi_sweep_strided_x.hip), the fused one -- 16 rows per thread at most -- into 20 / 24 segments of 16: two
        # partitions of the same systems, equal to rounding
        exact_fit = shape[0] in (320, 384, 448, 640, 768, 896) and not dense
        if (shape[2] % 16 == 0 or dense) and not exact_fit:
            # the same kernel pair (FAST interior tiles, GENERAL surface tiles) serves both forms: bit-identical
            assert np.array_equal(one, two), (variant, dense, float(np.abs(one - two).max()))
        else:
            # nz not a multiple of the 16-line tile: the fused form runs every tile through the GENERAL kernel, the
            # two-kernel form its solid tiles through the FAST one -- equal to rounding
            assert rel_linf(one, two) <= 1e-13, rel_linf(one, two)
    # a view into a larger buffer (halo planes around it, like a slab): neighbours outside the box are read
    # wherever the flags say so -- here they never do, and the result must not change
    import torch
    L = grid.layout
    big = torch.full(((shape[0] + 2) * L.sx,), float('nan'), dtype=torch.float64, device='cuda')
    view = big.as_strided(L.shape, L.strides, L.sx)
    view.copy_(torch.from_numpy(T0))
    out = L.empty()
    hip._explicit_sweep0_into(view, out, grid, mat, prm, packs[0], 25.0)
    assert np.array_equal(out.cpu().numpy(), one if variant is None else
                          hip.adi_explicit_sweep_axis0(T0, grid, mat, prm, packs[0], Tinf=25.0))


@pytest.mark.parametrize('shape,nsteps', [((24, 20, 32), 7), ((64, 64, 64), 10), ((16, 16, 16), 1)])
def test_graph_replayed_nsub_loop_matches_plain_steps(hip, shape, nsteps):
    """StagedStepper.run (two steps captured into a HIP graph and replayed) == the same number of plain steps, bit for
    bit; the graph is rebuilt when dt changes (drivers mutate params.dt between segments) and when the mask changes"""
    rng = np.random.default_rng(shape[0] + nsteps)
    mask = rng.random(shape) < 0.9
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    grid = hip.Grid3D(*shape, dx, mask)
    mat = hip.Material(7800.0, 490.0, 54.0)
    prm = hip.Params(30.0 * dx * dx / alpha, 0.5)
    packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0, neumann={'z-': 1e5})
    T0 = rng.uniform(20.0, 900.0, shape)
    st = hip.StagedStepper(grid, mat, prm, packs, Tinf=25.0)

    def plain(T, n):
        T = hip.to_device(T)
        for _ in range(n):
            T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=25.0)
        return T.get()
    got = st.run(hip.to_device(T0), nsteps).get()
    assert np.array_equal(got, plain(T0, nsteps))
    assert np.array_equal(st.run(T0, nsteps, graph=False).get(), got)
    prm.dt *= 0.5                                            # new segment: same stepper object, new dt
    got2 = st.run(hip.to_device(got), nsteps + 1).get()
    assert np.array_equal(got2, plain(got, nsteps + 1))
    mask2 = mask.copy(); mask2[:, :, : shape[2] // 2] = True  # birth: mask + packs change
    grid.mask = mask2
    packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0, neumann={'z-': 1e5})
    st = hip.StagedStepper(grid, mat, prm, packs, Tinf=25.0)
    assert np.array_equal(st.run(got2, 4).get(), plain(got2, 4))


def _ellipsoid(shape, holes=False):
    g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
    m = (g[0] / 0.46) ** 2 + (g[1] / 0.42) ** 2 + (g[2] / 0.47) ** 2 <= 1.0
    if holes:
        m &= ~(((g[0] - 0.1) / 0.12) ** 2 + (g[1] / 0.15) ** 2 + ((g[2] + 0.05) / 0.1) ** 2 <= 1.0)   # an inner void
    return m


@pytest.mark.parametrize('shape', [(512, 40, 48), (48, 512, 32), (40, 48, 512), (320, 24, 320), (96, 80, 96)])
@pytest.mark.parametrize('bc', ['lean', 'neumann', 'general'])
def test_curved_solids_surface_segments(hip, shape, bc):
    """ellipsoids (with an inner void): every line through the solid starts and ends inside a register segment, the
    case the FAST kernels take as TAIL / HEAD segments (one reciprocal chain along the run) instead of queueing the unit;
    long lines along each axis in turn so that the 8- and 16-row tilings of all three kernel families see them"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape) + len(bc))
    mask = _ellipsoid(shape, holes=(bc != 'lean'))
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    kw = dict(dir_mask=None, dir_value=None, neumann=None)
    if bc != 'lean':
        kw['neumann'] = {'x-': 2e5, 'y+': rng.uniform(0, 1e5, shape), 'z-': 1.5e5}
    if bc == 'general':
        dm = np.zeros(shape, bool); dm[:, :, : shape[2] // 3] = mask[:, :, : shape[2] // 3] & (rng.random(mask[:, :, : shape[2] // 3].shape) < 0.02)
        kw.update(dir_mask=dm, dir_value=rng.uniform(30.0, 60.0, shape))
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), robin_h=rng.uniform(20.0, 400.0, shape), Tinf=20.0, theta=0.5,
             dt=120.0 * dx * dx / alpha, nsteps=2, births=None, **kw)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    assert np.array_equal(got[~mask], c['T0'][~mask])


def _thin_walls(shape, axis, rng, walls=((1, 0.12), (2, 0.2), (4, 0.29), (6, 0.38), (9, 0.47))):
    """tubes along `axis` with walls of 1 .. 9 voxels, a thin plate and a few isolated single voxels: every line across
    a wall is a short run that starts and ends inside one register segment (ISLAND segments of the FAST kernels; walls
    of 9 voxels and runs that straddle a segment boundary go the TAIL / HEAD / queue way)"""
    g = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing='ij')
    oth = [a for a in range(3) if a != axis]
    m = np.zeros(shape, bool)
    for wall, frac in walls:
        r = np.sqrt(((g[oth[0]] - shape[oth[0]] / 2.0) / shape[oth[0]]) ** 2 + ((g[oth[1]] - shape[oth[1]] / 2.0) / shape[oth[1]]) ** 2)
        w = wall / float(min(shape[oth[0]], shape[oth[1]]))
        m |= (r >= frac - w) & (r <= frac)
    sl = [slice(None)] * 3
    sl[axis] = slice(shape[axis] // 3, shape[axis] // 3 + 3)          # a 3-voxel plate across the tube axis
    m[tuple(sl)] = True
    sl[axis] = slice(0, 2)                                              # and one touching the box face
    m[tuple(sl)] = True
    m |= rng.random(shape) < 0.002                                      # isolated voxels
    return m


@pytest.mark.parametrize('cfl', [1e-13, 1e-9, 1e-3, 0.3, 3e4])
def test_curved_solid_over_the_range_of_time_steps(hip, cfl):
    """the surface-segment lanes of the FAST kernels condense their run with a growing three-term recurrence instead of a
    reciprocal chain (adi_core.hpp, mixed_condense): growth (2 + 1/(theta*gamma))^15 per segment, so sweeps with
    theta*gamma < 1e-12 go to the GENERAL kernels; every regime against the oracle on an ellipsoid with a void"""
    from oracle import adi_oracle as orc
    shape = (256, 48, 256)
    g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
    mask = ((g[0] / 0.47) ** 2 + (g[1] / 0.49) ** 2 + (g[2] / 0.48) ** 2 <= 1.0) & \
        ~((g[0] / 0.12) ** 2 + (g[1] / 0.2) ** 2 + (g[2] / 0.1) ** 2 <= 1.0)
    rng = np.random.default_rng(41)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1500.0, shape),
             dir_mask=None, dir_value=None, neumann={'z-': 2e5}, robin_h=350.0, Tinf=20.0, theta=0.5,
             dt=cfl * dx * dx / alpha, nsteps=2, births=None)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert np.array_equal(got[~mask], want[~mask])
    assert rel_linf(got, want) <= TOL, (cfl, rel_linf(got, want))


@pytest.mark.parametrize('cfl', [1e-11, 1e-10, 4e-10, 1.9e-9, 3e-9])
@pytest.mark.parametrize('shape', [(32, 512, 48), (32, 480, 48), (512, 16, 48)])
def test_tiny_time_steps_on_long_lines_crossing_a_curved_surface(hip, shape, cfl):
    """theta*gamma between 5e-12 and 1.5e-9 on 480 / 512-row lines (30 / 32 rows per lane in the strided FAST kernels along
    axis 1, 16 in the fused one along axis 0) that cross the surface of an ellipsoid: the band in which mixed_condense's
    growing recurrence, (2 + 1/tg)^(rows-2), overflowed fp64 with round 3's gate of 1e-12 (inf / inf = NaN).  Below
    kMixedMinTg = 1e-9 the GENERAL kernels take the sweep; just above it the recurrence runs at its largest growth."""
    from oracle import adi_oracle as orc
    g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
    mask = ((g[0] / 0.47) ** 2 + (g[1] / 0.49) ** 2 + (g[2] / 0.48) ** 2 <= 1.0)
    rng = np.random.default_rng(int(cfl * 1e12) + shape[0])
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1500.0, shape),
             dir_mask=None, dir_value=None, neumann=None, robin_h=350.0, Tinf=20.0, theta=0.5,
             dt=cfl * dx * dx / alpha, nsteps=2, births=None)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert np.isfinite(got).all()
    assert np.array_equal(got[~mask], want[~mask])
    assert rel_linf(got, want) <= TOL, (cfl, rel_linf(got, want))


@pytest.mark.parametrize('shape,axis', [((256, 64, 48), 2), ((256, 48, 64), 1), ((64, 256, 48), 0), ((48, 64, 256), 0),
                                        ((96, 96, 96), 2), ((512, 32, 32), 1)])
@pytest.mark.parametrize('bc', ['lean', 'neumann'])
def test_thin_walls_island_segments(hip, shape, axis, bc):
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape) * 3 + axis + len(bc))
    mask = _thin_walls(shape, axis, rng)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    kw = dict(dir_mask=None, dir_value=None, neumann=None)
    if bc == 'neumann':
        kw['neumann'] = {'x+': 2e5, 'y-': rng.uniform(0, 1e5, shape), 'z+': 1.5e5}
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), robin_h=rng.uniform(20.0, 400.0, shape), Tinf=20.0, theta=0.5,
             dt=120.0 * dx * dx / alpha, nsteps=2, births=None, **kw)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    assert np.array_equal(got[~mask], c['T0'][~mask])


@pytest.mark.parametrize('shape,axis', [((256, 64, 64), 2), ((64, 256, 64), 0), ((64, 64, 256), 0), ((512, 64, 64), 1)])
def test_mid_walls_island_segments(hip, shape, axis):
    """walls of 10 and 13 voxels: runs of 9 .. 15 rows inside one 16-row segment (the odd-row pivots of island_solve are
    recomputed in the back substitution), the rest cross a segment boundary (TAIL + HEAD)"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape) * 5 + axis)
    mask = _thin_walls(shape, axis, rng, walls=((10, 0.24), (13, 0.49)))
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), robin_h=rng.uniform(20.0, 400.0, shape), Tinf=20.0, theta=0.5,
             dt=120.0 * dx * dx / alpha, nsteps=2, births=None, dir_mask=None, dir_value=None,
             neumann={'x-': 1e5, 'z+': rng.uniform(0, 1e5, shape)})
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    assert np.array_equal(got[~mask], c['T0'][~mask])


@pytest.mark.parametrize('shape', [(256, 64, 64), (64, 256, 64), (64, 64, 256), (96, 128, 96)])
@pytest.mark.parametrize('bc', ['lean', 'neumann'])
def test_slots_gap_segments(hip, shape, bc):
    """a solid block with slots 1 .. 12 voxels wide cut across every axis at irregular positions: lines cross a slot
    inside one register segment (GAP: a HEAD run and a TAIL run in one lane), at its edge (HEAD | TAIL) or twice (queued)"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape) * 7 + len(bc))
    mask = np.ones(shape, bool)
    for ax in range(3):
        pos = 5
        for w in (1, 3, 6, 9, 12, 2, 5):
            pos += int(rng.integers(9, 30))
            if pos + w >= shape[ax] - 3:
                break
            sl = [slice(None)] * 3
            sl[ax] = slice(pos, pos + w)
            o = (ax + 1) % 3
            sl[o] = slice(shape[o] // 4, shape[o])           # the slot does not cut the block in two
            mask[tuple(sl)] = False
            pos += w
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    kw = dict(dir_mask=None, dir_value=None, neumann=None)
    if bc == 'neumann':
        kw['neumann'] = {'x-': 2e5, 'y+': rng.uniform(0, 1e5, shape), 'z-': 1.5e5}
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), robin_h=rng.uniform(20.0, 400.0, shape), Tinf=20.0, theta=0.5,
             dt=120.0 * dx * dx / alpha, nsteps=2, births=None, **kw)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, rel_linf(got, want)
    assert np.array_equal(got[~mask], c['T0'][~mask])


def _blobs(shape, rng, nblob):
    """union of random ellipsoids and boxes, minus a few: smooth surfaces, flat faces, concavities and inner voids"""
    g = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing='ij')
    m = np.zeros(shape, bool)
    for b in range(nblob):
        c = [rng.uniform(0.15, 0.85) * n for n in shape]
        r = [max(1.5, rng.uniform(0.08, 0.45) * n) for n in shape]
        if rng.random() < 0.5:
            blob = sum(((g[a] - c[a]) / r[a]) ** 2 for a in range(3)) <= 1.0
        else:
            blob = np.ones(shape, bool)
            for a in range(3):
                blob &= np.abs(g[a] - c[a]) <= r[a]
        if b >= 2 and rng.random() < 0.35:
            m &= ~blob
        else:
            m |= blob
    return m


@pytest.mark.parametrize('seed', range(12))
def test_structured_masks_fuzz(hip, seed):
    """random unions / differences of ellipsoids and boxes on grids whose long axis takes the 8-, 16- and 32-row tilings
    (padding, off-mask, TAIL and HEAD segments, queued GENERAL units in one sweep), random BC mix, vs the oracle"""
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(1000 + seed)
    longs = [64, 96, 128, 160, 256, 320, 384, 512, 640]
    n = int(rng.choice(longs))
    others = [int(rng.choice([8, 16, 24, 32, 48])) for _ in range(2)]
    ax = seed % 3
    shape = others[:]
    shape.insert(ax, n)
    shape = tuple(shape)
    if seed % 4 == 3:                                    # two long axes
        shape = tuple(int(rng.choice([96, 128, 160])) if a != ax else min(n, 256) for a in range(3))
    mask = _blobs(shape, rng, int(rng.integers(2, 6)))
    if not mask.any():
        mask[tuple(s // 2 for s in shape)] = True
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    kind = seed % 3
    dm = (rng.random(shape) > 0.985) & mask if kind == 0 else None
    neumann = {'x+': 1e5, 'z-': rng.uniform(0, 2e5, shape), 'y-': 5e4} if kind in (0, 1) else None
    robin = [250.0, {'x-': 100.0, 'y+': 60.0, 'z+': rng.uniform(0, 700, shape)}, rng.uniform(0, 500, shape)][seed % 3]
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=dm,
             dir_value=(rng.uniform(50, 90, shape) if dm is not None else None), neumann=neumann, robin_h=robin, Tinf=25.0,
             theta=float(rng.choice([0.5, 1.0])), dt=float(rng.choice([2.0, 60.0, 700.0])) * dx * dx / alpha, nsteps=2,
             births=None)
    got = run_cart_case(hip, c)['T_final']
    want = run_cart_case(orc, c)['T_final']
    assert rel_linf(got, want) <= TOL, (shape, kind, rel_linf(got, want))
    assert np.array_equal(got[~mask], c['T0'][~mask])


def test_no_fallback_promise_is_learnt_per_configuration(hip):
    """sparse bit 2 (include/adi_hip.h): after a sweep whose unit queue came back empty the host layer skips the queue reset
    and the GENERAL launch for the same (mask, packs, variant, shape); a configuration that does queue units never gets the
    bit, and a mask change starts over.  Results are those of the plain path in every case."""
    from oracle import adi_oracle as orc
    shape = (256, 16, 32)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    rng = np.random.default_rng(3)
    T0 = rng.uniform(20.0, 900.0, shape)

    def run(api, mask, nsteps, grid=None, packs=None):
        g = grid or api.Grid3D(*shape, dx, mask)
        mat = api.Material(7800.0, 490.0, 54.0); prm = api.Params(80.0 * dx * dx / alpha, 0.5)
        pk = packs or api.precompute_coeff_packs_unified(g, mat, robin_h=300.0)
        T = np.array(T0)
        step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
        for _ in range(nsteps):
            T = step(T, g, mat, prm, pk, Tinf=20.0)
        return T, g, pk

    solid = np.ones(shape, bool)
    got, g, pk = run(hip, solid, 3)
    want, _, _ = run(orc, solid, 3)
    assert rel_linf(got, want) <= TOL
    learnt = [v for p in pk for v in p._nofb.values()]
    assert learnt and all(v is True for v in learnt)               # the all-solid box queues nothing: every sweep carries the bit
    # a Dirichlet plane queues units: never promised, still right
    dm = np.zeros(shape, bool); dm[:, 0, :] = True
    res = []
    for api in (hip, orc):
        gg = api.Grid3D(*shape, dx, solid)
        mat = api.Material(7800.0, 490.0, 54.0); prm = api.Params(80.0 * dx * dx / alpha, 0.5)
        pp = api.precompute_coeff_packs_unified(gg, mat, robin_h=300.0, dir_mask=dm, dir_value=55.0)
        T = np.array(T0)
        step = getattr(api, 'adi_step_hip_coeff', None) or api.adi_step_numba_coeff
        for _ in range(3):
            T = step(T, gg, mat, prm, pp, Tinf=20.0)
        res.append(T)
        if api is hip:
            assert not all(v is True for p in pp for v in p._nofb.values())
    assert rel_linf(res[0], res[1]) <= TOL
    # the mask changes under the same Grid3D: new version, nothing promised until seen again
    holes = rng.random(shape) > 0.1
    g.mask = holes
    mat = hip.Material(7800.0, 490.0, 54.0)
    pk2 = hip.precompute_coeff_packs_unified(g, mat, robin_h=300.0)
    got2, _, _ = run(hip, holes, 3, grid=g, packs=pk2)
    want2, _, _ = run(orc, holes, 3)
    assert rel_linf(got2, want2) <= TOL
    assert not all(v is True for p in pk2 for v in p._nofb.values())


@pytest.mark.parametrize('n', [320, 384, 448, 640, 768, 896, 1280, 288, 352, 416, 480, 576, 960])
@pytest.mark.parametrize('ax', [0, 1, 2])
def test_exact_fit_row_counts(hip, n, ax):
    """lines whose length is 20, 24 or 28 times a power of two: the FAST kernels take 20 / 24 / 28 rows per lane (thread) so
    that the line fills the lanes of the interface solve exactly (adi_sweep_contig_x.hip, adi_sweep_strided_x.hip); the
    strided ones also 18 / 22 / 26 / 30 rows (288 ... 480 and 576 ... 960 rows; adi_sweep_strided_y.hip).  Solid
    blocks, voids, a curved solid and the general pack, against the oracle.  n = 1280 is NOT an exact-fit case: lines
    beyond 1024 rows take the thread-per-line kernel (k_sweep_generic) on every axis; it is here as that path's test."""
    from oracle import adi_oracle as orc
    alpha = 54.0 / (7800.0 * 490.0)
    for kind in ('solid', 'holes', 'ellipsoid', 'general'):
        rng = np.random.default_rng(n + ax)
        shape = [12, 16, 20]
        shape[ax] = n
        if ax != 2:
            shape[2] = 32
        shape = tuple(shape)
        mask = np.ones(shape, bool)
        if kind == 'holes':
            mask = rng.random(shape) > 0.03
        if kind == 'ellipsoid':
            g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
            mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
        dm = dv = neu = None
        if kind == 'general':
            dm = np.zeros(shape, bool); dm[:, 0, :] = True
            dv = 77.0
            neu = {'x+': 1e5, 'z-': rng.uniform(0, 1e5, shape)}
        dx = 1e-3
        c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1500.0, shape),
                 dir_mask=dm, dir_value=dv, neumann=neu, robin_h=350.0, Tinf=20.0, theta=0.5, dt=150.0 * dx * dx / alpha,
                 nsteps=2, births=None)
        err = rel_linf(run_cart_case(hip, c)['T_final'], run_cart_case(orc, c)['T_final'])
        assert err <= TOL, (n, ax, kind, err)


@pytest.mark.parametrize('n', [144, 176, 208, 240, 288, 352, 416, 480, 160, 224])
def test_fused_exact_fit_row_counts(hip, n):
    """the fused explicit + axis-0 kernel with 9 ... 15 rows per thread (adi_sweep_strided_fx.hip / _fy.hip): lines of every
    multiple of 16 rows up to 256 and of 32 rows up to 512 cut into exactly 16 / 32 segments.  Solid blocks (lean build),
    scalar faces with fluxes, voids, a curved solid (surface segments), per-voxel h -- against the oracle, and the fused
    kernel against the two-kernel form of the same stages"""
    from oracle import adi_oracle as orc
    from adi_thermal_fields_amd import _lib
    alpha = 54.0 / (7800.0 * 490.0)
    shape = (n, 6, 64)
    assert hip.recommended_dims(*shape) == shape                 # (an exact fit is not padded)
    for kind in ('solid', 'solid_q', 'holes', 'ellipsoid', 'array_h'):
        rng = np.random.default_rng(n + len(kind))
        mask = np.ones(shape, bool)
        if kind == 'holes':
            mask = rng.random(shape) > 0.03
        if kind == 'ellipsoid':
            g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
            mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
        neu = {'x+': 1e5, 'x-': -2e4, 'y+': 3e4} if kind == 'solid_q' else None
        rh = rng.uniform(0.0, 700.0, shape) if kind == 'array_h' else {'x-': 350.0, 'x+': 20.0, 'y-': 100.0, 'z+': 500.0}
        dx = 1e-3
        c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1500.0, shape),
                 dir_mask=None, dir_value=None, neumann=neu, robin_h=rh, Tinf=20.0, theta=0.5, dt=150.0 * dx * dx / alpha,
                 nsteps=2, births=None)
        err = rel_linf(run_cart_case(hip, c)['T_final'], run_cart_case(orc, c)['T_final'])
        assert err <= TOL, (n, kind, err)
        grid = hip.Grid3D(*shape, dx, mask)
        mat = hip.Material(7800.0, 490.0, 54.0); prm = hip.Params(c['dt'], 0.5)
        packs = hip.precompute_coeff_packs_unified(grid, mat, neumann=neu, robin_h=rh)
        assert hip.fused_supported(grid)
        U1 = hip.adi_explicit_sweep_axis0(c['T0'], grid, mat, prm, packs[0], Tinf=20.0)
        U2 = hip.adi_sweep_axis(0, hip.adi_explicit_rhs(c['T0'], grid, mat, prm), grid, mat, prm, packs[0], Tinf=20.0)
        assert rel_linf(U1, U2) <= 1e-13, (n, kind, rel_linf(U1, U2))
