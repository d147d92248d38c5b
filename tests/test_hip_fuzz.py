"""GPU: seeded random sweep over shapes, masks and boundary mixes, HIP against the CPU oracle (<= 1e-10 relative L-inf,
BASELINE.json's bar).  The fixed cases of test_hip_parity.py were each written for a feature; this file is the net
under them: line lengths on both sides of every tiling switch (8 / 16 / 32 rows per thread, 8- and 16-line tiles, whole
and ragged tiles, padded segment counts), dense hand-built packs and sparse device-built packs, every pack variant
(Dirichlet and / or Neumann present), solid boxes, random holes, curved solids and thin walls, theta in {0.5, 1}.
One-off soaks with the same generators (scripts/fuzz_soak.py): round 2's final build 3 400 Cartesian and 1 900 cylindrical cases,
no failure, worst relative L-inf 7.2e-13 / 9.9e-13; round 3 (face constants, reciprocal-free surface runs, 16 rows from 160-row
lines, 128-thread FAST workgroups, in-place cylindrical sweeps, padded extents): see profiles/r03_fuzz_soak*.txt."""
import numpy as np
import pytest

from helpers import rel_linf, run_cart_case

pytestmark = pytest.mark.gpu
STEEL = dict(rho=7800.0, cp=490.0, k=54.0)
ALPHA = STEEL['k'] / (STEEL['rho'] * STEEL['cp'])

# (long axis length, position of the long axis): the other two extents are drawn small so the oracle stays fast
LONG = [64, 72, 96, 128, 136, 160, 192, 256, 264, 320, 384, 512, 520, 640,
        100, 130, 250, 257, 300, 402]       # ragged lengths: these run on padded extents (Layout, adi_recommended_dims)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n_long = int(rng.choice(LONG))
    ax = int(rng.integers(0, 3))
    small = lambda: int(rng.choice([3, 4, 6, 8, 16, 17, 24, 32, 33, 48]))
    shape = [small(), small(), small()]
    shape[ax] = n_long
    if rng.random() < 0.5:                     # a second longish axis: tiles of the other strided sweep
        shape[(ax + 1) % 3] = int(rng.choice([40, 64, 80, 70, 100]))
    while shape[0] * shape[1] * shape[2] > 400000:
        i = int(np.argmin([s if j != ax else 1 << 30 for j, s in enumerate(shape)]))
        shape[(ax + 1) % 3] = max(3, shape[(ax + 1) % 3] // 2); shape[(ax + 2) % 3] = max(3, shape[(ax + 2) % 3] // 2)
    shape = tuple(shape)
    kind = rng.choice(['solid', 'holes', 'ellipsoid', 'walls', 'sparse_holes'])
    if kind == 'solid':
        mask = np.ones(shape, bool)
    elif kind == 'holes':
        mask = rng.random(shape) > 0.2
    elif kind == 'sparse_holes':
        mask = rng.random(shape) > 0.01
    elif kind == 'ellipsoid':
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
        mask &= ~((g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.02)
    else:
        mask = np.zeros(shape, bool)
        w = int(rng.integers(1, 6))
        sl = [slice(None)] * 3
        sl[ax] = slice(shape[ax] // 3, shape[ax] // 3 + w)
        mask[tuple(sl)] = True
        sl[ax] = slice(2 * shape[ax] // 3, 2 * shape[ax] // 3 + 2 * w + 9)
        mask[tuple(sl)] = True
    dx = float(rng.choice([2.5e-4, 1e-3]))
    cfl = float(rng.choice([0.7, 20.0, 200.0, 3000.0]))
    bc = rng.choice(['lean', 'dir', 'neu', 'general', 'array_h', 'scalar_faces'])
    dir_mask = dir_value = neumann = None
    robin_h = float(rng.uniform(50.0, 900.0))
    if bc in ('dir', 'general'):
        dir_mask = np.zeros(shape, bool)
        sl = [slice(None)] * 3
        sl[int(rng.integers(0, 3))] = 0 if rng.random() < 0.5 else -1
        dir_mask[tuple(sl)] = True
        dir_mask &= mask
        dir_value = rng.uniform(20.0, 1200.0, shape) if rng.random() < 0.5 else 333.0
    if bc in ('neu', 'general'):
        faces = ['x-', 'x+', 'y-', 'y+', 'z-', 'z+']
        neumann = {str(rng.choice(faces)): float(rng.uniform(-2e5, 2e6)), str(rng.choice(faces)): rng.uniform(0, 1e5, shape)}
    if bc == 'array_h':
        robin_h = {'x-': rng.uniform(0, 800.0, shape), 'y+': 300.0, 'z-': rng.uniform(0, 100.0, shape), 'z+': 40.0}
    if bc == 'scalar_faces':               # a different scalar on every face, scalar fluxes: the sweeps get face constants with q
        robin_h = {f: float(rng.uniform(0.0, 900.0)) for f in ['x-', 'x+', 'y-', 'y+', 'z-', 'z+'] if rng.random() < 0.8}
        neumann = {f: float(rng.uniform(-2e5, 2e6)) for f in ['x-', 'x+', 'y-', 'y+', 'z-', 'z+'] if rng.random() < 0.4}
        if rng.random() < 0.15:
            cfl = 1e-13                    # a vanishing time step: GENERAL kernels only (adi_core.hpp, kMixedMinTg)
    return dict(shape=shape, dx=dx, mat=dict(STEEL), mask=mask, T0=rng.uniform(20.0, 1500.0, shape), dir_mask=dir_mask,
                dir_value=dir_value, neumann=neumann, robin_h=robin_h, Tinf=float(rng.uniform(0.0, 40.0)),
                theta=float(rng.choice([0.5, 1.0])), dt=cfl * dx * dx / ALPHA, nsteps=2, births=None), (kind, bc, cfl)


@pytest.mark.parametrize('seed', range(48))
def test_random_case_vs_oracle(seed):
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    c, tag = _case(seed)
    got = run_cart_case(hip, c)
    want = run_cart_case(orc, c)
    for key in ('T_step1', 'T_final'):
        err = rel_linf(got[key], want[key])
        assert err <= 1e-10, (seed, c['shape'], tag, key, err)
    assert np.array_equal(got['T_final'][~c['mask']], c['T0'][~c['mask']])          # off-mask cells are never touched


@pytest.mark.parametrize('seed', range(8))
def test_random_case_device_resident_and_staged(seed):
    """the same cases through the device-resident loop (StagedStepper.step / run with and without the HIP graph): the same
    kernels on the same data, so bit-identical to the NumPy-in / NumPy-out calls"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    c, tag = _case(100 + seed)
    want = run_cart_case(hip, c)['T_final']
    grid = hip.Grid3D(*c['shape'], c['dx'], c['mask'])
    mat = hip.Material(**c['mat']); prm = hip.Params(c['dt'], c['theta'])
    packs = hip.precompute_coeff_packs_unified(grid, mat, dir_mask=c['dir_mask'], dir_value=c['dir_value'],
                                               neumann=c['neumann'], robin_h=c['robin_h'])
    st = hip.StagedStepper(grid, mat, prm, packs, c['Tinf'])
    T = hip.to_device(c['T0'])
    for _ in range(c['nsteps']):
        T = st.step(T)
    assert np.array_equal(T.get(), want), (seed, c['shape'], tag)
    assert np.array_equal(st.run(hip.to_device(c['T0']), c['nsteps'], graph=False).get(), want)
    assert np.array_equal(st.run(hip.to_device(c['T0']), c['nsteps'], graph=True).get(), want)


# ---- cylindrical BE step -------------------------------------------------------------------------------------------------
# shapes on both sides of every condition of the FAST kernels (adi_cyl.hip: r -- the (phi, z) plane a multiple of 64 lines;
# phi -- nz a multiple of 32, nphi a multiple of the segment length; z -- no Dirichlet closure, nz a multiple of 16 with a
# power-of-two segment count), all nine z closures, source term, masked step, annular grids
def _cyl_case(seed):
    rng = np.random.default_rng(5000 + seed)
    nr = int(rng.choice([1, 2, 5, 8, 16, 24, 64, 128]))
    nphi = int(rng.choice([1, 2, 3, 8, 16, 36, 64, 128, 256]))
    nz = int(rng.choice([3, 12, 16, 32, 40, 64, 128, 130, 256, 512]))
    while nr * nphi * nz > 300000:
        if nphi >= nz and nphi > 8:
            nphi //= 2
        elif nz > 16:
            nz //= 2
        else:
            nr = max(1, nr // 2)
    shape = (nr, nphi, nz)
    dr, dz = float(rng.choice([2.5e-4, 8e-4])), float(rng.choice([2.5e-4, 1.1e-3]))
    kinds = ['neumann0', 'dirichlet', 'robin']
    zbc = dict(kind_bot=str(rng.choice(kinds)), kind_top=str(rng.choice(kinds)), h_bot=float(rng.uniform(0, 300)),
               h_top=float(rng.uniform(0, 800)), T_inf_bot=25.0, T_inf_top=15.0, T_bot=300.0, T_top=80.0)
    c = dict(shape=shape, dr=dr, dz=dz, dphi=2.0 * np.pi / nphi, R=nr * dr, mat=dict(STEEL), T0=rng.uniform(20.0, 1200.0, shape),
             robin_r=(float(rng.choice([0.0, 400.0])), 20.0), zbc=zbc, dt=float(rng.choice([0.02, 0.3, 5.0])), nsteps=2, S=None,
             active=None, R_in=float(rng.choice([0.0, 0.0, 0.03])))
    mode = rng.choice(['plain', 'source', 'masked'])
    if mode == 'source':
        c['S'] = rng.uniform(0.0, 3e8, shape)
    elif mode == 'masked':
        c['active'] = rng.random(shape) > 0.3
        c['robin_inner'] = (10.0, 45.0); c['robin_void'] = (5.0, 27.0)
    return c, str(mode)


def _run_cyl(api, c, to_state=lambda x: x, to_host=lambda x: np.asarray(x)):
    nr, nphi, nz = c['shape']
    grid = api.GridCyl(nr, nphi, nz, c['dr'], c['dphi'], c['dz'], c['R_in'] + c['R'], R_in=c['R_in'])
    mat = api.Material(**c['mat']); prm = api.Params(c['dt'], 1.0, "be")
    rr = api.RobinR(*c['robin_r']); zbc = api.ZBC(**c['zbc'])
    T = to_state(np.array(c['T0']))
    for _ in range(c['nsteps']):
        if c['active'] is not None:
            T = api.adi_step_masked(T, grid, mat, prm, rr, zbc, c['active'], robin_inner=api.RobinR(*c['robin_inner']),
                                    robin_void=api.RobinR(*c['robin_void']))
        else:
            T = api.adi_step(T, grid, mat, prm, rr, zbc, S=c['S'])
    return to_host(T)


@pytest.mark.parametrize('seed', range(40))
def test_random_cyl_case_vs_oracle(seed):
    import adi_thermal_fields_amd.adi3d_hip_cyl as hipcyl
    from oracle import cyl_oracle as cyl
    c, mode = _cyl_case(seed)
    got = _run_cyl(hipcyl, c)
    want = _run_cyl(cyl, c)
    assert rel_linf(got, want) <= 1e-10, (seed, c['shape'], mode, c['zbc']['kind_bot'], c['zbc']['kind_top'], rel_linf(got, want))
    dev = _run_cyl(hipcyl, c, to_state=hipcyl.to_device, to_host=lambda d: d.get())
    assert np.array_equal(dev, got)                      # device-resident fields: the same kernels, the same bits
