"""GPU: BASELINE.json's full-size configurations.
config 2 (256^3 Robin, 20 steps) and the 512^3 headline workload (the very stepper bench.py times: 16 rows per lane,
padded 2 MiB planes, fused explicit + axis-0 kernel, no-fallback promise) are checked against the CPU oracle directly
(OpenMP build of the same arithmetic, bit-identical to the serial oracle: about 1.5 s per 512^3 step on the GPU box's
host cores), 512^3 also with a curved solid and every array of the general pack live; on top of that 512^3 is checked
through size-independent properties of the scheme (ambient fixed point, linearity of the step in (T, Tinf), mirror
symmetry, maximum principle)."""
import numpy as np
import pytest

from helpers import rel_linf

pytestmark = pytest.mark.gpu
STEEL = dict(rho=7800.0, cp=490.0, k=54.0)


def _setup(hip, n, cfl=200.0, dx=5e-4):
    grid = hip.Grid3D(n, n, n, dx, np.ones((n, n, n), bool))
    mat = hip.Material(**STEEL)
    alpha = mat.k / (mat.rho * mat.cp)
    prm = hip.Params(cfl * dx * dx / alpha, 0.5)
    packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
    return grid, mat, prm, packs


def test_config2_256_robin_20_steps_vs_oracle():
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    n = 256
    grid, mat, prm, packs = _setup(hip, n)
    T0 = np.random.default_rng(0).uniform(20.0, 1000.0, (n, n, n))
    T = hip.to_device(T0)
    for _ in range(20):
        T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
    og = orc.Grid3D(n, n, n, grid.dx, np.ones((n, n, n), bool))
    om = orc.Material(**STEEL); op = orc.Params(prm.dt, prm.theta)
    opacks = orc.precompute_coeff_packs_unified(og, om, robin_h=500.0)
    want = orc.adi_run(T0, og, om, op, opacks, Tinf=20.0, nsteps=20, omp=True)
    got = T.get()
    err = rel_linf(got, want)
    assert err <= 1e-10, err          # BASELINE.json: <= 1e-10 relative L-inf


def test_config2_256_disk_dirichlet_neumann_general_pack_vs_oracle():
    """SURVEY.md 8(d) config 2, second half: 256^3, disk-extruded mask, Dirichlet top plane, Neumann flux on z-, Robin
    on the lateral faces -- every array of the general pack (42 B/cell data model) is live, the mask has a curved
    surface along axes 0 and 1, and the contiguous lines end in a Dirichlet cell.  20 steps, cfl 200, theta 0.5, against
    the OpenMP build of the oracle: <= 1e-10 relative L-inf."""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    n = 256
    dx = 5e-4
    X = (np.arange(n) + 0.5 - n / 2) * dx
    disk = np.sqrt(X[:, None] ** 2 + X[None, :] ** 2) <= 0.45 * n * dx
    mask = np.ascontiguousarray(np.repeat(disk[:, :, None], n, axis=2))
    dm = np.zeros((n, n, n), bool); dm[:, :, -1] = mask[:, :, -1]
    kw = dict(dir_mask=dm, dir_value=20.0, neumann={'z-': 2e6},
              robin_h={'x-': 500.0, 'x+': 500.0, 'y-': 500.0, 'y+': 500.0})
    T0 = np.random.default_rng(0).uniform(20.0, 1000.0, (n, n, n))
    alpha = STEEL['k'] / (STEEL['rho'] * STEEL['cp'])
    dt = 200.0 * dx * dx / alpha
    res = []
    for api in (hip, orc):
        grid = api.Grid3D(n, n, n, dx, mask)
        mat = api.Material(**STEEL); prm = api.Params(dt, 0.5)
        packs = api.precompute_coeff_packs_unified(grid, mat, **kw)
        if api is hip:
            assert packs[2].variant == 0 and packs[0].has_q is True      # the general-pack kernels serve this run
            T = hip.to_device(T0)
            for _ in range(20):
                T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
            res.append(T.get())
        else:
            res.append(orc.adi_run(T0, grid, mat, prm, packs, Tinf=20.0, nsteps=20, omp=True))
    got, want = res
    assert np.array_equal(got[~mask], T0[~mask])                          # off-mask cells are never touched
    assert np.all(got[dm] == 20.0)                                        # Dirichlet cells hold their value
    err = rel_linf(got, want)
    assert err <= 1e-10, err


def test_512_bench_workload_3_steps_vs_oracle():
    """The timed workload of bench.py itself (adi3d_numba_coeff.py:290-302 at 512^3: all-solid box, Robin h = 500 on all
    faces, cfl 200, theta 0.5) through the StagedStepper bench.py uses -- FAST fused explicit + axis-0 kernel with 16 rows
    per thread on 2 MiB-pitch padded planes, the two FAST sweeps, and the no-fallback promise learnt on the third step --
    3 steps WITHOUT the promise and 3 steps WITH it from the same T0, both against the OpenMP oracle: <= 1e-10."""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    n = 512
    grid, mat, prm, packs = _setup(hip, n)
    assert grid.layout.sx > n * n                      # padded planes
    T0 = np.random.default_rng(1).uniform(20.0, 1000.0, (n, n, n))
    st = hip.StagedStepper(grid, mat, prm, packs, 20.0)
    assert st.fused and st.stage_names[0] == 'explicit+sweep_axis0'
    T = hip.to_device(T0)
    for _ in range(3):
        T = st.step(T)
    first = T.get()
    T = hip.to_device(T0)
    for _ in range(2):                                 # five steps of the configuration have run: the promise is learnt
        T = st.step(T)
    nofb = [v for p in packs for v in getattr(p, '_nofb', {}).values()]
    assert len(nofb) == 3 and all(v is True for v in nofb), nofb
    T = hip.to_device(T0)
    for _ in range(3):
        T = st.step(T)
    second = T.get()
    del T
    og = orc.Grid3D(n, n, n, grid.dx, np.ones((n, n, n), bool))
    om = orc.Material(**STEEL); op = orc.Params(prm.dt, prm.theta)
    opacks = orc.precompute_coeff_packs_unified(og, om, robin_h=500.0, _share=True)
    want = orc.adi_run(T0, og, om, op, opacks, Tinf=20.0, nsteps=3, omp=True)
    assert rel_linf(first, want) <= 1e-10, rel_linf(first, want)
    assert np.array_equal(first, second)               # the promise changes which launches run, not one bit of the result


def test_512_ellipsoid_general_pack_2_steps_vs_oracle():
    """512^3 box holding a curved solid (ellipsoid, semi-axes 0.47 / 0.49 / 0.48 of the box) with every array of the
    general pack live: Dirichlet cells on one plane through the solid, Neumann flux on the exposed z- faces, Robin on the
    others.  Surface segments (TAIL / HEAD) in all three FAST kernels, Dirichlet tiles through the GENERAL kernels,
    2 steps against the OpenMP oracle: <= 1e-10; off-mask cells untouched, Dirichlet cells at their value."""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    n = 512
    dx = 5e-4
    c = [((np.arange(n) + 0.5) / n - 0.5) / a for a in (0.47, 0.49, 0.48)]
    mask = (c[0][:, None, None] ** 2 + c[1][None, :, None] ** 2 + c[2][None, None, :] ** 2) <= 1.0
    dm = np.zeros((n, n, n), bool); dm[:, :, 96] = mask[:, :, 96]
    kw = dict(dir_mask=dm, dir_value=20.0, neumann={'z-': 2e6},
              robin_h={'x-': 500.0, 'x+': 500.0, 'y-': 500.0, 'y+': 500.0, 'z+': 300.0})
    T0 = np.random.default_rng(2).uniform(20.0, 1000.0, (n, n, n))
    alpha = STEEL['k'] / (STEEL['rho'] * STEEL['cp'])
    dt = 200.0 * dx * dx / alpha
    grid = hip.Grid3D(n, n, n, dx, mask)
    mat = hip.Material(**STEEL); prm = hip.Params(dt, 0.5)
    packs = hip.precompute_coeff_packs_unified(grid, mat, **kw)
    assert packs[2].variant == 0 and packs[0].has_q is True
    T = hip.to_device(T0)
    for _ in range(2):
        T = hip.adi_step_hip_coeff(T, grid, mat, prm, packs, Tinf=20.0)
    got = T.get()
    del T, packs, grid
    og = orc.Grid3D(n, n, n, dx, mask)
    om = orc.Material(**STEEL); op = orc.Params(dt, 0.5)
    opacks = orc.precompute_coeff_packs_unified(og, om, _share=True, **kw)
    want = orc.adi_run(T0, og, om, op, opacks, Tinf=20.0, nsteps=2, omp=True)
    assert np.array_equal(got[~mask], T0[~mask])
    assert np.all(got[dm] == 20.0)
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)


def test_512_properties():
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    n = 512
    grid, mat, prm, packs = _setup(hip, n)
    dev = torch.device('cuda')
    L = grid.layout

    def step(t, Tinf):
        return hip.adi_step_hip_coeff(hip.DeviceField(t), grid, mat, prm, packs, Tinf=Tinf).t

    # 1. the ambient temperature is a fixed point of the Robin problem
    amb = L.empty(); amb.fill_(20.0)
    out = step(amb, 20.0)
    assert float((out - 20.0).abs().max()) <= 1e-10
    # 2. linearity: step(a*T1 + b*T2; a*I1 + b*I2) = a*step(T1; I1) + b*step(T2; I2)
    g = torch.Generator(device=dev); g.manual_seed(3)
    T1 = L.empty(); T1.copy_(torch.rand((n, n, n), dtype=torch.float64, device=dev, generator=g) * 900 + 20)
    T2 = L.empty(); T2.copy_(torch.rand((n, n, n), dtype=torch.float64, device=dev, generator=g) * 900 + 20)
    a, b = 0.75, -1.5
    lhs = step(a * T1 + b * T2, a * 20.0 + b * 35.0)
    rhs = a * step(T1, 20.0) + b * step(T2, 35.0)
    scale = float(rhs.abs().max())
    assert float((lhs - rhs).abs().max()) / scale <= 1e-11
    del lhs, rhs
    # 3. mirror symmetry along every axis (all-solid cube, the same h on all faces)
    for ax in range(3):
        f1 = step(torch.flip(T1, dims=(ax,)), 20.0)
        f2 = torch.flip(step(T1, 20.0), dims=(ax,))
        assert float((f1 - f2).abs().max()) / scale <= 1e-11, ax
        del f1, f2
    # 4. maximum principle: the step cannot leave the range spanned by the field and the ambient
    o = step(T1, 20.0)
    assert float(o.max()) <= float(T1.max()) + 1e-9 and float(o.min()) >= 20.0 - 1e-9


def _config4(nphi):
    """BASELINE.json configs[3] / SURVEY.md 8(d) config 4: cylindrical 128 x 256 x 512, dr = dz = 2.5e-4, BE, dt = 0.05,
    T0 = 20 with the top 16 z-planes at 1000 (birth-like), Robin wall h = 400, z: neumann0 / robin h = 500"""
    nr, nz = 128, 512
    T0 = np.full((nr, nphi, nz), 20.0)
    T0[:, :, -16:] = 1000.0
    return dict(shape=(nr, nphi, nz), dr=2.5e-4, dz=2.5e-4, dphi=2.0 * np.pi / 256, R=0.032,
                mat=dict(STEEL), T0=T0, robin_r=(400.0, 20.0),
                zbc=dict(kind_bot='neumann0', kind_top='robin', h_top=500.0, T_inf_top=20.0), dt=0.05, S=None,
                active=None)


def test_config4_cylindrical_128x256x512():
    """(1) 3 steps on the full grid against the NumPy oracle directly (rel L-inf <= 1e-10);
    (2) all 50 steps on the full grid against the oracle run on ONE phi column: T0 and the BCs do not depend on phi,
    the phi solve of a phi-constant field returns it, so the 3-D solution is that column replicated -- a
    size-independent property that lets the 16.7M-cell, 50-step run be checked in seconds (the oracle takes 16 s per
    full-size step)."""
    import adi_thermal_fields_amd.adi3d_hip_cyl as hipcyl
    from oracle import cyl_oracle as cyl
    from helpers import run_cyl_case
    c = _config4(256)
    c3 = dict(c, nsteps=3)
    got3 = run_cyl_case(hipcyl, c3)['T_final']
    want3 = run_cyl_case(cyl, c3)['T_final']
    assert rel_linf(got3, want3) <= 1e-10, rel_linf(got3, want3)
    del want3
    got = run_cyl_case(hipcyl, dict(c, nsteps=50))['T_final']
    col = run_cyl_case(cyl, dict(_config4(1), nsteps=50))['T_final']          # (128, 1, 512)
    assert rel_linf(got, np.broadcast_to(col, got.shape)) <= 1e-10, rel_linf(got, np.broadcast_to(col, got.shape))
    assert float(np.abs(got - got[:, :1, :]).max()) <= 1e-9                     # still phi-independent
    assert 20.0 - 1e-9 <= got.min() and got.max() <= 1000.0 + 1e-9              # maximum principle
