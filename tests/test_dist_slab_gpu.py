"""GPU: the slab decomposition with the HIP engine, several ranks inside one process on one GPU
(dist_slab.LocalComm, one thread per rank), against the single-domain HIP step and the oracle."""
import threading

import numpy as np
import pytest

import cases
from helpers import rel_linf, run_cart_case

pytestmark = pytest.mark.gpu


def _run_slabs(c, world, sizes, nsteps, opts=None, modes=None):
    import torch
    from adi_thermal_fields_amd import dist_slab
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    comms = dist_slab.LocalComm.make(world)
    out = [None] * world
    errs = []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]

            def loc(a):
                return a if (a is None or np.isscalar(a)) else np.asarray(a)[i0:i1]
            neumann = None if c['neumann'] is None else {f: loc(v) for f, v in c['neumann'].items()}
            robin_h = {f: loc(v) for f, v in c['robin_h'].items()} if isinstance(c['robin_h'], dict) else loc(c['robin_h'])
            o = opts or {}
            engine = dist_slab.HipEngine()
            if o.get('no_pad'):                        # the caller's planes as they are (ragged lines on the GENERAL kernels)
                engine.plane_dims = lambda ny, nz: (ny, nz)
            st = dist_slab.SlabStepper(c['mask'][i0:i1], c['dx'], hip.Material(**c['mat']),
                                       hip.Params(c['dt'], c['theta']), c['Tinf'], dir_mask=loc(c['dir_mask']),
                                       dir_value=loc(c['dir_value']), neumann=neumann, robin_h=robin_h,
                                       comm=comms[rank], engine=engine)
            st._force_exact = bool(o.get('force_exact', False))
            st._allow_window = bool(o.get('allow_window', True))
            st._allow_fused = bool(o.get('allow_fused', True))
            st._allow_dots = bool(o.get('allow_dots', True))
            st._keep_r0 = bool(o.get('keep_r0', True))
            st._allow_deferred = bool(o.get('allow_deferred', True))
            st._allow_deferred_exact = bool(o.get('allow_deferred_exact', True))
            # (off unless a test asks for it: most cases here were written for the two-pass forms it would otherwise replace)
            st._allow_deferred_lines = bool(o.get('allow_deferred_lines', False))
            # (solids riddled with voids: every line is flagged -- the sparse pass at its fullest; the product's cost rule would
            # leave them to the window form unless a case asks for the rule with 'cost_rule')
            st._deferred_lines_cost_ratio = 1.0 if o.get('cost_rule') else float('inf')
            if 'dots_max' in o:
                st.DOTS_MAX_NONUNIFORM = o['dots_max']
            T = hip.to_device(np.ascontiguousarray(c['T0'][i0:i1]))
            for s in range(nsteps):
                T = st.step(T, prefetch_halo=bool(o.get('prefetch', False)) and s + 1 < nsteps)
            out[rank] = T.get()
            if modes is not None:
                modes.add(st.axis0_mode)
        except Exception as e:   # surface the failure and release the other ranks
            errs.append(e)
            comms[rank].sh.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    if errs:
        raise errs[0]
    return np.concatenate(out, axis=0)


# pass-A forms: dot products inside the explicit stage (default), fused strided pass A (with / without R0 kept for
# pass B), explicit stage + pass A as separate kernels
PASS_A = {'dots': dict(dots_max=1.0), 'auto': {}, 'fused': dict(allow_dots=False), 'fused_recompute': dict(allow_dots=False, keep_r0=False),
          'separate': dict(allow_dots=False, allow_fused=False)}


@pytest.mark.parametrize('world,case_name,form', [(2, 'holes_mixed', 'dots'), (2, 'holes_mixed', 'fused'),
                                                  (2, 'holes_mixed', 'separate'), (2, 'holes_mixed', 'auto'), (3, 'kat2', 'dots'),
                                                  (3, 'kat2', 'fused_recompute'), (4, 'long_line_70', 'dots'),
                                                  (4, 'long_line_70', 'fused'), (3, 'slab_chunks', 'dots'),
                                                  (3, 'slab_chunks', 'separate'), (2, 'dirichlet_only_gamma07', 'dots')])
def test_slabs_match_single_domain(world, case_name, form):
    from oracle import adi_oracle as orc
    from adi_thermal_fields_amd.dist_slab import split_planes
    c = cases.cart_case(case_name)
    nsteps = 2
    nx = c['shape'][0]
    base = nx // world
    sizes = [base] * world
    sizes[-1] += nx - base * world            # ragged / odd sizes -> generic condensation kernel
    got = _run_slabs(c, world, sizes, nsteps, PASS_A[form])
    c2 = dict(c); c2['nsteps'] = nsteps
    want = run_cart_case(orc, c2)['T_final']
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)


def test_slabs_even_sizes_fast_condense_512_lines():
    """whole-segment slabs (the in-register condensation kernel) on lines of 4 x 64 rows, vs one domain on HIP"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(5)
    shape = (256, 6, 40)
    mask = rng.random(shape) > 0.08
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann={'x-': 4e5},
             robin_h=350.0, Tinf=20.0, theta=0.5, dt=300.0 * dx * dx / alpha, nsteps=2, births=None)
    want = run_cart_case(hip, c)['T_final']
    for form, o in PASS_A.items():
        got = _run_slabs(c, 4, [64, 64, 64, 64], 2, o)
        assert rel_linf(got, want) <= 1e-12, (form, rel_linf(got, want))


@pytest.mark.parametrize('cfl,opts,mode', [(0.1, dict(prefetch=True), 'window'), (3.0, dict(prefetch=True), 'slab'),
                                           (0.1, dict(allow_window=False), 'slab'), (300.0, dict(prefetch=True), 'exact'),
                                           (0.1, dict(force_exact=True), 'exact'),
                                           (3.0, dict(prefetch=True, allow_dots=False), 'slab'),
                                           (300.0, dict(prefetch=True, allow_dots=False), 'exact'),
                                           (3.0, dict(prefetch=True, allow_dots=False, allow_fused=False), 'slab')])
def test_slabs_interface_forms_agree(cfl, opts, mode):
    """4 slabs of 64 planes: neighbour-only interface solve on 16-plane windows (cfl 0.1), on whole slabs (cfl 3),
    all-gather solve (cfl 300: no decay) -- each against the single-domain HIP step"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(7)
    shape = (256, 10, 40)
    mask = rng.random(shape) > 0.05
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann={'x-': 4e5},
             robin_h=350.0, Tinf=20.0, theta=0.5, dt=cfl * dx * dx / alpha, nsteps=3, births=None)
    modes = set()
    got = _run_slabs(c, 4, [64, 64, 64, 64], 3, opts, modes)
    assert modes == {mode}, modes
    want = run_cart_case(hip, c)['T_final']
    assert rel_linf(got, want) <= 1e-13, rel_linf(got, want)


@pytest.mark.parametrize('shape,sizes,cfl,opts,mode', [
    ((256, 16, 64), [64] * 4, 3.0, dict(prefetch=True), 'deferred'),                       # weights reach ~50 of 64 rows
    ((256, 16, 64), [64] * 4, 3.0, dict(prefetch=True, allow_fused=False), 'deferred'),    # explicit stage as its own kernel
    ((254, 16, 64), [64, 62, 64, 64], 3.0, dict(prefetch=True), 'deferred'),               # uneven slabs
    ((1024, 16, 32), [512, 512], 200.0, dict(prefetch=True), 'deferred'),                  # the bench's slab thickness and cfl
    ((1536, 16, 32), [512] * 3, 200.0, dict(prefetch=True), 'deferred'),                   # a middle rank: both corrections
    ((192, 24, 40), [64] * 3, 1.0, dict(prefetch=True, no_pad=True), 'deferred'),          # nz not a multiple of 16: GENERAL kernels
    ((128, 70, 16), [64, 64], 1.0, dict(prefetch=True, no_pad=True), 'deferred'),          # ragged axis-1 lines
    ((192, 24, 40), [64] * 3, 1.0, dict(prefetch=True), 'deferred'),                       # the same on padded planes (the default): the
    ((128, 70, 16), [64, 64], 1.0, dict(prefetch=True), 'deferred'),                       #   padding's lines get exact zero corrections
    ((256, 70, 90), [64] * 4, 3.0, dict(prefetch=True), 'deferred'),                       # both plane extents padded
    ((256, 16, 64), [64] * 4, 3.0, dict(prefetch=True, allow_deferred=False), 'slab'),     # the two-pass form on the same grid
    ((256, 16, 64), [64] * 4, 300.0, dict(prefetch=True), 'deferred_exact'),               # nothing decays across 64 rows
    ((256, 16, 64), [64] * 4, 300.0, dict(prefetch=True, allow_fused=False), 'deferred_exact'),
    ((192, 24, 40), [64] * 3, 2000.0, dict(prefetch=True, no_pad=True), 'deferred_exact'), # GENERAL axis-1 kernels, stiff
    ((192, 24, 40), [64] * 3, 2000.0, dict(prefetch=True), 'deferred_exact'),              # ... on padded planes
    ((128, 70, 90), [64, 64], 200.0, dict(prefetch=True), 'deferred_exact'),               # both ranks with an end row, padded planes
    ((128, 16, 64), [64, 64], 200.0, dict(prefetch=True), 'deferred_exact'),               # two ranks, both with a global end row
    ((512, 16, 32), [64] * 8, 200.0, dict(prefetch=True), 'deferred_exact'),               # the strong-scaling shape: 8 x 64 planes
    ((254, 16, 64), [64, 62, 64, 64], 300.0, dict(prefetch=True), 'deferred_exact'),       # uneven thin slabs
    ((256, 16, 64), [64] * 4, 300.0, dict(prefetch=True, allow_deferred_exact=False), 'exact'),   # the two-pass all-gather form
])
def test_slabs_deferred_form(shape, sizes, cfl, opts, mode):
    """all-solid slabs: the sharded-axis sweep as the single-domain (fused) kernel with zero boundary values + one plane to
    each neighbour + 2 x 2 interface systems + the rank-two correction added by the axis-1 sweep's loads (FAST and GENERAL
    strided kernels), against the single-domain HIP step: <= 1e-13"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(31)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=np.ones(shape, bool),
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann={'x-': 3e5, 'x+': 1e5, 'y+': 2e5},
             robin_h=rng.uniform(100.0, 600.0, shape), Tinf=20.0, theta=0.5, dt=cfl * dx * dx / alpha, nsteps=3,
             births=None)
    modes = set()
    got = _run_slabs(c, len(sizes), sizes, 3, opts, modes)
    assert modes == {mode}, modes
    want = run_cart_case(hip, c)['T_final']
    assert rel_linf(got, want) <= 1e-13, rel_linf(got, want)


def test_slabs_deferred_form_declines_non_uniform_lines():
    """one void cell, one Dirichlet cell: a sharded-axis line is no longer uniform, so no rank may take the deferred form"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(32)
    shape = (256, 16, 64)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    for kind in ('void', 'dirichlet'):
        mask = np.ones(shape, bool)
        dm = None
        if kind == 'void':
            mask[200, 3, 5] = False                          # on the last rank only
        else:
            dm = np.zeros(shape, bool); dm[10, 2, 7] = True  # on the first rank only
        c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1200.0, shape),
                 dir_mask=dm, dir_value=(None if dm is None else 50.0), neumann=None, robin_h=300.0, Tinf=20.0, theta=0.5,
                 dt=3.0 * dx * dx / alpha, nsteps=2, births=None)
        modes = set()
        got = _run_slabs(c, 4, [64] * 4, 2, dict(prefetch=True), modes)
        assert modes == {'slab'}, (kind, modes)
        want = run_cart_case(hip, c)['T_final']
        assert rel_linf(got, want) <= 1e-13, (kind, rel_linf(got, want))


@pytest.mark.parametrize('deferred,mode', [(False, 'window'), (True, 'deferred')])
def test_slabs_window_solid_512_lines(deferred, mode):
    """all-solid grid (the FAST uniform-interior kernels) on 2 slabs of 256 planes, 32-plane windows"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(8)
    shape = (512, 16, 64)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=np.ones(shape, bool),
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann=None,
             robin_h=500.0, Tinf=20.0, theta=0.5, dt=1.0 * dx * dx / alpha, nsteps=2, births=None)
    modes = set()
    got = _run_slabs(c, 2, [256, 256], 2, dict(prefetch=True, allow_deferred=deferred), modes)
    assert modes == {mode}, modes
    want = run_cart_case(hip, c)['T_final']
    assert rel_linf(got, want) <= 1e-13, rel_linf(got, want)


def test_slabs_solid_512_lines_dots_all_ranks():
    """all-solid grid, Robin + a Neumann face on the global axis-0 ends: every line is uniform, the first and last
    rank's lines carry the modified end row (Sherman-Morrison branch of k_dots_finish), the middle rank none"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(9)
    shape = (384, 8, 32)
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=np.ones(shape, bool),
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann={'x-': 3e5, 'x+': 1e5},
             robin_h=rng.uniform(100.0, 600.0, shape), Tinf=20.0, theta=0.5, dt=150.0 * dx * dx / alpha, nsteps=2,
             births=None)
    want = run_cart_case(hip, c)['T_final']
    for o in (dict(prefetch=True), dict(force_exact=True)):
        modes = set()
        got = _run_slabs(c, 3, [128, 128, 128], 2, o, modes)
        assert rel_linf(got, want) <= 1e-12, (o, modes, rel_linf(got, want))


@pytest.mark.parametrize('geom', ['ellipsoid', 'plates'])
@pytest.mark.parametrize('form', ['separate', 'fused', 'dots', 'auto'])
def test_slabs_512_planes_per_rank_curved_solid(form, geom):
    """2 slabs of 512 planes (the bench's slab thickness: 32-row wide tiling of the unfused pass A, 16-row tiles of the
    fused one) through an ellipsoid: surface segments (TAIL / HEAD) in every pass-A form, against the one-domain step;
    'plates': the ellipsoid cut into plates 1 .. 8 planes thick across x (ISLAND segments in the pass-A condense)"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    rng = np.random.default_rng(12)
    shape = (1024, 6, 32)
    g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
    mask = (g[0] / 0.47) ** 2 + (g[1] / 0.49) ** 2 + (g[2] / 0.48) ** 2 <= 1.0
    if geom == 'plates':
        i = np.arange(shape[0])
        keep = (i % 23) < (1 + (i // 23) % 8)
        keep[500:530] = True                               # one thick plate across the slab interface
        mask &= keep[:, None, None]
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask,
             T0=rng.uniform(20.0, 1200.0, shape), dir_mask=None, dir_value=None, neumann={'x-': 1e5},
             robin_h=250.0, Tinf=20.0, theta=0.5, dt=150.0 * dx * dx / alpha, nsteps=2, births=None)
    want = run_cart_case(hip, c)['T_final']
    got = _run_slabs(c, 2, [512, 512], 2, dict(PASS_A[form], prefetch=True))
    assert rel_linf(got, want) <= 1e-12, rel_linf(got, want)


def test_slab_step_over_real_rccl_self_loop():
    """the stepper of a middle rank over torch.distributed P2P (single-rank RCCL group, send/recv to self on the side
    stream, events, prefetched halos) == the same rank over in-process loopback copies, bit for bit, in every pass-A form"""
    import os
    import torch
    import torch.distributed as dist
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist_slab.rccl_env_defaults()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        rng = np.random.default_rng(21)
        shape = (128, 24, 64)
        g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
        mask = (g[1] / 0.45) ** 2 + (g[2] / 0.48) ** 2 <= 1.0            # a cylinder along the sharded axis
        dx = 1e-3
        alpha = 54.0 / (7800.0 * 490.0)
        T0 = rng.uniform(20.0, 900.0, shape)
        solid = np.ones(shape, bool)
        for cfl, opts in ((150.0, {}), (150.0, dict(allow_dots=False)), (0.05, {}), (300.0, dict(force_exact=True)),
                          (3.0, dict(solid=True)), (3.0, dict(solid=True, allow_fused=False))):
            outs = []
            for comm in (dist_slab.LoopbackComm(4, 1), dist_slab.SelfLoopDistComm(4, 1)):
                st = dist_slab.SlabStepper(solid if opts.get('solid') else mask, dx, hip.Material(7800.0, 490.0, 54.0),
                                           hip.Params(cfl * dx * dx / alpha, 0.5), 20.0, robin_h=300.0, comm=comm)
                st._allow_fused = opts.get('allow_fused', True)
                st._allow_dots = opts.get('allow_dots', True); st._force_exact = opts.get('force_exact', False)
                T = hip.to_device(T0)
                for s in range(4):
                    T = st.step(T, prefetch_halo=(s < 3))
                torch.cuda.synchronize()
                outs.append((T.get(), st.axis0_mode))
            assert outs[0][1] == outs[1][1] and (outs[0][1] == 'deferred') == bool(opts.get('solid'))
            assert np.array_equal(outs[0][0], outs[1][0]), (cfl, opts, outs[0][1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('shape,kind,world,opts', [((256, 70, 90), 'holes', 4, dict(prefetch=True)),
                                                   ((192, 66, 250), 'solid', 3, dict(prefetch=True)),
                                                   ((128, 100, 37), 'ellipsoid', 2, dict(prefetch=True, allow_fused=False)),
                                                   ((256, 70, 90), 'solid', 4, dict(prefetch=True, force_exact=True))])
def test_slabs_on_ragged_planes(shape, kind, world, opts):
    """(ny, nz) that the kernels would not tile: the stepper pads the planes of every internal array (plane_dims), the ranks
    exchange physical planes, the caller passes and gets the logical box; against the oracle on the logical grid"""
    from oracle import adi_oracle as orc
    from adi_thermal_fields_amd import dist_slab
    from adi_thermal_fields_amd.dist_slab import split_planes
    rng = np.random.default_rng(sum(shape) + world)
    if kind == 'solid':
        mask = np.ones(shape, bool)
    elif kind == 'holes':
        mask = rng.random(shape) > 0.1
    else:
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = (g[0] ** 2 + g[1] ** 2 + g[2] ** 2) <= 0.23
    assert dist_slab.HipEngine().plane_dims(*shape[1:]) != shape[1:]
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1500.0, shape),
             dir_mask=None, dir_value=None, neumann={'y+': rng.uniform(0.0, 1e5, shape), 'z-': 2e4},
             robin_h={'x-': 300.0, 'x+': 50.0, 'y-': rng.uniform(0.0, 500.0, shape), 'z+': 100.0}, Tinf=25.0, theta=0.5,
             dt=3.0 * dx * dx / alpha, nsteps=2, births=None)
    got = _run_slabs(c, world, split_planes(shape[0], world), 2, opts)
    want = run_cart_case(orc, c)['T_final']
    assert got.shape == shape and rel_linf(got, want) <= 1e-10, rel_linf(got, want)
    assert np.array_equal(got[~mask], c['T0'][~mask])


@pytest.mark.parametrize('shape,kind,sizes,cfl,opts', [
    ((256, 16, 64), 'holes', [64] * 4, 0.3, dict(prefetch=True)),                              # FAST kernels, voids in every line
    ((256, 16, 64), 'holes', [64] * 4, 0.3, dict(prefetch=True, allow_fused=False)),
    ((256, 48, 64), 'ellipsoid', [64] * 4, 0.3, dict(prefetch=True)),                          # a curved solid across all slabs
    ((254, 16, 64), 'dirichlet', [64, 62, 64, 64], 0.3, dict(prefetch=True)),                  # Dirichlet cells, uneven slabs
    ((1024, 16, 32), 'ellipsoid', [512, 512], 40.0, dict(prefetch=True)),                      # the bench's slab thickness, K = 183
    ((192, 70, 90), 'ellipsoid', [64] * 3, 0.3, dict(prefetch=True)),                          # padded planes
    ((192, 24, 40), 'holes', [64] * 3, 0.3, dict(prefetch=True, no_pad=True)),                 # GENERAL axis-1 kernels
    ((256, 64, 96), 'cavity', [128, 128], 3.0, dict(prefetch=True, cost_rule=True)),          # a part with a cavity: the cost rule takes it
])
def test_slabs_deferred_form_with_per_line_solutions(shape, kind, sizes, cfl, opts):
    """lines that are not uniform ('deferred_lines', ABI v17 / v18): per-line homogeneous solutions from two axis-0 sweeps per
    plan, the per-line 2 x 2 interface systems; lines that are uniform within reach of an interface take the scalar weights inside
    the axis-1 sweep (FAST and GENERAL strided kernels), the flagged ones -- crossing a void, the surface or a Dirichlet cell --
    their own weights from the sparse in-memory pass (adi_deferred_lines_apply) -- against the single-domain HIP step and the oracle"""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from oracle import adi_oracle as orc
    rng = np.random.default_rng(sum(shape))
    dx = 1e-3
    alpha = 54.0 / (7800.0 * 490.0)
    dm = dv = None
    if kind == 'holes':
        mask = rng.random(shape) > 0.06
    elif kind == 'ellipsoid':
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = (g[0] / 0.49) ** 2 + (g[1] / 0.46) ** 2 + (g[2] / 0.47) ** 2 <= 1.0
    elif kind == 'cavity':
        # a prism along the sharded axis with an off-centre cavity near the interface: most lines are uniform within reach
        g = np.meshgrid(*[(np.arange(s) + 0.5) / s - 0.5 for s in shape], indexing='ij')
        mask = ((g[1] / 0.47) ** 2 + (g[2] / 0.45) ** 2 <= 1.0) & \
            ~(((g[0] + 0.05) / 0.1) ** 2 + ((g[1] - 0.1) / 0.15) ** 2 + (g[2] / 0.2) ** 2 <= 1.0)
    else:
        mask = rng.random(shape) > 0.02
        dm = (rng.random(shape) < 0.01) & mask
        dv = rng.uniform(100.0, 400.0, shape)
    c = dict(shape=shape, dx=dx, mat=dict(rho=7800.0, cp=490.0, k=54.0), mask=mask, T0=rng.uniform(20.0, 1200.0, shape),
             dir_mask=dm, dir_value=dv, neumann={'x-': 3e5, 'y+': 2e5, 'z-': 1e4},
             robin_h={'x-': 300.0, 'x+': 80.0, 'y-': rng.uniform(100.0, 600.0, shape), 'z+': 500.0}, Tinf=20.0, theta=0.5,
             dt=cfl * dx * dx / alpha, nsteps=3, births=None)
    modes = set()
    got = _run_slabs(c, len(sizes), sizes, 3, dict(opts, allow_deferred_lines=True), modes)
    assert modes == {'deferred_lines'}, modes
    want = run_cart_case(hip, c)['T_final']
    assert rel_linf(got, want) <= 1e-12, rel_linf(got, want)
    assert rel_linf(got, run_cart_case(orc, c)['T_final']) <= 1e-10
    assert np.array_equal(got[~mask], c['T0'][~mask])
