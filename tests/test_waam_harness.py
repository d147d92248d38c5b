"""The WAAM layer-birth and single-track loops (adi_thermal_fields_amd.waam): host logic on CPU, and the
HIP backend against the oracle driven through the very same loop on the GPU."""
import numpy as np
import pytest

from helpers import rel_linf

STEEL = (7800.0, 490.0, 54.0)


def _setup(shape):
    from adi_thermal_fields_amd import waam
    mask = waam.synthetic_head_mask(*shape)
    layers = waam.plan_layers(mask, 2)
    dx = 1e-3
    times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
    return waam, mask, layers, dx, times


def test_layers_and_times_host_logic():
    waam, mask, layers, dx, times = _setup((24, 24, 30))
    covered = np.zeros(30, bool)
    for ks, ke in layers:
        assert ks <= ke and mask[:, :, ks].any() and mask[:, :, ke].any()
        covered[ks:ke + 1] = True
    assert np.array_equal(covered, mask.any(axis=(0, 1)))        # every non-empty plane is born exactly once
    assert all(b[0] == a[1] + 1 for a, b in zip(layers, layers[1:]))
    assert len(times) == len(layers) and all(t2 > t1 for t1, t2 in zip(times, times[1:]))
    with pytest.raises(RuntimeError):
        waam.plan_layers(np.zeros((3, 3, 3), bool), 2)


@pytest.mark.parametrize('seed', range(6))
def test_layers_and_times_match_the_reference_restatement(seed):
    """plan_layers / birth_times against the line-by-line restatement of waam_from_stl_v7_mm.py:436-476
    (oracle/waam_oracle.py) on masks with empty planes inside, at the ends and in runs: identical layer lists,
    bit-identical times"""
    from adi_thermal_fields_amd import waam
    from oracle import waam_oracle as ref
    rng = np.random.default_rng(seed)
    nz = int(rng.integers(5, 60))
    mask = rng.random((6, 7, nz)) < 0.3
    empty = rng.random(nz) < (0.1 + 0.15 * seed)                 # whole planes switched off
    mask[:, :, empty] = False
    if not mask.any():
        mask[0, 0, nz // 2] = True
    for n_per in (1, 2, 3, 7):
        layers = waam.plan_layers(mask, n_per)
        assert layers == ref.plan_layers(mask, n_per)
        assert all(isinstance(k, int) for lay in layers for k in lay)
        t_new = waam.birth_times(mask, layers, 7e-4, bead_width=3e-3, scan_speed=0.013, eta_fill=1.2)
        t_ref = ref.birth_times(mask, layers, 7e-4, bead_width=3e-3, scan_speed=0.013, eta_fill=1.2)
        assert t_new == t_ref


@pytest.mark.parametrize('seed', range(6))
def test_schedule_is_the_reference_event_loop(seed):
    """waam.layer_birth_schedule against the loop of waam_from_stl_v7_mm.py:515-550 written out action by action: same
    actions, same order, same intervals (bit for bit), with births that coincide with output times, output times before the
    first birth and after the last one, and simultaneous births"""
    from adi_thermal_fields_amd import waam
    rng = np.random.default_rng(seed)
    tb = list(np.cumsum(rng.uniform(0.0, 2.0, 12)))
    if seed % 2:
        tb[5] = tb[4]                                            # two layers born at the same time
    to = sorted(set([0.5 * tb[0], tb[3], tb[-1], tb[-1] + 3.0] + list(rng.uniform(0.0, tb[-1], 4))))
    want, nb, t_now = [], 0, 0.0
    for te in sorted(set(to + tb)):
        while nb < len(tb) and tb[nb] <= te + 1e-15:
            seg = max(0.0, tb[nb] - t_now)
            if seg > 1e-15:
                want.append(('advance', seg))
            t_now = tb[nb]
            want.append(('birth', nb))
            nb += 1
        seg = max(0.0, te - t_now)
        if seg > 1e-15:
            want.append(('advance', seg))
        t_now = te
        if any(abs(te - x) <= 1e-12 for x in to):
            want.append(('frame', te))
    got = list(waam.layer_birth_schedule(tb, to))
    assert got == want
    assert [a for w, a in got if w == 'birth'] == list(range(12))
    assert abs(sum(a for w, a in got if w == 'advance') - max(to)) <= 1e-9


def test_layer_birth_loop_on_oracle_runs():
    """the loop itself is backend-agnostic: run it on the CPU oracle (tiny grid)"""
    from oracle import adi_oracle as orc
    waam, mask, layers, dx, times = _setup((10, 10, 12))
    frames = []
    T, nsteps = waam.run_layer_birth(orc, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times,
                                     [0.0, times[-1]], on_frame=lambda t, T, m: frames.append((t, T.max(), m.sum())))
    assert nsteps >= len(layers) - 1 and len(frames) == 2
    assert np.all(T[~mask] == 20.0)                  # never-born cells untouched
    assert 20.0 < T[mask].max() <= 1000.0 + 1e-9


@pytest.mark.gpu
def test_layer_birth_hip_matches_oracle():
    """BASELINE.json configs[4] shape (voxel mask + layer birth, pack rebuild per birth) on a 24x24x30 shrink"""
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    waam, mask, layers, dx, times = _setup((24, 24, 30))
    outs = [0.0, 0.5 * times[-1], times[-1]]
    want, n1 = waam.run_layer_birth(orc, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
    fr = []
    got, n2 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs,
                                   on_frame=lambda t, T, m: fr.append(T))
    assert n1 == n2 and len(fr) == 3
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)
    assert np.array_equal(got[~mask], want[~mask])


@pytest.mark.gpu
def test_single_track_hip_matches_oracle():
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import waam
    shape = (14, 18, 10)
    plate = np.zeros(shape, bool); plate[:, :, :4] = True
    box = (5, 9, 4, 7, 12)
    args = (plate, box, 1e-3, STEEL, 25.0, 20.0, 1500.0, 0.5, 0.02, 0.05)
    want = waam.run_single_track(orc, *args)
    got = waam.run_single_track(hip, *args)
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)


def _single_track_case():
    """a plate of 5 planes along axis 2, a track of 12 columns along axis 1 on top of it whose box spans planes 5..19 of
    the SHARDED axis: with slabs of 8 planes the deposit lands on ranks 0, 1 and 2, with its ends one plane inside rank 0
    and rank 2 (so the neighbours' halo coupling bits change with every column)"""
    shape = (24, 18, 12)
    plate = np.zeros(shape, bool); plate[:, :, :5] = True
    box = (5, 20, 5, 8, 12)
    return shape, plate, box, (1e-3, STEEL, 25.0, 20.0, 1500.0, 0.5, 0.02, 0.05)


@pytest.mark.gpu
def test_single_track_on_slabs_matches_single_domain_and_oracle():
    """configs[4] "moving source" on slabs (single_track_on_plate.py:157-177): 3 in-process ranks on one GPU (HIP engine)
    against the single-domain HIP run of the same loop (<= 1e-11) and against the CPU oracle (<= 1e-10)"""
    import threading
    import torch
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab, waam
    shape, plate, box, rest = _single_track_case()
    dx, _, h, Tinf, T_track, theta, dt, t_step = rest
    one = waam.run_single_track(hip, plate, box, *rest)
    want = waam.run_single_track(orc, plate, box, *rest)
    for world, sizes in ((3, [8, 8, 8]), (4, [6, 6, 6, 6]), (2, [12, 12])):
        comms = dist_slab.LocalComm.make(world)
        parts, errs = [None] * world, []

        def work(rank):
            try:
                torch.cuda.set_device(0)
                i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
                parts[rank] = waam.run_single_track_slab(comms[rank], i0, i1, plate, box, dx, hip.Material(*STEEL), hip.Params,
                                                         h, Tinf, T_track, theta, dt, t_step)
            except Exception as e:
                errs.append(e)
                comms[rank].sh.barrier.abort()
        ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        if errs:
            raise errs[0]
        got = np.concatenate(parts, axis=0)
        assert rel_linf(got, one) <= 1e-11, (world, rel_linf(got, one))
        assert rel_linf(got, want) <= 1e-10, (world, rel_linf(got, want))
        assert np.array_equal(got[~plate & (got == Tinf)], want[~plate & (got == Tinf)])


def _track_worker(rank, world, port, sizes, q):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from adi_thermal_fields_amd import dist_slab, waam
        from cpu_engine import CpuEngine
        from oracle import adi_oracle as orc
        shape, plate, box, rest = _single_track_case()
        dx, _, h, Tinf, T_track, theta, dt, t_step = rest
        i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
        part = waam.run_single_track_slab(dist_slab.TorchDistComm(), i0, i1, plate, box, dx, orc.Material(*STEEL), orc.Params,
                                          h, Tinf, T_track, theta, dt, t_step, engine=CpuEngine())
        q.put((rank, part))
    finally:
        dist.destroy_process_group()


def test_single_track_on_slabs_gloo_matches_oracle():
    """the same loop on CPU ranks over gloo (reference engine): the host logic of run_single_track_slab -- which ranks a
    column lands on, the collective mask / pack rebuild, the sub-stepping -- against the single-domain oracle loop"""
    import socket
    import torch.multiprocessing as mp
    from oracle import adi_oracle as orc
    from adi_thermal_fields_amd import waam
    shape, plate, box, rest = _single_track_case()
    want = waam.run_single_track(orc, plate, box, *rest)
    world, sizes = 3, [8, 8, 8]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_track_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([parts[r] for r in range(world)], axis=0)
    assert rel_linf(got, want) <= 1e-12, rel_linf(got, want)


@pytest.mark.gpu
def test_layer_birth_on_slabs_matches_single_domain():
    """configs[4] structure: layer birth on a slab decomposition (3 in-process ranks on one GPU, HIP engine) against
    the single-domain HIP run of the same loop"""
    import threading
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab
    waam, mask, layers, dx, times = _setup((24, 24, 30))
    outs = [0.0, times[-1]]
    want, n1 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
    world, sizes = 3, [8, 8, 8]
    comms = dist_slab.LocalComm.make(world)
    parts, errs = [None] * world, []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
            parts[rank], _ = waam.run_layer_birth_slab(comms[rank], i0, i1, mask, dx, hip.Material(*STEEL), hip.Params,
                                                       40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
        except Exception as e:
            errs.append(e)
            comms[rank].sh.barrier.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    if errs:
        raise errs[0]
    got = np.concatenate(parts, axis=0)
    assert rel_linf(got, want) <= 1e-11, rel_linf(got, want)


def _slab_run(world, sizes, mask, dx, layers, times, outs, theta=0.5):
    import threading
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab, waam
    comms = dist_slab.LocalComm.make(world)
    parts, errs = [None] * world, []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
            parts[rank], _ = waam.run_layer_birth_slab(comms[rank], i0, i1, mask, dx, hip.Material(*STEEL), hip.Params,
                                                       40.0, 20.0, 1000.0, theta, 2000.0, layers, times, outs)
        except Exception as e:
            errs.append(e)
            comms[rank].sh.barrier.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    if errs:
        raise errs[0]
    return np.concatenate(parts, axis=0)


@pytest.mark.gpu
def test_config5_shrink_64x64x80_oracle_hip_and_4_slabs():
    """BASELINE.json configs[4] on its 64 x 64 x 80 shrink: synthetic head (the STL and trimesh
    are not available), layers of 2 planes, cfl 2000 sub-stepping, pack rebuild per birth, Robin h = 40 on every face:
    CPU oracle == one-domain HIP (<= 1e-10) == 4 slabs (<= 1e-11)"""
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    waam, mask, layers, dx, times = _setup((64, 64, 80))
    outs = [0.0, times[-1]]
    want, n1 = waam.run_layer_birth(orc, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
    got, n2 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
    assert n1 == n2 and len(layers) == 40
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)
    assert np.array_equal(got[~mask], want[~mask])
    slabs = _slab_run(4, [16, 16, 16, 16], mask, dx, layers, times, outs)
    assert rel_linf(slabs, got) <= 1e-11, rel_linf(slabs, got)


@pytest.mark.gpu
def test_config5_full_256x256x320_4_slabs_match_one_domain():
    """the full 256 x 256 x 320 synthetic head (21 M cells, 6 M in the mask), layers of 2 planes (160 births, pack
    rebuild per birth), cfl 2000: 4 slabs of 64 planes in one process on one GPU against the one-domain HIP run of the
    same loop.  theta = 1 here: with the driver's default theta = 0.5 the reference's scheme leaves the physical range on
    this mask at this size (reference defect D9; scripts/d9_probe.py -> profiles/r03_d9_probe.txt, CPU
    oracle alone: inside [20, 1000] through step 22, -1.3e5 / +1.2e4 at step 23, +-7.2e7 at step 26, +-8e16 at step 43,
    growing 2 - 10x per step) -- nothing to assert on a diverged field -- while theta = 1 makes every sweep a max-norm
    contraction."""
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import waam
    shape = (256, 256, 320)
    mask = waam.synthetic_head_mask(*shape)
    layers = waam.plan_layers(mask, 2)
    dx = 1e-3
    times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
    outs = [0.0, times[-1]]
    theta = 1.0
    want, n1 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, theta, 2000.0, layers, times, outs)
    assert len(layers) == 160 and n1 >= 159
    assert np.all(want[~mask] == 20.0) and 20.0 < want[mask].max() <= 1000.0 + 1e-9 and want.min() >= 20.0 - 1e-9
    got = _slab_run(4, [64, 64, 64, 64], mask, dx, layers, times, outs, theta)
    assert rel_linf(got, want) <= 1e-11, rel_linf(got, want)


@pytest.mark.gpu
def test_config5_full_size_theta_half_as_specified_before_divergence():
    """BASELINE.json configs[4] exactly as SURVEY.md 8(d) writes it -- 256 x 256 x 320 synthetic head, layers of 2 planes,
    cfl 2000, theta = 0.5, Robin h = 40, Ts = 1000 -- over the window in which the reference scheme is still bounded.
    At theta = 0.5 / cfl = 2000 the scheme leaves the physical range on this mask (reference defect D9, DESIGN.md
    section 6; scripts/d9_probe.py -> profiles/r03_d9_probe.txt: the pinned oracle stays inside [20, 1000] through step
    22, reaches -1.3e5 / +1.2e4 at step 23 and +-7.2e7 at step 26), so the full-size comparison with the CPU oracle
    covers the first 21 births = 20 steps: HIP == OpenMP oracle to 1e-10."""
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import waam
    shape = (256, 256, 320)
    mask = waam.synthetic_head_mask(*shape)
    layers = waam.plan_layers(mask, 2)
    dx = 1e-3
    times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
    nb = 21
    outs = [times[nb - 1]]

    class OmpOracle:                      # the oracle module with its OpenMP step (same arithmetic, bit-identical)
        Grid3D, Material, Params = orc.Grid3D, orc.Material, orc.Params
        precompute_coeff_packs_unified = staticmethod(orc.precompute_coeff_packs_unified)

        @staticmethod
        def adi_step_numba_coeff(T, grid, mat, params, packs, Tinf=0.0):
            return orc.adi_run(T, grid, mat, params, packs, Tinf=Tinf, nsteps=1, omp=True)
    want, n1 = waam.run_layer_birth(OmpOracle, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers[:nb], times[:nb], outs)
    got, n2 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers[:nb], times[:nb], outs)
    assert n1 == n2 == 20, (n1, n2)
    assert want.max() <= 1000.0 + 1e-6 and want.min() >= 20.0 - 1e-6      # still inside the bounded window
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)
    assert np.array_equal(got[~mask], want[~mask])


@pytest.mark.gpu
def test_layer_birth_long_segments_use_graph_and_match_oracle():
    """small cfl -> segments of dozens of sub-steps: the device backend replays them from a HIP graph
    (StagedStepper.run); same result as the oracle driven through the same loop"""
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    waam, mask, layers, dx, times = _setup((16, 16, 20))
    outs = [0.0, times[-1]]
    want, n1 = waam.run_layer_birth(orc, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 0.25, layers, times, outs)
    got, n2 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 0.25, layers, times, outs)
    assert n1 == n2 and n1 / len(layers) >= waam.GRAPH_MIN_NSUB
    assert rel_linf(got, want) <= 1e-10, rel_linf(got, want)


@pytest.mark.gpu
def test_birth_kernel_and_incremental_packs_equal_the_full_rebuild():
    """adi_birth_planes + the plane-range rebuild of flags and coefficient arrays (BirthPacks.update) against the
    reference's sequence on the host (newborn = full & ~active; T[newborn] = Ts; active |= full; full pack rebuild,
    waam_from_stl_v7_mm.py:487-495, :534): identical masks, fields, newborn counts, and BIT-identical flags and packs
    after every birth -- including births out of order and a layer born twice"""
    import torch
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    waam, mask, layers, dx, times = _setup((20, 18, 26))
    nx, ny, nz = mask.shape
    mat = hip.Material(*STEEL)
    robin = {f: 40.0 for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')}
    grid = hip.Grid3D(nx, ny, nz, dx, np.zeros_like(mask))
    L = grid.layout
    d_full = L.to_layout(mask, torch.uint8)
    d_act = L.empty(torch.uint8, zero=True)
    grid.set_mask_device(d_act, all_solid=False)
    bp = hip.BirthPacks(grid, mat, robin_h=robin)
    rng = np.random.default_rng(1)
    T0 = rng.uniform(20.0, 300.0, mask.shape)
    T = hip.to_device(T0)
    T_host, act_host = T0.copy(), np.zeros_like(mask)
    order = list(range(len(layers)))
    order[2], order[5] = order[5], order[2]
    order.insert(4, order[1])                                        # a layer that is born again: nothing is newborn
    ref_grid = hip.Grid3D(nx, ny, nz, dx, act_host)
    for li in order:
        ks, ke = layers[li]
        cnt = hip.birth_planes(T, d_act, d_full, grid, ks, ke + 1, 1000.0)
        grid.set_mask_device(d_act, max(0, ks - 1), min(nz, ke + 2), all_solid=False)
        packs = bp.update(ks - 1, ke + 2)
        born = np.zeros_like(mask); born[:, :, ks:ke + 1] = mask[:, :, ks:ke + 1]
        newborn = born & ~act_host
        T_host[newborn] = 1000.0
        act_host |= born
        assert int(cnt.item()) == int(newborn.sum())
        assert np.array_equal(grid.mask, act_host) and np.array_equal(T.get(), T_host)
        ref_grid.mask = act_host
        ref_packs = hip.precompute_coeff_packs_unified(ref_grid, mat, robin_h=robin)
        assert torch.equal(grid.d_flags, ref_grid.d_flags)
        for a in range(3):
            assert np.array_equal(packs[a].coeff, ref_packs[a].coeff), (li, a)
            assert np.array_equal(packs[a].qflux, ref_packs[a].qflux), (li, a)


@pytest.mark.gpu
def test_layer_birth_on_a_ragged_grid_one_domain_and_slabs():
    """a head of 72 x 66 x 50 voxels -- extents the kernels would not tile, so the single-domain grid runs on a padded
    physical box (Layout) and the slabs on padded planes (SlabStepper.plane_dims): births in place on the padded device
    mask, incremental flag / pack rebuilds, frames downloaded as the logical box; single domain against the oracle, 3
    slabs against the single domain"""
    from oracle import adi_oracle as orc
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab
    waam, mask, layers, dx, times = _setup((72, 66, 50))
    assert hip.Layout(*mask.shape).padded and dist_slab.HipEngine().plane_dims(66, 50) != (66, 50)
    layers, times = layers[:8], times[:8]                                  # the first 16 planes: enough births, a quick oracle
    outs = [0.0, times[-1]]
    want, n1 = waam.run_layer_birth(orc, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs)
    frames = []
    got, n2 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 0.5, 2000.0, layers, times, outs,
                                   on_frame=lambda t, T, m: frames.append((np.array(T), np.array(m))))
    assert n1 == n2 and rel_linf(got, want) <= 1e-10, rel_linf(got, want)
    assert all(T.shape == mask.shape and m.shape == mask.shape for T, m in frames)
    assert np.array_equal(got[~mask], want[~mask])
    slabs = _slab_run(3, [24, 24, 24], mask, dx, layers, times, outs)
    assert slabs.shape == mask.shape and rel_linf(slabs, got) <= 1e-11, rel_linf(slabs, got)
