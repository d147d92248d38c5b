"""CPU, world_size 2 and 3 over gloo: the slab decomposition (halo exchange, pass-A condensation,
all-gather, interface solve, pass-B injection) reproduces the single-domain oracle step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import cases

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case_name, sizes, nsteps, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from adi_thermal_fields_amd import dist_slab
        from cpu_engine import CpuEngine
        from oracle import adi_oracle as orc
        c = cases.cart_case(case_name)
        i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]

        def loc(a):
            return a if (a is None or np.isscalar(a)) else np.asarray(a)[i0:i1]
        neumann = None if c['neumann'] is None else {f: loc(v) for f, v in c['neumann'].items()}
        robin_h = {f: loc(v) for f, v in c['robin_h'].items()} if isinstance(c['robin_h'], dict) else loc(c['robin_h'])
        st = dist_slab.SlabStepper(c['mask'][i0:i1], c['dx'], orc.Material(**c['mat']),
                                   orc.Params(c['dt'], c['theta']), c['Tinf'], dir_mask=loc(c['dir_mask']),
                                   dir_value=loc(c['dir_value']), neumann=neumann, robin_h=robin_h,
                                   comm=dist_slab.TorchDistComm(), engine=CpuEngine())
        T = torch.from_numpy(np.ascontiguousarray(c['T0'][i0:i1]))
        for _ in range(nsteps):
            T = st.step(T)
        q.put((rank, T.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,case_name', [(2, 'holes_mixed'), (3, 'kat2'), (2, 'dirichlet_only_gamma07'),
                                             (2, 'slab_chunks')])
def test_slab_decomposition_matches_single_domain(world, case_name):
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import adi_oracle as orc
    from helpers import run_cart_case, rel_linf
    c = cases.cart_case(case_name)
    nsteps = 2
    nx = c['shape'][0]
    base = nx // world
    sizes = [base] * world
    sizes[-1] += nx - base * world          # uneven, odd slab sizes on purpose
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case_name, sizes, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([parts[r] for r in range(world)], axis=0)
    c2 = dict(c); c2['nsteps'] = nsteps
    want = run_cart_case(orc, c2)['T_final']
    assert rel_linf(got, want) <= 1e-12, rel_linf(got, want)


def test_split_planes_even():
    from adi_thermal_fields_amd.dist_slab import split_planes
    assert split_planes(512, 8) == [64] * 8
    for nx, w in ((70, 3), (513, 4), (12, 5)):
        s = split_planes(nx, w)
        assert sum(s) == nx and all(v > 0 for v in s) and all(v % 2 == 0 for v in s[:-1])
