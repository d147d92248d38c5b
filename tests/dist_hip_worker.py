"""One rank of a real multi-PROCESS run of the slab decomposition on the HIP engine (launched by tests/test_dist_nccl_spawn.py,
tests/test_dist_hip_processes.py or by hand:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
           tests/dist_hip_worker.py [--transport nccl|gloo-staged]).
Every rank builds the same seeded global case, steps its slab with SlabStepper + HipEngine, the slabs are gathered on rank 0
and compared with the one-domain HIP step of the whole grid computed there: <= 1e-12 relative L-inf in every interface form.

  --transport nccl          one process per GPU over torch.distributed 'nccl' (= RCCL); needs W GPUs
  --transport gloo-staged   every rank on cuda:0, payloads staged through pinned host memory and sent by gloo
                            (dist_slab.HostStagedDistComm, a test transport): separate processes, HIP contexts and
                            allocators on the one GPU of a test box, where RCCL refuses a second rank per device

Exit code 0 = all cases agree; the process group is created before any other GPU work, as RCCL wants it."""
import argparse
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))

# (name, planes per rank or None = 64, cfl, stepper options, steps, the form every rank must have chosen or None)
CASES = [
    ('voids, gentle step',            None, 0.05,  {}, 3, 'deferred_lines'),
    ('voids, per-line form off',      None, 0.05,  dict(allow_deferred_lines=False), 3, 'window'),
    ('voids, stiff: slab + dots',     None, 150.0, {}, 3, None),
    ('voids, stiff: tiled pass A',    None, 150.0, dict(allow_dots=False), 2, None),
    ('voids, all-gather interface',   None, 300.0, dict(force_exact=True), 2, 'exact'),
    ('solid box: deferred',           None, 3.0,   dict(solid=True), 3, 'deferred'),
    ('solid box, thin slabs, stiff',  16,   200.0, dict(solid=True, end_rows=True), 3, 'deferred_exact'),
    ('voids, uneven slabs',           'uneven', 0.05, {}, 3, None),
    ('solid box, uneven thin slabs',  'uneven16', 200.0, dict(solid=True, end_rows=True), 2, None),
    ('solid box, thin slabs, mesh all-gather', 16, 200.0, dict(solid=True, end_rows=True, mesh=True), 2, 'deferred_exact'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--transport', choices=['nccl', 'gloo-staged'], default='nccl')
    a = ap.parse_args()
    rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE']); local = int(os.environ.get('LOCAL_RANK', rank))
    from adi_thermal_fields_amd.dist_slab import rccl_env_defaults
    staged = a.transport == 'gloo-staged'
    if staged:
        local = 0
    rccl_env_defaults()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    tmo = datetime.timedelta(seconds=120)
    if staged:
        dist.init_process_group('gloo', timeout=tmo)
    else:
        dist.init_process_group('nccl', device_id=dev, timeout=tmo)
    assert dist.get_world_size() == world and dist.get_backend() == ('gloo' if staged else 'nccl')
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab
    dx = 1e-3
    mat = hip.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    failures = []
    for name, planes, cfl, opts, nsteps, want_form in CASES:
        rng = np.random.default_rng(77)
        if planes == 'uneven':
            nx = 64 * world - 2
        elif planes == 'uneven16':
            nx = 16 * world + 2
        else:
            nx = (planes or 64) * world
        shape = (nx, 24, 48)
        g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
        mask = (g[1] / 0.46) ** 2 + (g[2] / 0.47) ** 2 <= 1.0            # a cylinder along the sharded axis
        mask &= rng.random(shape) > 0.02                                    # ... with a few voids
        if opts.get('solid'):
            mask = np.ones(shape, bool)                                     # all-solid: the deferred forms
        T0 = rng.uniform(20.0, 900.0, shape)
        sizes = dist_slab.split_planes(nx, world)
        i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
        prm = hip.Params(cfl * dx * dx / alpha, 0.5)
        robin = 300.0
        if opts.get('end_rows'):
            # per-voxel Robin coefficients: the rows at the two global ends of every sharded-axis line differ from line to line
            robin = rng.uniform(100.0, 900.0, shape)
        bc = dict(robin_h=robin, neumann={'x+': 2e5})
        bc_local = dict(bc, robin_h=(np.ascontiguousarray(robin[i0:i1]) if isinstance(robin, np.ndarray) else robin))
        comm = dist_slab.HostStagedDistComm() if staged else \
            dist_slab.TorchDistComm(all_gather_mode='mesh' if opts.get('mesh') else 'auto')     # (auto: measured on the spot)
        st = dist_slab.SlabStepper(mask[i0:i1], dx, mat, prm, 20.0, comm=comm, **bc_local)
        st._allow_dots = opts.get('allow_dots', True); st._force_exact = opts.get('force_exact', False)
        st._allow_deferred_lines = opts.get('allow_deferred_lines', True)
        st._deferred_lines_cost_ratio = float('inf')        # (random voids flag nearly every line: the sparse pass on all of them)
        T = hip.to_device(np.ascontiguousarray(T0[i0:i1]))
        for s in range(nsteps):
            T = st.step(T, prefetch_halo=(s + 1 < nsteps))
        torch.cuda.synchronize()
        got = dist_slab.gather_slabs(T.t, sizes, host_staged=staged)
        # every rank must have picked the same interface form
        forms = [None] * world
        dist.all_gather_object(forms, st.axis0_mode)
        if rank == 0:
            got = got.cpu().numpy()
            grid = hip.Grid3D(*shape, dx, mask)
            packs = hip.precompute_coeff_packs_unified(grid, mat, **bc)
            W = hip.to_device(T0)
            for _ in range(nsteps):
                W = hip.adi_step_hip_coeff(W, grid, mat, prm, packs, Tinf=20.0)
            want = W.get()
            err = float(np.abs(got - want).max() / np.abs(want).max())
            print('[%s x%d] %s (cfl %g, planes %s): form %s, rel L-inf vs one domain %.3e'
                  % (a.transport, world, name, cfl, sizes, forms[0], err), flush=True)
            if not err <= 1e-12 or len(set(forms)) != 1 or (want_form is not None and forms[0] != want_form):
                failures.append((name, forms, err))
        dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and failures:
        print('FAILED', failures, flush=True)
        sys.exit(1)


if __name__ == '__main__':
    main()
