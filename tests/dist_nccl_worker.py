"""One rank of a real multi-process RCCL run of the slab decomposition (launched by tests/test_dist_nccl_spawn.py or by
hand:  python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P
tests/dist_nccl_worker.py).  Every rank builds the same seeded global case, steps its slab with SlabStepper + HipEngine
over torch.distributed 'nccl' (= RCCL), the slabs are gathered on rank 0 and compared with the one-domain HIP step of
the whole grid computed there: <= 1e-12 relative L-inf in every interface form (deferred with per-line solutions, window, slab / dots, exact, deferred).
Exit code 0 = all forms agree; the process group is created before any other GPU work, as RCCL wants it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE']); local = int(os.environ.get('LOCAL_RANK', rank))
    from adi_thermal_fields_amd.dist_slab import rccl_env_defaults
    rccl_env_defaults()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    assert dist.get_world_size() == world and dist.get_backend() == 'nccl'
    import adi_thermal_fields_amd.adi3d_hip_coeff as hip
    from adi_thermal_fields_amd import dist_slab
    rng = np.random.default_rng(77)
    nx = 64 * world
    shape = (nx, 24, 48)
    g = np.meshgrid(*[(np.arange(n) + 0.5) / n - 0.5 for n in shape], indexing='ij')
    mask = (g[1] / 0.46) ** 2 + (g[2] / 0.47) ** 2 <= 1.0            # a cylinder along the sharded axis
    mask &= rng.random(shape) > 0.02                                    # ... with a few voids
    dx = 1e-3
    mat = hip.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    T0 = rng.uniform(20.0, 900.0, shape)
    sizes = dist_slab.split_planes(nx, world)
    i0 = sum(sizes[:rank]); i1 = i0 + sizes[rank]
    worst = 0.0
    failures = []
    mask_voids = mask
    for cfl, opts, nsteps in ((0.05, {}, 3), (0.05, dict(allow_deferred_lines=False), 3), (150.0, {}, 3),
                              (150.0, dict(allow_dots=False), 2), (300.0, dict(force_exact=True), 2), (3.0, dict(solid=True), 3)):
        mask = np.ones(shape, bool) if opts.get('solid') else mask_voids      # all-solid: the deferred form
        prm = hip.Params(cfl * dx * dx / alpha, 0.5)
        st = dist_slab.SlabStepper(mask[i0:i1], dx, mat, prm, 20.0, robin_h=300.0, neumann={'x+': 2e5},
                                   comm=dist_slab.TorchDistComm())
        st._allow_dots = opts.get('allow_dots', True); st._force_exact = opts.get('force_exact', False)
        st._allow_deferred_lines = opts.get('allow_deferred_lines', True)      # (voids + decay: the per-line deferred form)
        T = hip.to_device(np.ascontiguousarray(T0[i0:i1]))
        for s in range(nsteps):
            T = st.step(T, prefetch_halo=(s + 1 < nsteps))
        torch.cuda.synchronize()
        mine = T.t.contiguous()
        parts = [torch.empty((sizes[r],) + shape[1:], dtype=torch.float64, device=dev) for r in range(world)] if rank == 0 else None
        # slabs may differ by two planes: gather through point-to-point copies of the right sizes
        if rank == 0:
            parts[0].copy_(mine)
            for r in range(1, world):
                dist.recv(parts[r], src=r)
        else:
            dist.send(mine, dst=0)
        if rank == 0:
            got = torch.cat(parts, dim=0).cpu().numpy()
            grid = hip.Grid3D(*shape, dx, mask)
            packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0, neumann={'x+': 2e5})
            W = hip.to_device(T0)
            for _ in range(nsteps):
                W = hip.adi_step_hip_coeff(W, grid, mat, prm, packs, Tinf=20.0)
            want = W.get()
            err = float(np.abs(got - want).max() / np.abs(want).max())
            worst = max(worst, err)
            print('[nccl x%d] cfl %g %s: form %s, rel L-inf vs one domain %.3e' % (world, cfl, opts, st.axis0_mode, err), flush=True)
            if not err <= 1e-12:
                failures.append((cfl, opts, st.axis0_mode, err))
        dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and failures:
        print('FAILED', failures, flush=True)
        sys.exit(1)


if __name__ == '__main__':
    main()
