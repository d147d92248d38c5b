"""Try the real TorchDistComm path with 2 ranks sharing GPU 0 (RCCL may refuse duplicate devices; then this
just reports the refusal).  Small grid, compares against the single-domain HIP step."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl')
    x = torch.ones(4, device='cuda') * rank
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print('rank', rank, 'allreduce ok', x[0].item(), flush=True)
except Exception as e:
    print('rank', rank, 'NCCL on a shared GPU refused:', repr(e)[:300], flush=True)
    sys.exit(0)
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab
rng = np.random.default_rng(1)
shape = (128, 40, 48)
mask = rng.random(shape) > 0.05
T0 = rng.uniform(20, 900, shape)
dx = 1e-3; alpha = 54.0 / (7800 * 490)
mat = hip.Material(7800.0, 490.0, 54.0); prm = hip.Params(150 * dx * dx / alpha, 0.5)
i0, i1 = (0, 64) if rank == 0 else (64, 128)
st = dist_slab.SlabStepper(mask[i0:i1], dx, mat, prm, 20.0, robin_h=300.0)
T = hip.to_device(np.ascontiguousarray(T0[i0:i1]))
print('rank', rank, 'selfcheck', st.self_check(T), flush=True)
for _ in range(3):
    T = st.step(T)
got = T.get()
grid = hip.Grid3D(*shape, dx, mask)
packs = hip.precompute_coeff_packs_unified(grid, mat, robin_h=300.0)
W = hip.to_device(T0)
for _ in range(3):
    W = hip.adi_step_hip_coeff(W, grid, mat, prm, packs, Tinf=20.0)
want = W.get()[i0:i1]
print('rank', rank, 'rel diff vs single domain', float(np.abs(got - want).max() / np.abs(want).max()), flush=True)
dist.destroy_process_group()
