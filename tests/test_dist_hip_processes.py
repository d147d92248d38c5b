"""GPU (one is enough): several REAL processes drive HipEngine + SlabStepper.  Two / three / four fresh `torch.distributed.run`
children, backend gloo, every rank on cuda:0, payloads staged through pinned host memory (dist_slab.HostStagedDistComm, a test
transport), run every case of tests/dist_hip_worker.py -- each interface form, uneven slabs, thin slabs with per-line end rows
-- against the one-domain HIP step: <= 1e-12.  What this covers that the other slab tests do not: separate processes with
separate HIP contexts and allocators and a real rendezvous on the HIP engine (the gloo tests run a CPU engine, the in-process
ranks are threads of one context, the RCCL self-loop is one rank).  The real `nccl` transport between GPUs stays with
tests/test_dist_nccl_spawn.py, which needs two devices.

The children are started from this process as ordinary child processes (nothing is exec'ed in place of a process that has
initialised the GPU); 1 + 4 processes on the card at most (the GPU boxes allow 6)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('world', [2, 3, 4])
def test_real_processes_on_one_gpu_match_one_domain(world):
    import dist_hip_worker
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, 'dist_hip_worker.py'),
           '--transport', 'gloo-staged']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    print(r.stdout[-6000:])
    assert r.returncode == 0, r.stdout[-6000:]
    assert r.stdout.count('rel L-inf vs one domain') == len(dist_hip_worker.CASES)
    assert 'FAILED' not in r.stdout
