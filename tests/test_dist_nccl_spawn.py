"""GPU, more than one device: SlabStepper + HipEngine over REAL multi-process RCCL ('nccl' backend), one process per
GPU, against the one-domain HIP step (tests/dist_hip_worker.py --transport nccl).  Skipped on a one-GPU box -- there the transport is
covered by the single-rank RCCL self-loop test and the gloo multi-process tests.  The ranks are started by
torch.distributed.run as a CHILD process (nothing is exec'ed in place of this GPU-initialised test process), at most 4 of
them (the GPU boxes allow few processes per card)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_real_rccl_ranks_match_one_domain():
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip('needs at least two GPUs (found %d)' % ndev)
    world = min(ndev, 4)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, 'dist_hip_worker.py'), '--transport', 'nccl']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-4000:]
    import dist_hip_worker
    assert r.stdout.count('rel L-inf vs one domain') == len(dist_hip_worker.CASES)
