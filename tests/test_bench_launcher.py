"""CPU: `python bench.py --gpus N` started plainly must start its own ranks.  The launcher is driven
with a stub worker over gloo, so the plumbing -- fresh child processes through torch.distributed.run, rendezvous on
127.0.0.1, rank 0's JSON line relayed, non-zero exit with the failing rank's tail, timeout -- is proven without a GPU;
and the real bench.py is started with --gpus 2 in this GPU-less container, where it must fail INSIDE the ranks with a
clear message instead of refusing in the parent."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STUB = os.path.join(HERE, 'stub_rank_worker.py')


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.parametrize('world', [2, 4])
def test_launcher_relays_rank0_line(world):
    rc, line, tail = _bench().launch_ranks(world, STUB, ['ok', '--steps', '3'], 120)
    assert rc == 0, tail
    d = json.loads(line)
    assert d['metric'] == 'stub' and d['n_gpus'] == world and d['value'] == world * (world + 1) / 2
    assert d['args'] == ['--steps', '3']


def test_launcher_reports_failing_rank():
    rc, line, tail = _bench().launch_ranks(2, STUB, ['fail'], 120)
    assert rc != 0 and line is None
    assert 'stub rank 1: deliberate failure' in tail


def test_launcher_without_a_line_is_a_failure():
    b = _bench()
    rc, line, tail = b.launch_ranks(2, STUB, ['mute'], 120)
    assert rc == 0 and line is None


def test_launcher_timeout_kills_its_own_process_group():
    t0 = time.time()
    rc, line, tail = _bench().launch_ranks(2, STUB, ['hang'], 8)
    assert rc == 124 and line is None
    assert time.time() - t0 < 60


def test_timeouts_stay_inside_the_drivers_limit():
    """the driver kills `bench.py` at 600 s: the launcher's own limit and the process group's collective timeout must fire
    before that, so that a hang ends with the tail of the ranks' stderr / a watchdog stack trace instead of nothing"""
    a = _bench().parse([])
    assert a.launch_timeout <= 480.0 and a.pg_timeout <= 120.0
    assert a.launch_timeout + 60.0 < 600.0 and a.pg_timeout < a.launch_timeout
    calls = [ln for ln in open(os.path.join(ROOT, 'bench.py')) if 'init_process_group(' in ln]
    assert len(calls) >= 4 and all('timeout=' in ln for ln in calls), calls


def test_launcher_clears_inherited_rank_environment():
    """a parent that itself sits inside a torchrun job must not leak its RANK / WORLD_SIZE into the children"""
    os.environ['WORLD_SIZE'] = '7'
    try:
        rc, line, tail = _bench().launch_ranks(2, STUB, ['ok'], 120)
    finally:
        del os.environ['WORLD_SIZE']
    assert rc == 0 and json.loads(line)['n_gpus'] == 2, tail


@pytest.mark.skipif(torch.cuda.is_available(), reason='this test is about the GPU-less container')
def test_plain_gpus_2_fails_inside_the_ranks_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode not in (0, 2), (r.returncode, r.stderr[-2000:])     # 2 was the old parent-side refusal
    assert 'no GPU visible' in r.stderr and 'bench.py rank' in r.stderr, r.stderr[-2000:]
    assert 'must be launched with torch.distributed.run' not in r.stderr
    assert r.stdout.strip() == ''
