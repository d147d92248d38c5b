"""GPU: bench.py prints ONE JSON line with the contract's keys (small grid, a few steps): the single-GPU line with `roofline`,
`cpu_baseline` and `parity_rel_linf`, the rehearsal lines of both scaling modes marked as rehearsals, the cylindrical and the
curved-solid variants."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline']


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                   # exactly one line on stdout
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d['higher_is_better'] is True and d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'hbm' and rf['peak'] == 8000.0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    assert d['value'] > 0 and d['ms_per_step'] > 0 and 'workload' in d['config'] and 'model' not in d['config']
    return d


def test_single_gpu_line_with_cpu_baseline_and_parity():
    d = _bench('--n', '64', '--cpu-n', '64', '--steps', '4', '--warmup', '2')
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['scaling'] == 'weak'
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['cores'] == 1 and d['cpu_baseline_all_cores']['cores'] >= 1
    assert d['parity_rel_linf'] <= 1e-10 and d['parity']['steps'] == 3
    assert 'north_star_x_sweep' in d and d['north_star_x_sweep']['target_frac'] == 0.60
    assert 'prewarm' in d and set(d['general_pack_sweeps_42B']) == {'sweep_axis0', 'sweep_axis1', 'sweep_axis2_contig'}
    # the other BASELINE.json configurations ride in the same line (after the timed region)
    al = d['also']
    assert set(al) == {'config2_256', 'config4_cyl', 'ellipsoid_64', 'robin_field_ellipsoid_64'}
    for k, v in al.items():
        assert v['ms_per_step'] > 0 and v['steps_per_s'] > 0 and len(v['kernels']) >= 3, k
        for kk in v['kernels'].values():
            assert kk['ms'] > 0 and kk['bytes_per_cell'] >= 16 and 0 < kk['frac'] < 1, (k, kk)
    assert al['config4_cyl']['roofline']['bound'] == 'hbm' and set(al['config4_cyl']['kernels']) == {'sweep_r', 'sweep_phi', 'sweep_z_contig'}
    rf = al['robin_field_ellipsoid_64']
    assert rf['bytes_per_cell_step'] > al['ellipsoid_64']['bytes_per_cell_step']      # its sweeps load coefficient arrays
    assert rf['target_vs_scalar_h'] == 1.10 and rf['vs_scalar_h_ellipsoid'] > 0


@pytest.mark.parametrize('scaling,form', [('weak', 'deferred_exact'), ('strong', 'deferred_exact')])   # (256 rows at cfl 200: no decay)
def test_rehearsal_lines_are_marked(scaling, form):
    d = _bench('--n', '256', '--steps', '3', '--warmup', '2', '--no-cpu', '--rehearse-world', '4', '--scaling', scaling)
    assert 'rehearsal' in d and d['n_gpus'] == 1 and d['scaling'] == scaling
    assert d['comm_overlap']['axis0_interface'] == form and d['comm_overlap']['selfcheck_rel_diff'] <= 1e-12
    assert d['config']['planes_per_gpu'] == (256 if scaling == 'weak' else 64)


def test_curved_solid_and_cylindrical_lines():
    d = _bench('--n', '128', '--steps', '3', '--warmup', '2', '--no-cpu', '--mask', 'ellipsoid')
    assert 'ellipsoid' in d['config']['workload']
    d = _bench('--config', 'cyl', '--steps', '4', '--warmup', '2', '--no-cpu')
    assert d['metric'].startswith('adi_cyl') and set(d['kernels']) == {'sweep_r', 'sweep_phi', 'sweep_z_contig'}


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_two_real_ranks_line_proves_itself_against_one_domain(scaling):
    """`bench.py --gpus 2` started plainly: the parent starts two ranks (separate processes), here on the one GPU of the box
    over the gloo-staged test transport.  The line of an N > 1 run must carry `parity_vs_one_domain` -- 3 steps of the n^3 grid
    cut over the ranks against the same steps on one domain, <= 1e-12 -- and, with weak scaling, the
    same for the timed job's own slab thickness; plus the `ranks` record."""
    d = _bench('--gpus', '2', '--n', '128', '--steps', '3', '--warmup', '2', '--no-cpu', '--transport', 'gloo-staged',
               '--scaling', scaling)
    assert d['n_gpus'] == 2 and d['scaling'] == scaling
    p = d['parity_vs_one_domain']
    assert p['ok'] is True and p['rel_linf'] <= 1e-12 and p['bar'] == 1e-12 and p['steps'] == 3 and p['forms_agree']
    assert p['grid'] == '128x128x128' and p['planes_per_rank'] == [64, 64] and p['form'] is not None
    if scaling == 'weak':
        w = p['weak_form']
        assert w['ok'] is True and w['rel_linf'] <= 1e-12 and w['planes_per_rank'] == [128, 128] and w['grid'] == '256x128x128'
    else:
        assert 'weak_form' not in p
    r = d['ranks']
    assert r['world_size_from_process_group'] == 2 and 'gloo-staged' in r['transport'] and len(r['loop_seconds_per_rank']) == 2
    assert d['comm_overlap']['selfcheck_rel_diff'] <= 1e-12


def test_single_rank_rccl_line_runs_the_parity_calls_over_nccl():
    """--force-dist: the slab path over a one-rank RCCL process group; the calls of parity_vs_one_domain (gather, object
    all-gather, the one-domain leg) go through the real `nccl` backend here, the only place a one-GPU box can exercise them"""
    d = _bench('--n', '128', '--steps', '3', '--warmup', '2', '--no-cpu', '--force-dist')
    p = d['parity_vs_one_domain']
    assert p['ok'] is True and p['rel_linf'] <= 1e-12 and p['planes_per_rank'] == [128]
    assert d['ranks']['backend'] == 'nccl' and d['ranks']['world_size_from_process_group'] == 1


def test_cylindrical_replicas_line():
    """`bench.py --config cyl --gpus 2`: the cylindrical path does not shard -- N independent replicas, the
    barrier and max-over-ranks timing of the contract; two real processes on the one GPU over the gloo-staged test transport"""
    d = _bench('--config', 'cyl', '--gpus', '2', '--steps', '4', '--warmup', '2', '--no-cpu', '--transport', 'gloo-staged')
    assert d['n_gpus'] == 2 and d['config']['decomposition'] == 'replicas only' and d['metric'].startswith('adi_cyl')
