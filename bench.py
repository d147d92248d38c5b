#!/usr/bin/env python3
"""bench.py -- headline benchmark: ADI steps/s and HBM GB/s, 512^3 fp64 Cartesian Robin, 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 512] [--scaling weak|strong] [--mask box|ellipsoid] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) this process touches no GPU: it starts the N
ranks itself as fresh child processes (torch.distributed.run, one per GPU), relays rank 0's JSON line and exits
non-zero with the failing rank's tail if a rank fails or the run exceeds --launch-timeout.

One "step" = one full ADI time step (explicit stage + three implicit sweeps) of the hot path over a
synthetic field that is already resident in HBM.  Workload (BASELINE.json configs[2] / SURVEY.md 8(d)
config 3): n^3 cells per GPU (n = 512), all-solid mask, dx = 5e-4, steel, Robin h = 500 on all six faces,
Tinf = 20, theta = 0.5, cfl = 200, T0 ~ U(20, 1000) seeded per rank.  For N > 1 the grid is cut into slabs along
memory axis 0: (N*n, n, n) with --scaling weak (the default: n^3 cells per GPU), (n, n, n) with --scaling strong
(BASELINE.json configs[2]: 512^3 over 8 GPUs = 64 planes per GPU).  Halo planes for the explicit stage and the
reduced interface system of the sharded-axis sweep travel over RCCL.

Prints ONE JSON line on rank 0.  `value` = 512^3-cell-equivalent ADI steps per second summed over all
ranks (= plain steps/s at N = 1, n = 512).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)      # >= 10 warm-up + >= 50 timed steps
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n', type=int, default=512, help='cells per axis per GPU')
    ap.add_argument('--config', choices=['cart', 'cyl'], default='cart',
                    help="cart: the headline 512^3 Cartesian workload (default); cyl: BASELINE.json configs[3], the "
                         "cylindrical 128 x 256 x 512 BE step on one GPU (replicas only at N > 1: it does not shard)")
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-n', type=int, default=512, help='edge of the CPU-baseline sample (512: BASELINE.md section 3)')
    ap.add_argument('--force-dist', action='store_true',
                    help='one GPU only: run the slab code path over a real single-rank RCCL process group (checks the '
                         'torch.distributed plumbing: init, all_gather, all_reduce, barrier); marked in the JSON line')
    ap.add_argument('--rehearse-world', type=int, default=0,
                    help='one GPU only: run the code path of a middle rank of a W-GPU job with a loopback communicator '
                         '(no wire time); the line is marked "rehearsal" and is not a measurement of W GPUs')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='N > 1: weak = n^3 cells per GPU, global (N*n, n, n) (default); strong = n^3 cells in total, '
                         'n/N planes per GPU (BASELINE.json configs[2]: 512^3 over 8 GPUs = 64 planes each)')
    ap.add_argument('--mask', choices=['box', 'ellipsoid'], default='box',
                    help='box: all-solid (the headline workload); ellipsoid: the same box holding a curved solid '
                         '(semi-axes 0.47 / 0.49 / 0.48 of the box), i.e. every in-mask line crosses the surface twice')
    ap.add_argument('--launch-timeout', type=float, default=480.0,
                    help='seconds the self-started ranks of a plain `--gpus N` run may take before they are killed (below the '
                         "driver's own 600 s limit, so that a hang still ends with the tail of the ranks' stderr)")
    ap.add_argument('--transport', choices=['nccl', 'gloo-staged'], default='nccl',
                    help='N > 1: nccl = RCCL, one rank per GPU (default, the product path); gloo-staged = a TEST transport that puts '
                         'every rank on cuda:0 and stages the payloads through pinned host memory (dist_slab.HostStagedDistComm): '
                         'runs the whole N > 1 code path with real processes on a one-GPU box; the line is marked')
    ap.add_argument('--no-also', action='store_true',
                    help='N = 1: skip the `also` object (the other BASELINE.json configs, 20 steps each, after the timed region)')
    ap.add_argument('--pg-timeout', type=float, default=120.0,
                    help='seconds a collective of the process group may wait before the rank aborts with a stack trace')
    return ap.parse_args(argv)


# ---- plain `python bench.py --gpus N`: this process starts the ranks -------------------------------------------------
def launch_ranks(nproc, script, script_args, timeout_s, extra_env=None):
    """Start `nproc` ranks of `script` as CHILD processes through torch.distributed.run (one per GPU, rendezvous on
    127.0.0.1 at a free port), wait for them and return (exit code, the last JSON line a rank wrote to stdout or None,
    tail of the ranks' stderr).  The caller has not touched the GPU and nothing is exec'ed in its place.  A run that
    exceeds `timeout_s` is ended by killing the process GROUP this call created (never a pattern) and reports 124."""
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port), script] + list(script_args)
    with tempfile.TemporaryFile(mode='w+') as ferr:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=ferr, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
            rc = p.returncode
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGTERM)
                out, _ = p.communicate(timeout=20)
            except (subprocess.TimeoutExpired, ProcessLookupError):
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                out, _ = p.communicate()
            rc = 124
        ferr.seek(0)
        err = ferr.read()
    line = None
    for ln in (out or '').splitlines():
        ln = ln.strip()
        if ln.startswith('{') and ln.endswith('}'):
            try:
                if 'metric' in json.loads(ln):
                    line = ln
            except ValueError:
                pass
    return rc, line, '\n'.join(err.splitlines()[-60:])


def launch_self(a, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks, relay rank 0's line."""
    # the arguments travel in the environment: torch.distributed.run's own parser claims abbreviations of ITS options among the
    # script's arguments (`--n 128` is "ambiguous: --nnodes, --nproc-per-node, ...")
    rc, line, tail = launch_ranks(a.gpus, os.path.abspath(__file__), [], a.launch_timeout,
                                  extra_env={'ADI_BENCH_ARGV': json.dumps(list(argv))})
    if rc == 0 and line is not None:
        print(line, flush=True)
        return 0
    why = ('timed out after %.0f s (--launch-timeout)' % a.launch_timeout if rc == 124 else
           'exit code %d' % rc if rc != 0 else 'no JSON line from rank 0')
    print('bench.py --gpus %d: the ranks started with torch.distributed.run failed: %s\n---- tail of the ranks\' stderr '
          '----\n%s' % (a.gpus, why, tail), file=sys.stderr, flush=True)
    return rc if rc != 0 else 1


def need_gpu(rank):
    if not torch.cuda.is_available():
        print('bench.py rank %d: no GPU visible (torch.cuda.is_available() is False); the benchmark has no CPU path'
              % rank, file=sys.stderr, flush=True)
        sys.exit(3)


def make_mask(kind, shape, i0=0, nx_global=None):
    """mask of planes [i0, i0 + shape[0]) of a (nx_global, ny, nz) grid: all-solid box, or an ellipsoid with semi-axes
    0.47 / 0.49 / 0.48 of the box (a curved solid: every in-mask line crosses the surface twice)"""
    nxl, ny, nz = shape
    nxg = nxl if nx_global is None else nx_global
    if kind == 'box':
        return np.ones(shape, bool)
    x = ((np.arange(i0, i0 + nxl) + 0.5) / nxg - 0.5) / 0.47
    y = ((np.arange(ny) + 0.5) / ny - 0.5) / 0.49
    z = ((np.arange(nz) + 0.5) / nz - 0.5) / 0.48
    return (x[:, None, None] ** 2 + y[None, :, None] ** 2 + z[None, None, :] ** 2) <= 1.0


def cpu_baseline(n, mask=None, T0=None, seed=0, steps_1t=3, steps_all=9):
    """The oracle (oracle/adi_oracle.c, a port of the reference's Numba path) timed on this host's cores on a bounded
    sample of the same workload: n^3 cells (default 512) x `steps_1t` steps on ONE thread -- the
    reference's kernels are serial -- and x `steps_all` steps on all cores (OpenMP over lines, working set first-touched
    in parallel).  `mask` / `T0`: the mask and the initial field of the timed workload (so that the single-thread leg's
    field after `steps_1t` steps can be compared with the GPU's).  Returns (single-thread dict, all-cores dict, field
    after `steps_1t` steps)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import adi_oracle as orc
    shape = (n, n, n)
    grid = orc.Grid3D(n, n, n, 5e-4, np.ones(shape, bool) if mask is None else mask)
    mat = orc.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    prm = orc.Params(200.0 * grid.dx ** 2 / alpha, 0.5)
    packs = orc.precompute_coeff_packs_unified(grid, mat, robin_h=500.0, _share=True)
    if T0 is None:
        T0 = np.random.default_rng(seed).uniform(20.0, 1000.0, shape)
    out = []
    field = None
    for omp, steps in ((False, steps_1t), (True, steps_all)):
        if omp:
            orc.adi_run(T0, grid, mat, prm, packs, Tinf=20.0, nsteps=1, omp=True)   # warm the OpenMP pool
        t0 = time.perf_counter()
        res = orc.adi_run(T0, grid, mat, prm, packs, Tinf=20.0, nsteps=steps, omp=omp)
        dt = time.perf_counter() - t0
        if not omp:
            field = res
        del res
        cells_per_s = steps * n ** 3 / dt
        out.append(dict(value=cells_per_s / 512 ** 3, unit='steps/s (512^3-cell equivalent)',
                        cores=(os.cpu_count() if omp else 1), kind='port',
                        sample='%d^3 cells x %d steps of the same Robin workload, %.1f s, oracle/adi_oracle%s.c%s'
                               % (n, steps, dt, '_omp' if omp else '',
                                  ' (time includes the parallel first-touch copy of the working set)' if omp else ''),
                        cell_updates_per_s=cells_per_s))
    return out[0], out[1], field


# The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout of every process that
# initialises it, so file descriptor 1 is pointed at stderr for the whole run and the line is written to a saved
# duplicate of the original stdout.
_STDOUT_FD = None


def claim_stdout():
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(text):
    sys.stdout.flush()
    if _STDOUT_FD is None:
        print(text, flush=True)
    else:
        os.write(_STDOUT_FD, (text + '\n').encode())


class Collect:
    """the bench's own small collectives (timings, per-rank records, the barrier): device tensors over nccl, host tensors
    over gloo (the gloo-staged test transport has no device collectives)"""

    def __init__(self, dist, active, staged, dev):
        self.dist, self.active, self.staged, self.dev = dist, active, staged, dev

    def _t(self, vals):
        return torch.tensor(vals, dtype=torch.float64, device='cpu' if self.staged else self.dev)

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def max(self, x):
        if not self.active:
            return float(x)
        t = self._t([float(x)])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_rows(self, vals):
        """every rank's list of floats -> (world, len) array on every rank"""
        t = self._t(list(vals))
        out = torch.empty(self.dist.get_world_size() * t.numel(), dtype=torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.view(self.dist.get_world_size(), -1).cpu().numpy()


def seeded_planes(i0, i1, ny, nz, seed=1):
    """planes [i0, i1) of the seeded host field T0[i] = default_rng((seed, i)).uniform(20, 1000, (ny, nz)): every rank can make
    its own planes, rank 0 the whole field, without anyone generating what it does not need"""
    out = np.empty((i1 - i0, ny, nz))
    for i in range(i0, i1):
        out[i - i0] = np.random.default_rng((seed, i)).uniform(20.0, 1000.0, (ny, nz))
    return out


def parity_vs_one_domain(a, adi, dist_slab, dist, world, rank, mat, prm, dx, Tinf, make_comm, staged, sizes, ny, nz, label):
    """"the multi-GPU result must equal the 1-GPU result to <= 1e-12".  PARITY_STEPS steps of a grid of
    sum(sizes) x ny x nz cells cut into this job's slabs (sizes[r] planes on rank r, the interface form SlabStepper picks for
    them, halos prefetched as in the timed loop) from a seeded host field; the slabs are gathered on rank 0, which takes the
    same steps on ONE domain with the single-GPU stepper and compares.  Returns the record for the JSON line (rank 0; None
    elsewhere).  Collective; runs before the timed loop; an exception on any rank is reported in the record, not raised."""
    PARITY_STEPS = 3
    nx = int(sum(sizes))
    i0 = int(sum(sizes[:rank])); i1 = i0 + int(sizes[rank])
    rec = dict(grid='%dx%dx%d' % (nx, ny, nz), planes_per_rank=[int(v) for v in sizes], steps=PARITY_STEPS, bar=1e-12, what=label)
    err, form = None, None
    try:
        mask = make_mask(a.mask, (i1 - i0, ny, nz), i0, nx)
        T = adi.to_device(seeded_planes(i0, i1, ny, nz))
        st = dist_slab.SlabStepper.from_local(T.t, mask, dx, mat, prm, Tinf, robin_h=500.0, comm=make_comm())
        for s_ in range(PARITY_STEPS):
            T = st.step(T, prefetch_halo=(s_ + 1 < PARITY_STEPS))
        torch.cuda.synchronize()
        form = st.axis0_mode
        full = dist_slab.gather_slabs(T.t, [int(v) for v in sizes], host_staged=staged)
        del st, T
    except Exception as e:          # noqa: BLE001 -- the line must still be printed
        err, full = '%s: %s' % (type(e).__name__, e), None
    forms = [None] * world
    dist.all_gather_object(forms, (form, err))
    if rank != 0:
        return None
    rec['form'] = forms[0][0]
    rec['forms_agree'] = len(set(f for f, _ in forms)) == 1
    errs = ['rank %d: %s' % (r_, e) for r_, (_, e) in enumerate(forms) if e]
    if errs or full is None:
        rec.update(rel_linf=None, ok=False, error='; '.join(errs) or 'no field gathered')
        return rec
    try:
        gmask = make_mask(a.mask, (nx, ny, nz))
        grid = adi.Grid3D(nx, ny, nz, dx, gmask)
        one = adi.StagedStepper(grid, mat, prm, adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0), Tinf)
        W = adi.to_device(seeded_planes(0, nx, ny, nz))
        for _ in range(PARITY_STEPS):
            W = one.step(W)
        den = float(W.t.abs().max().item())
        rel = float((full - W.t).abs().max().item()) / (den if den > 0 else 1.0)
        rec.update(rel_linf=rel, ok=bool(rel <= rec['bar'] and rec['forms_agree']),
                   against='adi3d_hip_coeff.StagedStepper on the whole grid on rank 0 (one domain), same seeded T0 and mask')
        del one, W, grid, full
        torch.cuda.empty_cache()
    except Exception as e:          # noqa: BLE001
        rec.update(rel_linf=None, ok=False, error='one-domain leg on rank 0: %s: %s' % (type(e).__name__, e))
    return rec


def quiet_gc():
    """Python's cyclic collector out of the timed region, as `timeit` does: a full collection over the heap of a process that
    has imported torch takes 35 - 60 ms and fires whenever the allocation counters say so -- scripts/slab_outlier_probe.py
    caught it in the middle of a step loop (one 38 - 58 ms step among 1.2 ms ones, no allocator or kernel activity behind it):
    the "36 ms outliers" of round 3.  Collect now, park the survivors in the permanent generation, switch the collector off;
    the caller switches it on again after the loop."""
    gc.collect()
    gc.freeze()
    gc.disable()


def make_events(k):
    """k timing events that already exist on the device: torch creates the HIP event at the first record(), a few microseconds
    each -- hundreds of them inside a 30 ms timed region are a measurable part of it.  Recorded once here, re-recorded in the loop."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    for e in ev:
        e.record()
    return ev


def timed_steps(step_fn, nstages, steps=20, warmup=5, every=1):
    """`steps` calls of step_fn(events or None) after `warmup` untimed ones: (ms per step between two events around the whole
    loop, per-stage mean ms from events recorded on every `every`-th step -- small kernels notice their own event records)"""
    quiet_gc()             # BEFORE the warm-up: a collection takes 0.1 s or more of host time, the GPU idles and clocks down, and the
    evs = {s_: make_events(nstages + 1) for s_ in range(steps) if s_ % every == 0}
    e0, e1 = make_events(2)
    for _ in range(warmup):    # first steps after it run slow; the warm-up must flow straight into the timed loop
        step_fn(None)
    torch.cuda.synchronize()
    e0.record()
    for s_ in range(steps):
        step_fn(evs.get(s_))
    e1.record(); e1.synchronize()
    gc.enable()
    st = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(nstages)] for e in evs.values()])
    return e0.elapsed_time(e1) / steps, st.mean(axis=0)


def _kernel_table(names, bpc, ms, cells):
    out = {}
    for nm, b, m in zip(names, bpc, ms):
        gbs = b * cells / (m * 1e-3) / 1e9
        out[nm] = dict(ms=round(float(m), 4), bytes_per_cell=round(float(b), 3), achieved_gbs=round(gbs, 1),
                       frac=round(gbs / HBM_PEAK_GBS, 4))
    return out


def also_configs(a, adi, mat, prm, dx, Tinf, dev):
    """The other BASELINE.json configurations and the per-voxel-h workload in the one line the driver records: 20 steps each
    (5 warm-up), AFTER the timed region and outside it, fields resident in HBM, same event method.
      config2_256                 configs[1]: 256^3 Cartesian Robin, all-solid
      config4_cyl                 configs[3]: cylindrical 128 x 256 x 512 BE (the loop of `bench.py --config cyl`)
      ellipsoid_<n>               the headline box holding a curved solid, per-face scalar h (`--mask ellipsoid`)
      robin_field_ellipsoid_<n>   the same solid with PER-VOXEL Robin coefficients on five faces + a Neumann face: the packs
                                  of the STL-corrected drivers (quick_compare_robin_end_robin_corrected.py:174-207,
                                  voxel_bc_correction.py:110-167), whose sweeps load their coefficients behind the flags
    Every entry: ms_per_step, steps_per_s, kernels {stage: ms, bytes_per_cell, achieved_gbs, frac}."""
    out = {}

    def cart(n_, mask, packs_kw, tag):
        grid = adi.Grid3D(n_, n_, n_, dx, mask)
        packs = adi.precompute_coeff_packs_unified(grid, mat, **packs_kw)
        stp = adi.StagedStepper(grid, mat, prm, packs, Tinf)
        gen = torch.Generator(device=dev); gen.manual_seed(5)
        state = [adi.DeviceField(torch.rand((n_, n_, n_), dtype=torch.float64, device=dev, generator=gen) * 980.0 + 20.0)]

        def step(ev):
            state[0] = stp.step(state[0], events=ev)
        ms, st = timed_steps(step, len(stp.stage_names), every=(4 if n_ <= 256 else 1))
        assert bool(torch.isfinite(state[0].t).all().item())
        out[tag] = dict(ms_per_step=round(ms, 4), steps_per_s=round(1e3 / ms, 2), cells=n_ ** 3,
                        in_mask_fraction=round(float(np.mean(mask)), 4),
                        bytes_per_cell_step=round(float(sum(stp.stage_bytes_per_cell)), 3),
                        kernels=_kernel_table(stp.stage_names, stp.stage_bytes_per_cell, st, n_ ** 3))
        del stp, packs, grid, state
        torch.cuda.empty_cache()

    cart(256, np.ones((256, 256, 256), bool), dict(robin_h=500.0), 'config2_256')
    n = a.n
    emask = make_mask('ellipsoid', (n, n, n))
    cart(n, emask, dict(robin_h=500.0), 'ellipsoid_%d' % n)
    # per-voxel h: a smooth field of the order of the scalar one (the corrected drivers scale h by a projected-area ratio per
    # exposed voxel), built on the device; five faces carry it, 'z-' has none and takes a flux instead, as in the reference's driver
    ax = [torch.linspace(0.0, 1.0, n, dtype=torch.float64, device=dev) for _ in range(3)]
    hfield = 500.0 * (1.0 + 0.2 * torch.sin(7.0 * ax[0])[:, None, None] * torch.cos(5.0 * ax[1])[None, :, None]
                      + 0.1 * torch.sin(9.0 * ax[2])[None, None, :])
    cart(n, emask, dict(robin_h={f: hfield for f in ('x-', 'x+', 'y-', 'y+', 'z+')}, neumann={'z-': 2.0e5}),
         'robin_field_ellipsoid_%d' % n)
    del hfield
    e, r = out['ellipsoid_%d' % n], out['robin_field_ellipsoid_%d' % n]
    r['vs_scalar_h_ellipsoid'] = round(r['ms_per_step'] / e['ms_per_step'], 4)
    r['target_vs_scalar_h'] = 1.10

    import adi_thermal_fields_amd.adi3d_hip_cyl as cyl
    nr, nphi, nz = 128, 256, 512
    g = cyl.GridCyl(nr, nphi, nz, 2.5e-4, 2 * np.pi / nphi, 2.5e-4, 0.032)
    cst = cyl.StagedCylStepper(g, cyl.Material(7800.0, 490.0, 54.0), cyl.Params(0.05, 1.0, "be"), cyl.RobinR(400.0, 20.0),
                               cyl.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0))
    T0 = np.full((nr, nphi, nz), 20.0); T0[:, :, -16:] = 1000.0
    X = g.layout.empty(); X.copy_(cyl.to_device(T0).t)
    ms, st = timed_steps(lambda ev: cst._step_inplace(X, events=ev), 3, every=4)
    N = nr * nphi * nz
    kt = _kernel_table(cst.stage_names, cst.stage_bytes_per_cell, st, N)
    dom = max(kt, key=lambda k: kt[k]['ms'])
    out['config4_cyl'] = dict(ms_per_step=round(ms, 4), steps_per_s=round(1e3 / ms, 2), cells=N, kernels=kt,
                              roofline=dict(bound='hbm', kernel=dom, achieved=kt[dom]['achieved_gbs'], peak=HBM_PEAK_GBS,
                                            unit='GB/s', frac=kt[dom]['frac'], traffic=measured_traffic(dom, 'cyl')))
    del cst, X
    torch.cuda.empty_cache()
    return out


def main_cyl(a):
    """BASELINE.json configs[3] / SURVEY.md 8(d) config 4: cylindrical (r, phi, z) = 128 x 256 x 512, dr = dz = 2.5e-4,
    BE, dt = 0.05, RobinR(400, 20), z: neumann0 / robin h = 500, T0 = 20 with the top 16 z-planes at 1000.
    One step = r sweep + phi sweep + z sweep on a field resident in HBM; 16 B/cell/sweep (field in + field out:
    the coefficients are per-index constants).  N > 1: independent replicas (the path does not shard)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    staged = a.transport == 'gloo-staged' and world > 1      # TEST transport: every replica on cuda:0
    local_rank = 0 if staged else int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, 'WORLD_SIZE (%d) != --gpus (%d)' % (world, a.gpus)
    need_gpu(rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        import datetime
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        tmo = datetime.timedelta(seconds=a.pg_timeout)
        if staged:
            dist.init_process_group('gloo', timeout=tmo)
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=tmo)
    col = Collect(dist, world > 1, staged, dev)
    import adi_thermal_fields_amd.adi3d_hip_cyl as cyl
    nr, nphi, nz = 128, 256, 512
    g = cyl.GridCyl(nr, nphi, nz, 2.5e-4, 2 * np.pi / nphi, 2.5e-4, 0.032)
    mat = cyl.Material(7800.0, 490.0, 54.0)
    prm = cyl.Params(0.05, 1.0, "be")
    rr = cyl.RobinR(400.0, 20.0)
    zbc = cyl.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0)
    T0 = np.full((nr, nphi, nz), 20.0); T0[:, :, -16:] = 1000.0
    T = cyl.to_device(T0)
    st = cyl.StagedCylStepper(g, mat, prm, rr, zbc)

    def sync():
        col.barrier()
        torch.cuda.synchronize()
    # The nsub loop of a driver on a resident field (StagedCylStepper.run): the loop owns its field, so the three sweeps
    # run IN PLACE -- every sweep kernel reads only the rows it writes -- and the working set is one 134 MB field, inside
    # the 256 MB Infinity Cache, instead of two (ping-pong buffers: 0.160 ms per step, in place: 0.144).
    X = g.layout.empty(); X.copy_(T.t)
    # the three kernels take ~45 us each, an event record a few: per-sweep events on every 4th step of the timed region
    # only (the others run the same launches without them), so that the events do not set the step time they measure
    sampled = [s_ for s_ in range(a.steps) if s_ % 4 == 0]
    ev = {s_: make_events(4) for s_ in sampled}
    quiet_gc()             # before the warm-up, which must flow straight into the timed loop (see timed_steps)
    for _ in range(a.warmup):
        st._step_inplace(X)
    sync()
    t0 = time.perf_counter()
    for s_ in range(a.steps):
        st._step_inplace(X, events=ev.get(s_))
    sync()
    gc.enable()
    T = cyl.DeviceField(X)
    elapsed = col.max(time.perf_counter() - t0)
    assert bool(torch.isfinite(T.t).all().item())
    if rank != 0:
        dist.destroy_process_group()
        return
    N = nr * nphi * nz
    all_ms = np.array([[ev[s_][i].elapsed_time(ev[s_][i + 1]) for i in range(3)] for s_ in sampled])
    ms = all_ms.mean(axis=0)
    kernels = {}
    for i, nm in enumerate(st.stage_names):
        gbs = st.stage_bytes_per_cell[i] * N / (ms[i] * 1e-3) / 1e9
        kernels[nm] = dict(ms=round(float(ms[i]), 4), ms_median=round(float(np.median(all_ms[:, i])), 4),
                           ms_min=round(float(all_ms[:, i].min()), 4), bytes_per_cell=st.stage_bytes_per_cell[i],
                           achieved_gbs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    dom = max(kernels, key=lambda k: kernels[k]['ms'])
    ms_per_step = elapsed / a.steps * 1e3
    line = dict(metric='adi_cyl_steps_per_sec_128x256x512_fp64', value=round(world * a.steps / elapsed, 3),
                unit='steps/s (128x256x512 cylindrical BE steps, all GPUs)', n_gpus=world, steps=a.steps, warmup=a.warmup,
                ms_per_step=round(ms_per_step, 4), higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64',
                data='synthetic',
                config=dict(workload='cylindrical (r,phi,z) 128x256x512 fp64, BE dt=0.05, RobinR(400,20), z: neumann0 / '
                                     'robin h=500, T0=20 with the top 16 z-planes at 1000 (BASELINE.json configs[3])',
                            cells_per_gpu=N, decomposition='replicas only' if world > 1 else 'none',
                            loop='the three sweeps in place on a resident field (StagedCylStepper.run)'),
                cell_updates_per_s=round(world * N * a.steps / elapsed, 1),
                step_achieved_gbs=round(48.0 * N / (ms_per_step * 1e-3) / 1e9, 1),
                roofline=dict(bound='hbm', kernel=dom, achieved=kernels[dom]['achieved_gbs'], peak=HBM_PEAK_GBS,
                              unit='GB/s', frac=kernels[dom]['frac'], traffic=measured_traffic(dom, 'cyl')),
                kernels=kernels)
    if not a.no_cpu:
        # the NumPy restatement of adi3d_cyl_phi_v3.adi_step (BE) on a bounded sample of the same workload
        from oracle import cyl_oracle as orc
        cn = (128, 256, 512)      # the full workload: one NumPy step takes 15-20 s
        og = orc.GridCyl(cn[0], cn[1], cn[2], 2.5e-4, 2 * np.pi / cn[1], 2.5e-4, cn[0] * 2.5e-4)
        om = orc.Material(7800.0, 490.0, 54.0); op = orc.Params(0.05, 1.0, "be")
        Tc = np.full(cn, 20.0); Tc[:, :, -16:] = 1000.0
        orr = orc.RobinR(400.0, 20.0); oz = orc.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0)
        t0 = time.perf_counter(); ksteps = 1
        for _ in range(ksteps):
            Tc = orc.adi_step(Tc, og, om, op, orr, oz)
        dtc = time.perf_counter() - t0
        line['cpu_baseline'] = dict(value=ksteps * float(np.prod(cn)) / N / dtc, unit='steps/s (128x256x512-cell equivalent)',
                                    cores=1, kind='port',
                                    sample='%dx%dx%d cells x %d steps of the same BE workload, %.1f s, oracle/cyl_oracle.py '
                                           '(NumPy, as the reference is)' % (cn + (ksteps, dtc)))
    emit(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def so_stamp():
    """identity of the built library the numbers come from: adi_build_stamp(), a hash of its sources and compile flags
    that does not depend on where the tree was built"""
    from adi_thermal_fields_amd import _lib
    return _lib.lib.adi_build_stamp().decode()


def measured_traffic(kernel, config='cart'):
    """HBM bytes per launch of `kernel` from the PMC passes kept in profiles/pmc_traffic.json -- only when that file was
    produced with THIS build of the library (it carries the library's stamp); otherwise null"""
    tp = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        d = json.load(open(tp))
        if d.get('library_stamp') != so_stamp():
            return None
        return (d.get(config) or {}).get(kernel)
    except Exception:
        return None


def main(argv=None):
    if argv is None:
        argv = sys.argv[1:]
        if not argv and 'WORLD_SIZE' in os.environ and 'ADI_BENCH_ARGV' in os.environ:
            argv = json.loads(os.environ['ADI_BENCH_ARGV'])      # a rank started by launch_self (see there)
    argv = list(argv)
    a = parse(argv)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started plainly: this process stays off the GPU and starts the ranks as fresh children
        sys.exit(launch_self(a, argv))
    claim_stdout()
    if a.config == 'cyl':
        return main_cyl(a)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    rehearse = a.rehearse_world if (a.rehearse_world > 1 and world == 1 and a.gpus == 1) else 0
    force_dist = a.force_dist and world == 1
    multi = world > 1 or rehearse > 1 or force_dist   # the slab code path
    staged = a.transport == 'gloo-staged' and world > 1      # TEST transport: every rank on cuda:0, payloads through the host
    local_rank = 0 if staged else int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, 'WORLD_SIZE (%d) != --gpus (%d)' % (world, a.gpus)
    need_gpu(rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import datetime
    import torch.distributed as dist
    if world > 1 or force_dist:
        from adi_thermal_fields_amd.dist_slab import rccl_env_defaults
        rccl_env_defaults()                      # dmabuf IPC; RCCL kernels on a hardware queue of their own
        # a collective that hangs (ranks in different interface forms, a lost peer) must end inside the driver's 600 s with a
        # stack trace from the process group's watchdog, not as "killed at limit, wrote nothing"
        tmo = datetime.timedelta(seconds=a.pg_timeout)
        if force_dist:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=tmo)
        elif staged:
            dist.init_process_group('gloo', timeout=tmo)
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=tmo)
    col = Collect(dist, world > 1 or force_dist, staged, dev)

    import adi_thermal_fields_amd.adi3d_hip_coeff as adi
    from adi_thermal_fields_amd import _lib

    n = a.n
    mat = adi.Material(7800.0, 490.0, 54.0)
    alpha = mat.k / (mat.rho * mat.cp)
    dx = 5e-4
    prm = adi.Params(200.0 * dx * dx / alpha, 0.5)
    Tinf = 20.0
    # this rank's slab: planes [i0, i0 + nxl) of a (nxg, n, n) grid
    W, r = (rehearse, rehearse // 2) if rehearse else (world, rank)
    if a.scaling == 'strong' and W > 1:
        from adi_thermal_fields_amd.dist_slab import split_planes
        sizes = split_planes(n, W)
        nxg, nxl, i0 = n, sizes[r], sum(sizes[:r])
    else:
        nxg, nxl, i0 = W * n, n, r * n
    shape = (nxl, n, n)
    mask = make_mask(a.mask, shape, i0, nxg)
    T0_host = None
    if not multi:
        # host field from a NumPy seed: the CPU-baseline leg starts the oracle from the same array (parity_rel_linf)
        T0_host = np.random.default_rng(1).uniform(20.0, 1000.0, shape)
        T = adi.to_device(T0_host)
    else:
        gen = torch.Generator(device=dev); gen.manual_seed(1 + rank)
        T = adi.DeviceField(torch.rand(shape, dtype=torch.float64, device=dev, generator=gen) * 980.0 + 20.0)

    stage_names = ['explicit', 'sweep_axis0', 'sweep_axis1', 'sweep_axis2_contig']
    parity_1d = None
    if not multi:
        grid = adi.Grid3D(n, n, n, dx, mask)
        packs = adi.precompute_coeff_packs_unified(grid, mat, robin_h=500.0)
        stepper = adi.StagedStepper(grid, mat, prm, packs, Tinf)
        stage_names = stepper.stage_names      # explicit stage folded into the axis-0 sweep where supported
        variant = packs[0].variant
    else:
        from adi_thermal_fields_amd import dist_slab

        def make_comm():
            if rehearse:      # a middle rank of `rehearse`: loopback copies, or (--force-dist) RCCL send/recv to itself
                return (dist_slab.SelfLoopDistComm if force_dist else dist_slab.LoopbackComm)(rehearse, rehearse // 2)
            return dist_slab.HostStagedDistComm() if staged else dist_slab.TorchDistComm()
        # Before anything is timed: does this job's decomposition reproduce the one-domain result?  (1) the n^3 grid cut over
        # the ranks (BASELINE.json configs[2]; the strong-scaling workload itself); (2) with weak scaling also this job's own slab
        # thickness -- n planes per rank, hence the interface form of the timed loop -- on a grid of reduced lateral extent
        # (W*n x 128 x 128), which rank 0 can step on one domain in milliseconds.
        if world > 1 or (force_dist and not rehearse):      # (--force-dist: the same calls over a one-rank RCCL group, a plumbing check)
            from adi_thermal_fields_amd.dist_slab import split_planes as _split
            parity_1d = parity_vs_one_domain(a, adi, dist_slab, dist, world, rank, mat, prm, dx, Tinf, make_comm, staged,
                                             _split(n, world), n, n, '%d^3 cut into slabs over the %d ranks (strong split)' % (n, world))
            if a.scaling != 'strong':
                lat = min(n, 128)
                p2 = parity_vs_one_domain(a, adi, dist_slab, dist, world, rank, mat, prm, dx, Tinf, make_comm, staged,
                                          [n] * world, lat, lat, "the timed job's own slab thickness (%d planes per rank, weak "
                                          'scaling) on a grid of reduced lateral extent' % n)
                if parity_1d is not None:
                    parity_1d = dict(parity_1d, weak_form=p2)
        stepper = dist_slab.SlabStepper.from_local(T.t, mask, dx, mat, prm, Tinf, robin_h=500.0, comm=make_comm())
        variant = stepper.variant
        overlap_err, overlap_on = stepper.self_check(T)     # pipeline on the second stream vs plain ordering

    def sync():
        col.barrier()
        torch.cuda.synchronize()

    # General-pack (42 B/cell) sweeps: the reference's own data model with every pack array read in full; the
    # contiguous one is the kernel the 60 % target of BASELINE.json is written against.  Measured with the same event method,
    # outside the timed region -- and BEFORE it: these ~45 ms of GPU work also bring the device to its steady clocks, which the
    # W warm-up steps alone (7 ms at the driver's W = 5) do not (same box, same build: 672 steps/s over 20 steps from a cold
    # start, 695 over 50; the line says so in `prewarm`).
    quiet_gc()             # no cyclic collection from here to the end of the timed region (a collection takes 0.1 s of host time or
    xs = None              # more: inside the loop it is a 40 ms outlier, between the warm-up and the loop it lets the GPU clock down)
    if not multi:
        Nc = nxl * n * n
        out = grid.layout.empty()
        tin = grid.layout.to_layout(T, torch.float64)
        xs = {}
        for ax, nm in ((2, 'sweep_axis2_contig'), (0, 'sweep_axis0'), (1, 'sweep_axis1')):
            ms = []
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            for it in range(13):
                e0.record()
                stepper.sweep_into(ax, tin, out, variant=_lib.SWEEP_GENERAL, dense=True)
                e1.record(); e1.synchronize()
                if it >= 3:
                    ms.append(e0.elapsed_time(e1))
            m = float(np.mean(ms))
            xs[nm] = dict(ms=round(m, 4), bytes_per_cell=42, achieved_gbs=round(42 * Nc / (m * 1e-3) / 1e9, 1),
                          frac=round(42 * Nc / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        del out, tin

    kw = dict(prefetch_halo=True) if multi else {}     # the loop feeds every step's output to the next unmodified
    T_start = T
    for _ in range(a.warmup):
        T = stepper.step(T, **kw)
    if multi:
        if a.warmup == 0:
            T = stepper.step(T, **kw)                      # the plan (and with it the stage list) exists after one step
        stage_names = stepper.stage_names                  # depends on the axis-0 plan chosen for this dt / mask
    nst = len(stage_names)
    ev = [make_events(nst + 1) for _ in range(a.steps)]
    sync()
    t0 = time.perf_counter()
    for s in range(a.steps):
        T = stepper.step(T, events=ev[s], **kw)     # HIP events on the launch stream between the stage kernels
    sync()
    t1 = time.perf_counter()
    gc.enable()
    elapsed = col.max(t1 - t0)
    assert bool(torch.isfinite(T.t).all().item())

    stage_ms = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(nst)] for s in range(a.steps)])
    mean_ms = stage_ms.mean(axis=0)
    N = nxl * n * n                                    # cells of this rank
    bytes_per_cell = dict(zip(stage_names, stepper.stage_bytes_per_cell))
    kernels = {}
    for i, nm in enumerate(stage_names):
        gbs = bytes_per_cell[nm] * N / (mean_ms[i] * 1e-3) / 1e9
        kernels[nm] = dict(ms=round(float(mean_ms[i]), 4), ms_median=round(float(np.median(stage_ms[:, i])), 4),
                           ms_min=round(float(stage_ms[:, i].min()), 4), bytes_per_cell=round(bytes_per_cell[nm], 3),
                           achieved_gbs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))

    # what every rank measured, so that the SCALE record shows what RCCL saw: the world size from the process group itself,
    # every rank's stage times and the payload each rank handed to the transport per step
    ranks_info = None
    total_cells = float(N)
    if world > 1 or force_dist:
        comm = stepper.comm
        allr = col.gather_rows(list(mean_ms) + [float(getattr(comm, 'bytes_sent', 0)), float(getattr(comm, 'n_exchanges', 0)),
                                                float(t1 - t0), float(N)])
        total_cells = float(allr[:, nst + 3].sum())
        nsteps_counted = a.steps + max(a.warmup, 1) + 2            # the counters run from construction (incl. self-check)
        ranks_info = dict(world_size_from_process_group=dist.get_world_size(), backend=dist.get_backend(),
                          stage_ms_per_rank={nm: [round(float(v), 4) for v in allr[:, i]] for i, nm in enumerate(stage_names)},
                          loop_seconds_per_rank=[round(float(v), 4) for v in allr[:, nst + 2]],
                          cells_per_rank=[int(v) for v in allr[:, nst + 3]],
                          mbytes_sent_per_rank_total=[round(float(v) / 1e6, 2) for v in allr[:, nst]],
                          exchanges_per_rank_total=[int(v) for v in allr[:, nst + 1]],
                          transport=('gloo-staged (TEST transport: every rank on cuda:0, payloads through pinned host memory; '
                                     'not a multi-GPU measurement)' if staged else 'nccl (RCCL)'),
                          # collective or point-to-point mesh all-gather, as measured on this node per payload size (dist_slab.TorchDistComm)
                          all_gather={str(k): v for k, v in getattr(comm, 'all_gather_choice', {}).items()},
                          note='byte / exchange counters cover %d steps (warm-up, self-check and timed loop)' % nsteps_counted)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    dom = max(kernels, key=lambda k: kernels[k]['ms'])
    traffic = measured_traffic(dom, 'cart') if a.mask == 'box' else None
    ms_per_step = elapsed / a.steps * 1e3
    ranks_in_job = rehearse if rehearse else world
    if rehearse:
        total_cells = float(N)                   # ONE rank ran: the line reports that rank's rate, not a W-GPU rate
    value = total_cells / 512 ** 3 * a.steps / elapsed
    strong = a.scaling == 'strong' and ranks_in_job > 1
    mask_txt = 'all-solid mask' if a.mask == 'box' else \
        'ellipsoid mask (semi-axes 0.47 / 0.49 / 0.48 of the box, %.1f %% of this rank\'s cells in the mask)' % (100.0 * mask.mean())
    line = dict(
        metric='adi_steps_per_sec_512cubed_fp64', value=round(value, 3),
        unit='steps/s (512^3-cell equivalent, all GPUs)', n_gpus=world, steps=a.steps, warmup=a.warmup,
        ms_per_step=round(ms_per_step, 4), higher_is_better=True, scaling=('strong' if strong else 'weak'),
        vs_baseline=None, dtype='f64', data='synthetic',
        config=dict(workload='%dx%dx%d fp64 Cartesian, Robin h=500 all faces, theta=0.5, cfl=200, %s'
                             % (nxg if multi else n, n, n, mask_txt),
                    cells_per_gpu=N, planes_per_gpu=nxl,
                    decomposition=('slabs along memory axis 0' if multi else 'none'),
                    sweep_variant={0: 'general', 1: 'no_dir', 2: 'no_q', 3: 'lean'}[variant]),
        cell_updates_per_s=round(total_cells * a.steps / elapsed, 1),
        step_achieved_gbs=round(sum(bytes_per_cell.values()) * N / (ms_per_step * 1e-3) / 1e9, 1),
        **({'comm_overlap': dict(enabled=overlap_on, selfcheck_rel_diff=overlap_err,
                                 axis0_interface=stepper.axis0_mode,
                                 pass_a=stepper.pass_a_form)} if multi else {}),
        **({'force_dist': 'slab code path over a single-rank RCCL process group' + (': halo / interface exchanges are RCCL '
                           'send/recv to self on the side stream' if rehearse else ' (plumbing check)')} if force_dist else {}),
        **({'rehearsal': 'ONE GPU running the code path of rank %d of %d (%s scaling, %d planes) with a loopback '
                         'communicator: per-rank compute time without wire time, not a %d-GPU measurement; `value` is this '
                         'one rank\'s rate' % (rehearse // 2, rehearse, a.scaling, nxl, rehearse)}
           if rehearse else {}),
        roofline=dict(bound='hbm', kernel=dom, achieved=kernels[dom]['achieved_gbs'], peak=HBM_PEAK_GBS,
                      unit='GB/s', frac=kernels[dom]['frac'], traffic=traffic),
        kernels=kernels,
        **({'ranks': ranks_info} if ranks_info is not None else {}),
        **({'parity_vs_one_domain': parity_1d} if parity_1d is not None else {}),
    )
    if xs is not None:
        line['general_pack_sweeps_42B'] = xs
        line['prewarm'] = ('the 42 B/cell sweep measurements (3 x 13 launches, about 45 ms of GPU work) run before the warm-up steps: '
                           'the timed region starts on a device at its steady clocks')
        # BASELINE.json's target is written against this kernel: ">= 60 % of MI355X HBM peak on the x-sweep batched
        # Thomas solve for a 512^3 fp64 Cartesian grid at 1 GPU" (x = the contiguous axis), the reference's own data model
        x = xs['sweep_axis2_contig']
        line['north_star_x_sweep'] = dict(kernel='sweep_axis2_contig, general pack (42 B/cell, every array read in full)',
                                          achieved=x['achieved_gbs'], peak=HBM_PEAK_GBS, unit='GB/s', frac=x['frac'],
                                          target_frac=0.60,
                                          lean_sparse_variant_in_step=kernels.get('sweep_axis2_contig'))
    if not multi and not a.no_also:
        line['also'] = also_configs(a, adi, mat, prm, dx, Tinf, dev)
    if not multi and not a.no_cpu:
        # parity of the very kernels that were timed: PARITY_STEPS steps of the timed stepper from T0 (or, with --cpu-n,
        # of the same workload at that edge) against the single-thread oracle leg started from the same host array
        PARITY_STEPS = 3
        cn = a.cpu_n
        if cn == n:
            Tp, pm, T0c = T_start, mask, T0_host
            pstep = stepper
        else:
            pm = make_mask(a.mask, (cn, cn, cn))
            T0c = np.random.default_rng(1).uniform(20.0, 1000.0, (cn, cn, cn))
            g2 = adi.Grid3D(cn, cn, cn, dx, pm)
            pstep = adi.StagedStepper(g2, mat, prm, adi.precompute_coeff_packs_unified(g2, mat, robin_h=500.0), Tinf)
            Tp = adi.to_device(T0c)
        for _ in range(PARITY_STEPS):
            Tp = pstep.step(Tp)
        got = Tp.get()
        del Tp, T, T_start
        torch.cuda.empty_cache()
        st, mt, want = cpu_baseline(cn, pm, T0c, steps_1t=PARITY_STEPS)
        den = float(np.abs(want).max())
        line['parity_rel_linf'] = float(np.abs(got - want).max() / (den if den > 0 else 1.0))
        line['parity'] = dict(rel_linf=line['parity_rel_linf'], bar=1e-10, steps=PARITY_STEPS, cells='%d^3' % cn,
                              against='oracle/adi_oracle.c, 1 thread, same T0 (NumPy default_rng(1)) and mask',
                              stepper='the StagedStepper that ran the timed loop' if cn == n else
                                      'a StagedStepper of the same workload at the --cpu-n edge')
        line['cpu_baseline'] = st
        line['cpu_baseline_all_cores'] = mt
    emit(json.dumps(line))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
