"""dist_slab -- the Cartesian ADI step on a grid cut into slabs along memory axis 0, one slab per GPU.

The reference is single-process (no NCCL/MPI anywhere); this module is the multi-GPU
part of the north star: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Per step and rank (nxl local planes, halo planes at both ends of every extended array):
  1. halo exchange of the state: one (ny, nz) plane to each neighbour (send/recv, 2 MiB at 512^2);
     the mask halo travels only when the mask changes (pack rebuild).
  2. explicit stage on the extended slab (interior result used).
  3. axis-0 sweep, whose lines span all ranks, as a reduced-interface solve:
       pass A  adi_sweep_condense: rows of every local line -> 6 numbers (first/last unknown as affine functions
               of the neighbours' adjacent unknowns),
       exchange + interface solve -> the two boundary values of every line,
       pass B  adi_sweep with the boundary values injected: the ordinary local sweep.
     Two forms of the middle step, chosen per (dt, theta, mask) by looking at the condensed matrix entries:
       neighbour-only ("windowed"): the coupling of a slab's last unknown to the unknown BEFORE the slab (aL), and
               of its first unknown to the one AFTER it (cF), decays like rho^rows (rho < 1 from diagonal
               dominance).  When both are <= 1e-17 on every line -- below the rounding of the data they would
               multiply -- the interface system splits into independent 2x2 systems between neighbours:
               send (gF,aF) down and (gL,aL,cL) up (send/recv, 6.3 + 4.2 MB at 512^2), adi_interface_pair.
               When the decay length K is below half the slab, pass A runs only on the first and last K planes
               (the same argument applied to the window) and its exchange hides behind the explicit stage of
               the middle planes.
       exact:  all_gather of the (6, ny*nz) block (12.6 MB per rank at 512^2) + adi_interface_solve (each rank
               merges the slabs below/above it and solves a 2x2 system per line); thin slabs / huge dt.
     Either way the result equals the single-domain sweep to rounding.
  4. axis-1 and axis-2 sweeps: lines are local, no communication.

The numerical work is behind an `engine` (HIP kernels through the C ABI in production) and the exchange
behind a `comm`, so the algebra of the decomposition can be tested on CPU ranks over gloo with a
reference engine (tests/) and, on one GPU, with several in-process ranks.
"""
import ctypes
import threading

import numpy as np
import torch

__all__ = ['SlabStepper', 'TorchDistComm', 'HostStagedDistComm', 'LocalComm', 'LoopbackComm', 'SelfLoopDistComm', 'HipEngine',
           'split_planes', 'rccl_env_defaults', 'gather_slabs']


def rccl_env_defaults():
    """Environment a rank wants BEFORE `init_process_group('nccl')` (setdefault: the caller's own settings win).
    * HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC.
    * TORCH_NCCL_HIGH_PRIORITY=1: torch runs RCCL kernels on an internal stream; at normal priority HIP may map that
      stream onto the same hardware queue as the stream the sweeps run on (measured: rocprofv3 showed both on one HSA
      queue), and then a halo send/recv posted "beside" a sweep runs after it instead -- nothing overlaps.  High-priority
      streams get queues of their own: per-rank step over the RCCL self-loop 1.65 -> 1.59 ms."""
    import os
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')


def split_planes(nx, world):
    """Even slab sizes (the fast condensation kernel wants whole 2/4/8-row segments)."""
    base = (nx // world) // 2 * 2
    sizes = [base] * world
    rem = nx - base * world
    i = 0
    while rem >= 2:
        sizes[i % world] += 2
        rem -= 2
        i += 1
    sizes[-1] += rem
    assert sum(sizes) == nx and all(s > 0 for s in sizes), (nx, world, sizes)
    return sizes


# ------------------------------------------------------------------------------------------- comm
class TorchDistComm:
    """torch.distributed (nccl = RCCL on ROCm, or gloo on CPU).

    all_gather_mode: how the all-gather of the interface forms that need every rank's planes ('exact', 'deferred_exact': thin
    slabs, strong scaling) travels.  'collective' = `all_gather_into_tensor` (RCCL picks its algorithm, ring-like on most
    topologies: world-1 steps, each bound by ONE link); 'mesh' = every rank sends its block to every other rank in one batch of
    point-to-point operations -- on an xGMI node every pair of GPUs has a link of its own (7 x ~153 GB/s per GPU), so the
    world-1 blocks leave over world-1 links at once; 'auto' (default) = measured: the first time a payload of a size class
    (>= 256 KiB, device tensors, nccl, world >= 3) comes by, both are timed on the spot (a few iterations each, the slower rank
    counts) and the faster one is kept for that size class.  The choice is collective -- it comes out of an all-reduce -- and is
    reported by bench.py in `ranks.all_gather`; nothing here has run on more than one GPU, which is why it is measured, not
    assumed."""

    def __init__(self, group=None, all_gather_mode='auto'):
        import torch.distributed as dist
        assert all_gather_mode in ('auto', 'collective', 'mesh')
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_sent = 0          # payload this rank has handed to the transport (bench.py reports it per step)
        self.n_exchanges = 0
        self.all_gather_mode = all_gather_mode
        self.tune_min_bytes, self.tune_any_backend = 256 << 10, False        # (tests lower / set these to walk the measuring path on gloo)
        self.all_gather_choice = {}  # size class (bytes, rounded up to a power of two) -> dict(mode, collective_ms, mesh_ms)

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        """send_lo -> rank-1 (received there as recv_hi), send_hi -> rank+1 (received there as recv_lo)."""
        dist = self.dist
        ops = []
        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, send_lo, self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.rank - 1, self.group))
            self.bytes_sent += send_lo.numel() * send_lo.element_size()
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, send_hi, self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.rank + 1, self.group))
            self.bytes_sent += send_hi.numel() * send_hi.element_size()
        if ops:
            self.n_exchanges += 1
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def _all_gather_mesh(self, out, inp):
        """every block straight to every peer: world-1 sends and world-1 receives in ONE batch"""
        dist, n = self.dist, inp.numel()
        flat, src = out.view(-1), inp.reshape(-1)
        flat[self.rank * n:(self.rank + 1) * n].copy_(src)
        ops = []
        for d in range(1, self.world):                 # peers in rotated order: no two ranks start on the same target
            r = (self.rank + d) % self.world
            ops.append(dist.P2POp(dist.isend, src, r, self.group))
        for d in range(1, self.world):
            r = (self.rank - d) % self.world
            ops.append(dist.P2POp(dist.irecv, flat[r * n:(r + 1) * n], r, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def _all_gather_pick(self, out, inp):
        """'collective' or 'mesh' for this payload (see the class docstring); measured once per size class when 'auto'"""
        if self.all_gather_mode != 'auto':
            return self.all_gather_mode
        nbytes = inp.numel() * inp.element_size()
        on_rccl = inp.is_cuda and self.dist.get_backend(self.group) == 'nccl'
        if nbytes < self.tune_min_bytes or self.world < 3 or not (on_rccl or self.tune_any_backend):
            return 'collective'
        cls = 1 << (nbytes - 1).bit_length()
        got = self.all_gather_choice.get(cls)
        if got is None:
            dist = self.dist
            times = []
            for fn in (lambda: dist.all_gather_into_tensor(out, inp, group=self.group), lambda: self._all_gather_mesh(out, inp)):
                for _ in range(2):
                    fn()
                if inp.is_cuda:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.current_stream().synchronize()
                    dist.barrier(group=self.group)
                    e0.record()
                    for _ in range(5):
                        fn()
                    e1.record(); e1.synchronize()
                    times.append(e0.elapsed_time(e1) / 5.0)
                else:
                    import time
                    dist.barrier(group=self.group)
                    t0 = time.perf_counter()
                    for _ in range(5):
                        fn()
                    times.append((time.perf_counter() - t0) * 1e3 / 5.0)
            t = torch.tensor(times, dtype=torch.float64, device=inp.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)          # the slowest rank counts; the same on every rank
            tc, tm = float(t[0]), float(t[1])
            got = self.all_gather_choice[cls] = dict(mode='mesh' if tm < 0.9 * tc else 'collective', collective_ms=round(tc, 4),
                                                     mesh_ms=round(tm, 4))
        return got['mode']

    def all_gather(self, out, inp):
        self.bytes_sent += inp.numel() * inp.element_size() * (self.world - 1)
        if self._all_gather_pick(out, inp) == 'mesh':
            self._all_gather_mesh(out, inp)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)


class HostStagedDistComm(TorchDistComm):
    """A TEST transport, not a product path: torch.distributed 'gloo' between processes whose fields live on a GPU.  Every
    payload goes device -> pinned host buffer -> gloo -> pinned host buffer -> device, the copies on the CURRENT stream, so
    the stream / event ordering SlabStepper sets up around an exchange is the one a RCCL run has.  RCCL refuses a second rank
    on the same device, gloo does not care: with this transport several REAL processes -- separate HIP contexts, separate
    allocators, a real rendezvous -- drive HipEngine + SlabStepper on the one GPU of a test box
    (tests/test_dist_hip_processes.py, `bench.py --transport gloo-staged`)."""

    def __init__(self, group=None):
        super().__init__(group)
        self._pin = {}

    def _host(self, tag, t):
        key = (tag, tuple(t.shape), t.dtype)
        h = self._pin.get(key)
        if h is None:
            h = torch.zeros(tuple(t.shape), dtype=t.dtype).pin_memory() if t.is_cuda else torch.zeros(tuple(t.shape), dtype=t.dtype)
            self._pin[key] = h
        return h

    @staticmethod
    def _wait_stream(t):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        dist, ops, back = self.dist, [], []
        if self.rank > 0:
            hs, hr = self._host('s_lo', send_lo), self._host('r_lo', recv_lo)
            hs.copy_(send_lo, non_blocking=True)
            ops += [dist.P2POp(dist.isend, hs, self.rank - 1, self.group), dist.P2POp(dist.irecv, hr, self.rank - 1, self.group)]
            back.append((recv_lo, hr))
            self.bytes_sent += send_lo.numel() * send_lo.element_size()
        if self.rank < self.world - 1:
            hs, hr = self._host('s_hi', send_hi), self._host('r_hi', recv_hi)
            hs.copy_(send_hi, non_blocking=True)
            ops += [dist.P2POp(dist.isend, hs, self.rank + 1, self.group), dist.P2POp(dist.irecv, hr, self.rank + 1, self.group)]
            back.append((recv_hi, hr))
            self.bytes_sent += send_hi.numel() * send_hi.element_size()
        if not ops:
            return
        self.n_exchanges += 1
        self._wait_stream(send_lo)
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for dst, h in back:
            dst.copy_(h, non_blocking=True)
        self._wait_stream(send_lo)       # the pinned buffers are reused by the next exchange, possibly issued on another stream

    def all_gather(self, out, inp):
        self.bytes_sent += inp.numel() * inp.element_size() * (self.world - 1)
        hi = self._host('ag_i', inp.reshape(-1))
        ho = self._host('ag_o', out.reshape(-1))
        hi.copy_(inp.reshape(-1), non_blocking=True)
        self._wait_stream(inp)
        self.dist.all_gather_into_tensor(ho, hi, group=self.group)
        out.view(-1).copy_(ho, non_blocking=True)
        self._wait_stream(inp)


def gather_slabs(local, sizes, group=None, host_staged=False):
    """The whole field on rank 0 (a tensor on `local`'s device; None on the other ranks) from slabs of sizes[r] planes each.
    Point-to-point copies of the right sizes (slabs may differ by two planes).  host_staged: through host memory (gloo)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert len(sizes) == world and local.shape[0] == sizes[rank], (sizes, tuple(local.shape), rank)
    mine = local.contiguous()
    if local.is_cuda:
        torch.cuda.current_stream(local.device).synchronize()
    if rank != 0:
        dist.send(mine.cpu() if host_staged else mine, dst=0, group=group)
        return None
    parts = [mine]
    for r in range(1, world):
        buf = torch.empty((sizes[r],) + tuple(local.shape[1:]), dtype=local.dtype, device='cpu' if host_staged else local.device)
        dist.recv(buf, src=r, group=group)
        parts.append(buf.to(local.device))
    return torch.cat(parts, dim=0)


class LocalComm:
    """Several ranks inside ONE process (one thread per rank): used to run the distributed algorithm on a
    single GPU in tests.  Not a product path."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = {}

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    @staticmethod
    def make(world):
        sh = LocalComm._Shared(world)
        return [LocalComm(sh, r) for r in range(world)]

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.sh.barrier.wait()

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        self.sh.slots[('lo', self.rank)] = send_lo
        self.sh.slots[('hi', self.rank)] = send_hi
        self._sync()
        if self.rank > 0:
            recv_lo.copy_(self.sh.slots[('hi', self.rank - 1)])
        if self.rank < self.world - 1:
            recv_hi.copy_(self.sh.slots[('lo', self.rank + 1)])
        self._sync()

    def all_gather(self, out, inp):
        self.sh.slots[('ag', self.rank)] = inp
        self._sync()
        n = inp.numel()
        flat = out.view(-1)
        for r in range(self.world):
            flat[r * n:(r + 1) * n].copy_(self.sh.slots[('ag', r)].view(-1))
        self._sync()


class LoopbackComm:
    """Rehearsal on ONE GPU: this rank talks to copies of itself, so a step executes every kernel and host call a
    rank of a `world`-GPU run executes, with no wire time (scripts/dist_probe.py, bench.py --rehearse-world).
    Not a product path."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        if self.rank > 0:
            recv_lo.copy_(send_hi)          # what a copy of this rank sitting below would send up
        if self.rank < self.world - 1:
            recv_hi.copy_(send_lo)

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))


class SelfLoopDistComm(TorchDistComm):
    """Rehearsal on ONE GPU over the REAL transport: a single-rank RCCL process group in which this rank plays rank
    `rank` of `world` and both its neighbours are itself (RCCL accepts batched send/recv to self).  Same data movement as
    LoopbackComm, but through torch.distributed P2P on the side stream, so the stream / event ordering of SlabStepper is
    exercised exactly as in a multi-GPU run (tests/test_dist_slab_gpu.py, bench.py --rehearse-world W --force-dist)."""

    def __init__(self, world, rank, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        assert dist.get_world_size(group) == 1
        self.me = dist.get_rank(group)
        self.rank, self.world = rank, world
        self.bytes_sent = 0
        self.n_exchanges = 0

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        dist, ops = self.dist, []
        if self.rank > 0:
            self.bytes_sent += send_hi.numel() * send_hi.element_size()
        if self.rank < self.world - 1:
            self.bytes_sent += send_lo.numel() * send_lo.element_size()
        # sends and receives to one peer match in order: what goes up (send_hi) must land in recv_lo, and vice versa
        if self.rank > 0 and self.rank < self.world - 1:
            ops = [dist.P2POp(dist.isend, send_hi, self.me, self.group), dist.P2POp(dist.irecv, recv_lo, self.me, self.group),
                   dist.P2POp(dist.isend, send_lo, self.me, self.group), dist.P2POp(dist.irecv, recv_hi, self.me, self.group)]
        elif self.rank > 0:
            ops = [dist.P2POp(dist.isend, send_hi, self.me, self.group), dist.P2POp(dist.irecv, recv_lo, self.me, self.group)]
        elif self.rank < self.world - 1:
            ops = [dist.P2POp(dist.isend, send_lo, self.me, self.group), dist.P2POp(dist.irecv, recv_hi, self.me, self.group)]
        if ops:
            self.n_exchanges += 1
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def all_gather(self, out, inp):
        self.bytes_sent += inp.numel() * inp.element_size() * (self.world - 1)
        one = torch.empty_like(inp)
        self.dist.all_gather_into_tensor(one, inp, group=self.group)          # the real collective, world size 1
        out.view(self.world, -1).copy_(one.view(1, -1).expand(self.world, -1))


# ------------------------------------------------------------------------------------------ engine
class HipEngine:
    """The product engine: hand-written HIP kernels through the C ABI (include/adi_hip.h)."""

    def __init__(self):
        from . import adi3d_hip_coeff as hip
        from . import _lib
        self.hip, self._lib, self.lib, self.check = hip, _lib, _lib.lib, _lib.check
        self.device = hip._device()
        self.box_hint = 0          # 2: every cell of every slab is in the mask (SlabStepper.set_mask decides, collectively)
        self.mask_epoch = 0        # bumped by SlabStepper.set_mask: the flags / packs are rebuilt in place
        self._nofb = {}            # no-fallback promise per sweep configuration (bit 2 of `sparse`, include/adi_hip.h)
        self._fconsts = {}         # coefficient storage -> per-face scalars of the pack built on it (h_face_consts)
        self._stream_ptr = None    # launch stream of the step in progress (SlabStepper.step pins it: one lookup per step)
        self._wb = {}              # workspace bytes per box shape

    def _sp(self):
        """the hipStream_t every kernel of this engine is launched on: torch's current stream, looked up once per step"""
        return self._stream_ptr if self._stream_ptr is not None else self.hip._stream()

    def plane_dims(self, ny, nz):
        """physical (ny, nz) of a slab's planes: adi_recommended_dims for a box with long lines along axis 0 (whatever the slab
        thickness of this rank: all ranks must agree)"""
        return tuple(self.hip.recommended_dims(64, ny, nz)[1:])

    def layout(self, nx, ny, nz, sx=None):
        # no padding at this level (Layout's default for whole grids): the stepper pads the planes of a slab itself
        # (SlabStepper: plane_dims), and its extended arrays, halo planes and sub-boxes all share one plane stride
        return self.hip.Layout(nx, ny, nz, sx, phys=(nx, ny, nz))

    def vec(self, n):
        """plan-time fp64 buffer (receive buffers, interface values, flags): ZERO-filled, so a protocol slip shows as a wrong
        number and never as a read of uninitialised memory (round 3's one GPU memory fault: 0 / 1 plane flags built on an
        uninitialised vec()).  Never called per step."""
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    def build_flags(self, L, mask_ext):
        flags = L.empty(torch.uint8, zero=True)
        self.check(self.lib.adi_build_nbr_flags(self.hip._p(mask_ext), L.nx, L.ny, L.nz, L.sx, self.hip._p(flags),
                                                self._sp()))
        return flags

    def _fc(self, pack):
        """h_face_consts of a pack tuple (coeff, dir_mask, dir_val, qflux) or of any view cut out of it: the per-face scalars
        the pack was built from (registered by build_packs under the coefficient array's storage), or None"""
        return self._fconsts.get(pack[0].untyped_storage().data_ptr())

    def build_packs(self, L, mask_ext, flags_ext, dx, mat, dir_mask, dir_value, neumann, robin_h):
        """precompute_coeff_packs_unified on the extended slab; returns packs whose arrays are extended too."""
        g = self.hip.Grid3D.__new__(self.hip.Grid3D)
        g.nx, g.ny, g.nz, g.dx, g.layout = L.nx, L.ny, L.nz, float(dx), L
        g._mask, g._d_mask, g._d_flags, g._scratch, g.mask_version = None, mask_ext, flags_ext, None, next(self.hip._MASK_VERSIONS)
        g.sync_mask = lambda: mask_ext            # the device mask (with halos) is authoritative here
        packs = self.hip.precompute_coeff_packs_unified(g, mat, dir_mask=dir_mask, dir_value=dir_value,
                                                        neumann=neumann, robin_h=robin_h)
        self._fconsts = {p.d_coeff.untyped_storage().data_ptr(): p.face_consts for p in packs if p.face_consts is not None}
        return packs

    # in-place updates after a layer birth (planes [k0, k1) of axis 2): what Grid3D.set_mask_device / BirthPacks.update do
    # for one domain, here on the extended arrays of a slab
    def build_flags_planes(self, L, mask_ext, flags_ext, k0, k1):
        self.check(self.lib.adi_build_nbr_flags_planes(self.hip._p(mask_ext), L.nx, L.ny, L.nz, L.sx, self.hip._p(flags_ext),
                                                       int(k0), int(k1), self._sp()))

    def build_packs_planes(self, L, mask_ext, packs_ext, dx, mat, specs, k0, k1):
        h = self.hip
        hm, hs, qm, qs = specs
        none6 = h.ptr_array([None] * 6)
        self.check(self.lib.adi_build_coeffs_planes(h._p(mask_ext), L.nx, L.ny, L.nz, L.sx, float(dx), mat.rho, mat.cp,
                                                    hm, hs, none6, qm, qs, none6,
                                                    h.ptr_array([p.d_coeff.data_ptr() for p in packs_ext]),
                                                    h.ptr_array([p.d_qflux.data_ptr() for p in packs_ext]),
                                                    int(k0), int(k1), self._sp()))

    def birth_planes(self, Li, T_int, act_int, full_int, k0, k1, Ts, count):
        """adi_birth_planes on the interior of a slab (pointers one plane into the extended arrays)"""
        h = self.hip
        self.check(self.lib.adi_birth_planes(h._p(T_int), h._p(act_int), h._p(full_int), Li.nx, Li.ny, Li.nz, Li.sx,
                                             int(k0), int(k1), float(Ts), h._p(count), self._sp()))

    def explicit(self, L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin=0, i_end=None):
        h = self.hip
        i_end = L.nx if i_end is None else i_end
        self.check(self.lib.adi_explicit_rhs_planes(h._p(T_ext), h._p(flags_ext), L.nx, L.ny, L.nz, L.sx, dx, dt, kappa,
                                                    theta, h._p(out_ext), i_begin, i_end, self._sp()))

    def _args(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf):
        h = self.hip
        return (axis, variant, h._p(t_in), h._p(flags), h._p(pack[0]), h._p(pack[1]), h._p(pack[2]), h._p(pack[3]),
                Li.nx, Li.ny, Li.nz, Li.sx, 1 | self.box_hint, theta, gam, dt, float(Tinf))   # packs from adi_build_coeffs: sparse

    def _workspace(self, Li):
        """unit queue of the FAST/GENERAL kernel pair (sized for the largest box seen)"""
        key = (Li.nx, Li.ny, Li.nz, Li.sx)
        need = self._wb.get(key)
        if need is None:
            need = 0
            for ax in range(3):
                b = ctypes.c_size_t(0)
                self.check(self.lib.adi_sweep_workspace_bytes(ax, Li.nx, Li.ny, Li.nz, Li.sx, ctypes.byref(b)))
                need = max(need, b.value)
            self._wb[key] = need
        if getattr(self, '_work', None) is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._work

    def _promise(self, entry, axis, variant, Li, flags, pack):
        """(key, bit): which units the FAST kernels queue depends on the flags, the Dirichlet mask, the variant and the shape
        only; a configuration seen to queue nothing skips the queue reset and the GENERAL launch from then on"""
        key = (entry, axis, variant, Li.nx, Li.ny, Li.nz, Li.sx, flags.data_ptr(), pack[0].data_ptr(),
               None if pack[1] is None else pack[1].data_ptr(), self.box_hint, self.mask_epoch)
        return key, (4 if self._nofb.get(key) is True else 0)

    def _learn(self, key, w):
        st = self._nofb.get(key, 0)
        if st is True or st is False:
            return
        st += 1                    # the read-back synchronises the host: wait for the third sweep of a configuration
        if st >= 3 and not torch.cuda.is_current_stream_capturing():
            st = int(w[:4].view(torch.int32)[0].item()) == 0
        self._nofb[key] = st

    def sweep(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, t_out, xlo=None, xhi=None):
        h = self.hip
        w = self._workspace(Li)
        key, bit = self._promise('sweep', axis, variant, Li, flags, pack)
        a = list(self._args(axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf))
        a[12] |= bit
        self.check(self.lib.adi_sweep(*a, h._p(t_out), h._p(xlo), h._p(xhi), self._fc(pack), h._p(w), w.numel(), self._sp()))
        self._learn(key, w)

    def condense(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, cond):
        h = self.hip
        w = self._workspace(Li)
        self.check(self.lib.adi_sweep_condense(*self._args(axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf),
                                               h._p(cond), self._fc(pack), h._p(w), w.numel(), self._sp()))

    # explicit stage folded into the axis-0 sweep / condensation (ABI v7).  The box (L.nx, L.ny, L.nz) starts at plane
    # i0, row j0 of the extended state T_ext; neighbours outside the box are read from T_ext itself.
    def fused_supported(self, nx, ny, nz, sx, cond_pass):
        return bool(self.lib.adi_explicit_fused_supported(nx, ny, nz, sx, 1 if cond_pass else 0))

    def _fused_args(self, variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf):
        h = self.hip
        tv = T_ext[i0:i0 + L.nx, j0:j0 + L.ny, :]
        vlo, vhi = h.valid_range(tv)
        return (variant, h._p(tv), vlo, vhi, h._p(flags), h._p(pack[0]), h._p(pack[1]), h._p(pack[2]), h._p(pack[3]),
                L.nx, L.ny, L.nz, L.sx, 1 | self.box_hint, dx, dt, kappa, theta, float(Tinf))

    def sweep0_fused(self, variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf, t_out, xlo=None, xhi=None):
        h = self.hip
        w = self._workspace(L)
        key, bit = self._promise('fused', 0, variant, L, flags, pack)
        a = list(self._fused_args(variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf))
        a[13] |= bit
        self.check(self.lib.adi_explicit_sweep0(*a, h._p(t_out), h._p(xlo), h._p(xhi), self._fc(pack), h._p(w), w.numel(),
                                                self._sp()))
        self._learn(key, w)

    # deferred form of the sharded-axis sweep (include/adi_hip.h, ABI v12): every line solved with zero boundary values by
    # the single-domain kernel, one plane to each neighbour, 2 x 2 interface systems, and the rank-two correction added by
    # the axis-1 sweep to what it loads
    def lines_all_uniform(self, Li, flags_int, dmask_int):
        """every sharded-axis line of the slab is solid and free of Dirichlet cells (reads one int: synchronises)"""
        h = self.hip
        nl = Li.ny * Li.nz
        cls = torch.empty(nl, dtype=torch.uint8, device=self.device)
        lst = torch.empty(nl + 1, dtype=torch.int32, device=self.device)
        self.check(self.lib.adi_axis0_classify(h._p(flags_int), h._p(dmask_int), Li.nx, Li.ny, Li.nz, Li.sx, h._p(cls),
                                               h._p(lst), self._sp()))
        return int(lst[0].item()) == 0

    def deferred_setup(self, n, theta, gam, tol):
        """-> dict(w: device weights with exact zeros beyond their reach, omega = w[0], reach)"""
        h = self.hip
        w = self.vec(n)
        om, reach = ctypes.c_double(0.0), ctypes.c_int(0)
        self.check(self.lib.adi_axis0_deferred_setup(n, theta, gam, tol, h._p(w), ctypes.byref(om), ctypes.byref(reach),
                                                     self._sp()))
        return dict(w=w, omega=om.value, reach=reach.value)

    def deferred_exact_setup(self, Li, flags_int, pack, theta, gam, dt, dfr, mat, scal):
        """once per plan: this rank's matrix entries mat [4][nlines] = (aF, cF, aL, cL), its (w[0], w[n-1]) in `scal`, and the
        per-line Sherman-Morrison factors of a global end row (first / last rank) -> opaque handle for deferred_exact_coef"""
        h = self.hip
        nl = Li.ny * Li.nz
        kap = self.vec(2 * nl)
        w = dfr['w']
        w0, wn = float(w[0].item()), float(w[Li.nx - 1].item())
        scal[0] = w0; scal[1] = wn
        self.check(self.lib.adi_deferred_exact_setup(h._p(flags_int[0]), h._p(flags_int[Li.nx - 1]), h._p(pack[0][0]),
                                                     h._p(pack[0][Li.nx - 1]), theta, gam, dt, w0, wn, nl, h._p(mat), h._p(kap),
                                                     self._sp()))
        return dict(kap=kap)

    def interface_uniform(self, g_all, mat_all, scal_all, world, rank, nlines, xlo, xhi):
        h = self.hip
        self.check(self.lib.adi_interface_solve_uniform(h._p(g_all), h._p(mat_all), h._p(scal_all), world, rank, nlines,
                                                        h._p(xlo), h._p(xhi), self._sp()))

    def deferred_exact_coef(self, dx_, xlo, xhi, nlines, clo, chi):
        h = self.hip
        self.check(self.lib.adi_deferred_exact_coef(h._p(xlo), h._p(xhi), h._p(dx_['kap']), nlines, h._p(clo), h._p(chi),
                                                    self._sp()))

    def interface_deferred(self, first, last, prev_last, next_first, omega, nlines, ulo, uhi):
        h = self.hip
        self.check(self.lib.adi_interface_deferred(h._p(first), h._p(last), h._p(prev_last), h._p(next_first), omega,
                                                   nlines, h._p(ulo), h._p(uhi), self._sp()))

    def sweep_corrected(self, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, t_out, ulo, uhi, w_corr):
        """axis-1 sweep of t_in + w[i] * ulo + w[n-1-i] * uhi (the correction is added to what the sweep loads)"""
        h = self.hip
        w = self._workspace(Li)
        key, bit = self._promise('sweep', 1, variant, Li, flags, pack)
        a = list(self._args(1, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf))
        a[12] |= bit
        self.check(self.lib.adi_sweep_corrected(*a[1:], h._p(t_out), h._p(ulo), h._p(uhi), h._p(w_corr), self._fc(pack),
                                                h._p(w), w.numel(), self._sp()))
        self._learn(key, w)

    # the deferred form for lines that are not uniform (ABI v17): per-line homogeneous solutions
    def homogeneous_solution(self, variant, Li, flags, pack, theta, gam, dt, lower):
        """w_lo (lower=True: unit value of the unknown below the slab) or w_hi of every sharded-axis line: the ordinary axis-0
        sweep of a zero field with Tinf = 0, no fluxes, zero Dirichlet values and d_xlo (d_xhi) = 1 -> (nx, ny, nz) tensor"""
        from . import _lib
        nl = Li.ny * Li.nz
        has_dir = variant in (_lib.SWEEP_GENERAL, _lib.SWEEP_NO_Q)
        v = _lib.SWEEP_NO_Q if has_dir else _lib.SWEEP_LEAN
        zero = Li.empty(zero=True)
        one = torch.ones(nl, dtype=torch.float64, device=self.device)
        pk = (pack[0], pack[1], (Li.empty(zero=True) if has_dir else None), None)
        out = Li.empty()
        # (dense reads, no face constants: the coefficient arrays as they are; a one-off per plan)
        h = self.hip
        w = self._workspace(Li)
        a = list(self._args(0, v, Li, zero, flags, pk, theta, gam, dt, 0.0))
        a[12] = 0
        self.check(self.lib.adi_sweep(*a, h._p(out), h._p(one if lower else None), h._p(None if lower else one), None,
                                      h._p(w), w.numel(), self._sp()))
        return out

    def interface_deferred_lines(self, first, last, prev_last, next_first, om, nlines, ulo, uhi, uni_lo=None, uni_hi=None,
                                 ulo_uni=None, uhi_uni=None):
        """ulo / uhi: the interface values of every line; ulo_uni / uhi_uni: the same on the lines uni_lo / uni_hi mark
        (one byte per line) and 0 elsewhere -- what the scalar weights of adi_sweep_corrected multiply"""
        h = self.hip
        self.check(self.lib.adi_interface_deferred_lines(h._p(first), h._p(last), h._p(prev_last), h._p(next_first),
                                                         h._p(om.get('lo_own')), h._p(om.get('hi_prev')), h._p(om.get('hi_own')),
                                                         h._p(om.get('lo_next')), nlines, h._p(ulo), h._p(uhi), h._p(uni_lo),
                                                         h._p(uni_hi), h._p(ulo_uni), h._p(uhi_uni), self._sp()))

    def deferred_lines_apply(self, Li, x, cells, wc, u, from_high_end, nrows=None):
        """x[i][cells[q]] += wc[r][q] * u[cells[q]], i = r or nx-1-r: the flagged lines of one side get their own weights
        (nrows[q]: rows of line q that can carry a non-zero weight; the rest is not read)"""
        h = self.hip
        if cells.numel() == 0:
            return
        self.check(self.lib.adi_deferred_lines_apply(h._p(x), Li.nx, Li.sx, Li.ny * Li.nz, h._p(cells), cells.numel(), h._p(wc),
                                                     int(wc.shape[0]), h._p(u), 1 if from_high_end else 0, h._p(nrows), self._sp()))


    def condense0_fused(self, variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf, cond, r0_out=None):
        """r0_out (optional, a view of the box in an array laid out like T_ext): also receives R0"""
        h = self.hip
        w = self._workspace(L)
        self.check(self.lib.adi_explicit_condense0(*self._fused_args(variant, L, T_ext, i0, j0, flags, pack, dx, dt,
                                                                     kappa, theta, Tinf),
                                                   h._p(cond), h._p(r0_out), self._fc(pack), h._p(w), w.numel(), self._sp()))

    # pass A folded into the marching explicit kernel: dot products of R0 with fixed weights (uniform lines), the rest
    # condensed from the stored R0 (include/adi_hip.h, adi_axis0_dots_*)
    def dots_supported(self, nxl, ny, nz, sx):
        return bool(self.lib.adi_axis0_dots_supported(nxl, ny, nz, sx))

    def dots_setup(self, Li, flags_int, dmask_int, theta, gam):
        """once per (dt, mask): weights, line classes, partial-sum buffer"""
        h = self.hip
        pb, lb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.check(self.lib.adi_axis0_dots_workspace(Li.nx, Li.ny, Li.nz, ctypes.byref(pb), ctypes.byref(lb)))
        dd = dict(weights=self.vec(Li.nx), part=self.vec(pb.value // 8),
                  cls=torch.empty(Li.ny * Li.nz, dtype=torch.uint8, device=self.device),
                  list=torch.empty(lb.value // 4, dtype=torch.int32, device=self.device))
        self.check(self.lib.adi_axis0_dots_setup(Li.nx, theta, gam, h._p(dd['weights']), self._sp()))
        self.check(self.lib.adi_axis0_classify(h._p(flags_int), h._p(dmask_int), Li.nx, Li.ny, Li.nz, Li.sx,
                                               h._p(dd['cls']), h._p(dd['list']), self._sp()))
        return dd

    def dots_nonuniform_fraction(self, dd):
        """share of the lines adi_axis0_classify left to the per-line condensation (reads one int: synchronises)"""
        return int(dd['list'][0].item()) / float(dd['cls'].numel())

    def dots_ichunk(self, n_line):
        return int(self.lib.adi_axis0_dots_ichunk(n_line))

    def explicit_dots(self, L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin, i_end, dd, i_org, n_line):
        """planes [i_begin, i_end) of lines that occupy planes [i_org, i_org + n_line): whole chunks of dots_ichunk(n_line)"""
        h = self.hip
        self.check(self.lib.adi_explicit_rhs_dots(h._p(T_ext), h._p(flags_ext), L.nx, L.ny, L.nz, L.sx, dx, dt, kappa,
                                                  theta, h._p(out_ext), i_begin, i_end, i_org, n_line,
                                                  h._p(dd['weights']), h._p(dd['part']), self._sp()))

    def dots_finish(self, variant, Li, dd, r0, flags, pack, theta, gam, dt, Tinf, line_begin, line_end, cond):
        h = self.hip
        self.check(self.lib.adi_axis0_dots_finish(variant, h._p(dd['part']), h._p(dd['weights']), h._p(dd['cls']),
                                                  h._p(dd['list']), h._p(r0), h._p(flags), h._p(pack[0]), h._p(pack[1]),
                                                  h._p(pack[2]), h._p(pack[3]), Li.nx, Li.ny, Li.nz, Li.sx, theta, gam, dt,
                                                  float(Tinf), line_begin, line_end, h._p(cond), self._sp()))

    def interface(self, cond_all, world, rank, nlines, xlo, xhi):
        h = self.hip
        self.check(self.lib.adi_interface_solve(h._p(cond_all), world, rank, nlines, h._p(xlo), h._p(xhi),
                                                self._sp()))


    def interface_pair(self, my_lo, my_hi, prev_hi, next_lo, nlines, xlo, xhi):
        h = self.hip
        self.check(self.lib.adi_interface_pair(h._p(my_lo), h._p(my_hi), h._p(prev_hi), h._p(next_lo), nlines,
                                               h._p(xlo), h._p(xhi), self._sp()))


def _interior(t_ext):
    """planes 1..n-2 of an extended array: same strides, pointer advanced by one plane"""
    return t_ext[1:-1]


class SlabStepper:
    """One rank's slab.  step(T) takes / returns the LOCAL (nxl, ny, nz) field (DeviceField, torch tensor or
    NumPy array); the state stays in HBM when fed back what step() returned."""

    _comm_priority = -1           # the side stream is a high-priority one: a HSA queue of its own (see rccl_env_defaults)

    def __init__(self, mask_local, dx, mat, params, Tinf=0.0, dir_mask=None, dir_value=None, neumann=None,
                 robin_h=None, comm=None, engine=None):
        self.engine = engine or HipEngine()
        self.comm = comm or TorchDistComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.mat, self.params, self.Tinf, self.dx = mat, params, float(Tinf), float(dx)
        self._bc = dict(dir_mask=dir_mask, dir_value=dir_value, neumann=neumann, robin_h=robin_h)
        mask_local = np.asarray(mask_local).astype(np.bool_)
        E = self.engine
        # Padded planes: a ragged (ny, nz) would send every strided line to the GENERAL kernels (adi_recommended_dims, DESIGN.md
        # section 2), so every internal array of the slab has planes of (self.ny, self.nz) >= the caller's (self.lny, self.lnz);
        # the cells beyond are off-mask, the planes that travel are the physical ones, and step() / set_mask() take and return
        # the logical box.  The extents depend on (ny, nz) alone: every rank picks the same.
        self.nxl, self.lny, self.lnz = mask_local.shape
        self.ny, self.nz = E.plane_dims(self.lny, self.lnz) if hasattr(E, 'plane_dims') else (self.lny, self.lnz)
        self._padded = (self.ny, self.nz) != (self.lny, self.lnz)
        self.Lext = E.layout(self.nxl + 2, self.ny, self.nz)
        self.Lint = E.layout(self.nxl, self.ny, self.nz)
        assert self.Lint.sx == self.Lext.sx
        self.nlines = self.ny * self.nz
        # zero-filled once: the outer halo planes of the first / last rank and the padded tail of every plane are never
        # written; the kernels read them only through selects today, and must not meet NaNs if that ever changes
        self._ext_bufs = [self.Lext.empty(zero=True), self.Lext.empty(zero=True)]
        self._cur = 0
        self._tmp = [self.Lext.empty(zero=True), self.Lext.empty(zero=True)]
        self._mask_version = 0
        self._a0_key, self._a0, self.axis0_mode = None, None, None
        self._no_overlap, self._force_exact = False, False
        self._allow_window = True                  # False keeps 'window' plans off (whole-slab condensation only)
        self._allow_fused = True                   # False: explicit stage as its own kernel (R0 through HBM)
        self.fused = False
        self._keep_r0 = True                       # False: pass B re-evaluates the explicit stage instead of reading R0
        self._allow_dots = True                    # False: pass A as its own kernel (reads the slab a second time)
        self._allow_deferred = True                # False: never the deferred form (zero-boundary solve + correction on load)
        self._allow_deferred_exact = True          # False: thin slabs (no decay) keep the two-pass all-gather form
        self._allow_deferred_lines = True          # False: lines that are not uniform keep the two-pass forms
        self._deferred_lines_cost_ratio = 1.0      # ... and so do slabs whose flagged lines cost more than this x pass A (tests: inf)
        self._plan_steps = None                    # steps taken under the current axis-0 plan (None: no plan yet)
        self._allow_quick_replan = True            # False: every plan is made by the full (measuring) planner
        self._dots_sticky, self._nx_all, self._fused_by_K, self._chunk_cache = None, None, {}, {}
        self._send_g_only = True                   # False: every step exchanges the matrix parts of the interface too
        self._comm_stream, self._use_streams = None, False
        self._halo_ready, self._halo_event = None, None
        self._gam = 0.0
        self.set_mask(mask_local)

    def _pad(self, a, fill=0):
        """host array over the caller's planes (n, lny, lnz) -> the slab's physical planes (n, ny, nz); scalars / None as they are"""
        if not self._padded or a is None or np.isscalar(a):
            return a
        a = np.asarray(a)
        assert a.shape[1:] == (self.lny, self.lnz), (a.shape, (self.lny, self.lnz))
        out = np.full((a.shape[0], self.ny, self.nz), fill, dtype=a.dtype)
        out[:, :self.lny, :self.lnz] = a
        return out

    def _logical(self, t):
        """the caller's box of an internal (n, ny, nz) tensor (a view)"""
        return t[:, :self.lny, :self.lnz] if self._padded else t

    @classmethod
    def from_local(cls, T0_local, mask_local, dx, mat, params, Tinf, **kw):
        """convenience for bench.py: stepper for this rank's slab (T0 only fixes nothing here; the field is
        passed to step())."""
        return cls(mask_local, dx, mat, params, Tinf, **kw)

    # -- mask / packs (rebuilt together, like the reference's drivers do after every birth) ----------
    def set_mask(self, mask_local):
        """grid.mask = ...; packs = precompute_coeff_packs_unified(...) for this slab.  Exchanges the mask
        halo planes, rebuilds the neighbour flags and the coefficient packs on the extended slab."""
        E, L = self.engine, self.Lext
        self._mask_version += 1
        if hasattr(E, 'mask_epoch'):
            E.mask_epoch += 1                      # what the engine learnt about empty unit queues belongs to the old mask
            E._nofb.clear()
        solid_logical = bool(np.asarray(mask_local).all())
        mask_local = self._pad(np.asarray(mask_local).astype(np.bool_), False)
        m_ext = np.zeros((self.nxl + 2, self.ny, self.nz), dtype=np.bool_)
        m_ext[1:-1] = mask_local
        d_mask = L.to_layout(m_ext, torch.uint8)
        lo = d_mask[0].contiguous(); hi = d_mask[-1].contiguous()
        self.comm.exchange_planes(d_mask[1].contiguous(), d_mask[-2].contiguous(), lo, hi)
        if self.rank > 0:
            d_mask[0].copy_(lo)
        if self.rank < self.world - 1:
            d_mask[-1].copy_(hi)
        self.d_mask_ext = d_mask
        if hasattr(E, 'box_hint'):                 # all-solid on every rank -> the kernels' leaner build (a hint only)
            f = E.vec(1); f.fill_(1.0 if solid_logical else 0.0)
            allf = E.vec(self.world)
            self.comm.all_gather(allf, f)
            self._solid_everywhere = float(allf.min()) >= 1.0          # the caller's box, on every rank
            E.box_hint = 2 if (self._solid_everywhere and not self._padded) else 0   # (the kernels' hint is about the physical box)
        self.flags_ext = E.build_flags(L, d_mask)

        def ext(a, fill):
            if a is None or np.isscalar(a):
                return a
            e = np.full((self.nxl + 2, self.ny, self.nz), fill, dtype=np.asarray(a).dtype)
            e[1:-1] = self._pad(a, fill)
            return e
        bc = self._bc
        neumann = None if bc['neumann'] is None else {f: ext(v, 0.0) for f, v in bc['neumann'].items()}
        robin_h = bc['robin_h']
        if isinstance(robin_h, dict):
            robin_h = {f: ext(v, 0.0) for f, v in robin_h.items()}
        else:
            robin_h = ext(robin_h, 0.0)
        self.packs_ext = E.build_packs(L, d_mask, self.flags_ext, self.dx, self.mat, ext(bc['dir_mask'], False),
                                       ext(bc['dir_value'], 0.0), neumann, robin_h)
        self.variant = self.packs_ext[0].variant
        from . import _lib
        self._bpc = [getattr(p, 'bytes_per_cell', float(_lib.SWEEP_BYTES_PER_CELL[self.variant])) for p in self.packs_ext]
        self._explicit_bpc = float(_lib.EXPLICIT_BYTES_PER_CELL)

        def interior_pack(p):
            return tuple(None if t is None else _interior(t) for t in (p.d_coeff, p.d_dir_mask, p.d_dir_val, p.d_qflux))
        self.packs_int = [interior_pack(p) for p in self.packs_ext]
        self.flags_int = _interior(self.flags_ext)

    def _scalar_specs(self):
        """(h_mode, h_scalar, q_mode, q_scalar) ctypes arrays when every face specification is a scalar or absent, else None"""
        import ctypes
        from ._lib import FACES, FACE_NONE, FACE_SCALAR
        bc = self._bc
        if bc['dir_mask'] is not None:
            return None
        hm, hs, qm, qs = [], [], [], []
        for f in FACES:
            rh = bc['robin_h']
            v = None if rh is None else (rh.get(f, 0.0) if isinstance(rh, dict) else rh)
            q = None if bc['neumann'] is None else bc['neumann'].get(f)
            for val, m_, s_ in ((v, hm, hs), (q, qm, qs)):
                if val is None:
                    m_.append(FACE_NONE); s_.append(0.0)
                elif np.isscalar(val):
                    m_.append(FACE_SCALAR); s_.append(float(val))
                else:
                    return None
        return ((ctypes.c_int * 6)(*hm), (ctypes.c_double * 6)(*hs), (ctypes.c_int * 6)(*qm), (ctypes.c_double * 6)(*qs))

    def device_births_supported(self):
        """layer births can be applied to the device mask in place (set_mask_device): the product engine, face
        specifications that are scalars, no Dirichlet cells"""
        return hasattr(self.engine, 'build_flags_planes') and self._scalar_specs() is not None

    def set_mask_device(self, k_begin, k_end):
        """`grid.mask = ...; packs = precompute_...` after a layer birth that switched cells on IN the device mask of the slab
        (the interior planes of self.d_mask_ext, planes [k_begin, k_end) of axis 2; waam.run_layer_birth_slab): the mask halo
        planes travel, the flags and the six coefficient arrays are rebuilt in place on the planes whose exposure can have
        changed -- no host copy of the mask, no allocation.  Collective (every rank calls it for every birth)."""
        E, L = self.engine, self.Lext
        specs = self._scalar_specs()
        assert specs is not None and hasattr(E, 'build_flags_planes')
        self._mask_version += 1
        E.mask_epoch += 1
        E._nofb.clear()
        d_mask = self.d_mask_ext
        self.comm.exchange_planes(d_mask[1], d_mask[-2], d_mask[0], d_mask[-1])     # (no neighbour: the halo plane stays empty)
        k0, k1 = max(0, int(k_begin) - 1), min(self.nz, int(k_end) + 1)
        E.build_flags_planes(L, d_mask, self.flags_ext, k0, k1)
        E.build_packs_planes(L, d_mask, self.packs_ext, self.dx, self.mat, specs, k0, k1)
        E.box_hint = 0                 # a part that is still growing is not an all-solid box
        self._solid_everywhere = False

    @property
    def stage_names(self):
        if self._a0 is not None and self._a0['mode'].startswith('deferred'):
            a1 = 'interface+sweep_axis1_corrected'
            if self._a0['fused']:
                return ['halo+explicit+sweep_axis0_zero_boundary', a1, 'sweep_axis2_contig']
            return ['halo+explicit', 'sweep_axis0_zero_boundary', a1, 'sweep_axis2_contig']
        if self._fused_now():
            return ['halo+explicit+sweep_axis0_distributed', 'sweep_axis1', 'sweep_axis2_contig']
        return ['halo+explicit', 'sweep_axis0_distributed', 'sweep_axis1', 'sweep_axis2_contig']

    @property
    def pass_a_form(self):
        """how pass A of the sharded-axis sweep runs under the current plan (bench.py reports it)"""
        p = self._a0 or {}
        if str(p.get('mode')).startswith('deferred'):
            return 'none (zero-boundary solve; planes 0 and n-1 of its result are the pass-A right-hand sides)'
        return 'dots_in_explicit' if p.get('dots') else ('fused' if p.get('fused') else 'separate')

    @property
    def stage_bytes_per_cell(self):
        """algorithmic HBM bytes per local cell of the stages (pass A re-reads the inputs of the rows it covers)"""
        bpc = self._bpc
        frac = 0.0
        if self._a0 is not None and self._a0['mode'].startswith('deferred'):
            # one pass per sweep; the two interface planes the axis-1 sweep re-reads are 2 * ny * nz values per slab
            return [bpc[0], bpc[1], bpc[2]] if self._a0['fused'] else [self._explicit_bpc, bpc[0], bpc[1], bpc[2]]
        if self.world > 1:
            frac = 1.0 if (self._a0 is None or self._a0['mode'] != 'window') else min(1.0, 2.0 * self._a0['K'] / self.nxl)
            if self._a0 is not None and self._a0.get('dots'):
                frac = 0.0                                  # pass A rides on the explicit stage: no second read
        if self._fused_now():
            if self._a0 is not None and self._a0.get('keep_r0'):
                return [2.0 * bpc[0], bpc[1], bpc[2]]      # pass A: state + flags in, R0 out; pass B: R0 + flags in, U out
            return [bpc[0] + frac * (bpc[0] - 8), bpc[1], bpc[2]]
        return [self._explicit_bpc, bpc[0] + frac * (bpc[0] - 8), bpc[1], bpc[2]]

    def _fused_supported(self, K):
        """explicit stage folded into pass B (whole slab) and, with neighbours, into pass A on K planes"""
        E, L = self.engine, self.Lint
        if not self._allow_fused or not hasattr(E, 'sweep0_fused'):
            return False
        ok = E.fused_supported(self.nxl, self.ny, self.nz, L.sx, False)
        if self.world > 1:
            ok = ok and E.fused_supported(K, self.ny, self.nz, L.sx, True)
        return bool(ok)

    def _fused_now(self):
        if self.world == 1:
            return self._fused_supported(self.nxl)
        return bool(self._a0['fused']) if self._a0 is not None else self._fused_supported(self.nxl)

    @staticmethod
    def local_numpy(T):
        """host copy of a local field in whatever form step() returned it"""
        if isinstance(T, np.ndarray):
            return T
        t = T if isinstance(T, torch.Tensor) else T.t
        return t.cpu().contiguous().numpy()

    # -- the step -----------------------------------------------------------------------------------
    def _load_state(self, T):
        """-> extended buffer whose interior holds T (no copy when T is the view step() returned)."""
        t = T if isinstance(T, (torch.Tensor, np.ndarray)) else getattr(T, 't', T)
        for i, buf in enumerate(self._ext_bufs):
            if isinstance(t, torch.Tensor) and t.device == buf.device and \
                    t.data_ptr() == _interior(buf).data_ptr() and tuple(t.stride()) == tuple(buf.stride()):
                self._cur = i
                return buf
        buf = self._ext_bufs[self._cur]
        self._halo_ready = None                    # a foreign field: whatever halo was prefetched is not its halo
        src = t if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(t), dtype=np.float64))
        self._logical(_interior(buf)).copy_(src)
        return buf

    # -- how the axis-0 interface system is solved ------------------------------------------------------
    DECAY_TOL = 1e-17      # |aL|, |cF| at or below this are dropped (they multiply values of the size of the data)
    SLAB_CHUNK_EDGES = None      # e.g. (0.125, 0.5): cumulative line fractions of the 'slab' pipeline's chunks (None: halves)
    DOTS_MAX_NONUNIFORM = 0.05   # share of axis-0 lines that are not uniform above which pass A leaves the dot-product form

    def _chunk_ranges(self, nch, fractions=None):
        """chunks of lines (ranges of j) of the axis-0 pipeline.  fractions: cumulative edges in (0, 1), e.g. a small
        first chunk whose exchange is the only exposed one, each later exchange hiding behind the previous chunk's pass B"""
        if fractions is not None and self.ny >= 64:
            edges = [0] + [int(round(f * self.ny / 2.0)) * 2 for f in fractions] + [self.ny]
        else:
            nch = nch if self.ny >= 8 * nch else 1
            edges = [round(i * self.ny / nch) for i in range(nch + 1)]
        edges = sorted(set(edges))
        return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1) if edges[i + 1] > edges[i]]

    def _streams(self):
        E = self.engine
        self._use_streams = (E.device.type == 'cuda') and isinstance(self.comm, TorchDistComm) \
            and not self._no_overlap
        if self._use_streams and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=E.device, priority=self._comm_priority)
        return self._use_streams

    def _window_guess(self, gam):
        """rows after which the coupling through a run of interior rows (-tg, 1+2tg, -tg) has decayed below
        DECAY_TOL: rho = 2tg / (1 + 2tg + sqrt(1 + 4tg)) per row; every other row is more dominant"""
        tg = self.params.theta * gam
        if tg <= 0.0:
            return 8
        rho = 2.0 * tg / (1.0 + 2.0 * tg + np.sqrt(1.0 + 4.0 * tg))
        need = int(np.ceil(np.log(self.DECAY_TOL) / np.log(rho))) + 2
        K = 8
        while K < need:            # powers of two: the in-register condensation kernels tile those
            K *= 2
        return K

    def _plan_axis0(self, Ti, gam):
        """-> plan dict for the current (dt, theta, mask): mode 'exact' | 'window' (first/last K planes) |
        'slab' (the whole slab is its own window).  Collective: every rank calls it at the same step."""
        prm = self.params
        key = (float(prm.dt), float(prm.theta), self._mask_version, self._force_exact, self._no_overlap,
               self._allow_fused, self._allow_window, self._keep_r0, self._allow_dots, self._allow_deferred,
               self._allow_deferred_exact, self._allow_deferred_lines, self._deferred_lines_cost_ratio, self._allow_quick_replan)
        if self._a0_key == key:
            self._plan_steps += 1
            return self._a0
        # a plan that lives for a few steps only (a layer-birth loop: every birth changes the mask) does not repay the two
        # extra sweeps and host synchronisations the per-line deferred form costs per plan; the same on every rank
        short_lived = self._plan_steps is not None and self._plan_steps < 16
        self._plan_steps = 0
        E, v = self.engine, self.variant
        self._streams()
        nl, fl, pk = self.nlines, self.flags_int, self.packs_int[0]
        first, last = self.rank == 0, self.rank == self.world - 1
        if short_lived and self._allow_quick_replan:
            plan = self._quick_plan(gam)
            if plan is not None:
                self._a0_key, self._a0 = key, plan
                self.axis0_mode = plan['mode']
                return plan
        # Every rank evaluates BOTH candidate forms on its own slab and the form is agreed collectively: the eligibility
        # of 'window' (4 K <= planes of this rank) and the decay tests depend on the rank's slab -- split_planes hands out
        # slabs that differ by two planes -- and ranks running different forms would exchange messages of different sizes
        # (a hang or corruption on RCCL).  window: all ranks pass it with the same K; else slab: all ranks pass it;
        # else exact.
        def decayed(k):
            Lw = E.layout(k, self.ny, self.nz, self.Lint.sx)
            worst = 0.0
            c = E.vec(6 * nl)
            if not first:      # first-rows window: (gF, aF) are used, cF must have decayed
                E.condense(0, v, Lw, Ti[:k], fl[:k], tuple(None if t is None else t[:k] for t in pk), prm.theta,
                           gam, prm.dt, self.Tinf, c)
                worst = max(worst, float(c.view(6, nl)[2].abs().max()))
            if not last:       # last-rows window: (gL, cL) are used, aL must have decayed
                o = self.nxl - k
                E.condense(0, v, Lw, Ti[o:], fl[o:], tuple(None if t is None else t[o:] for t in pk), prm.theta,
                           gam, prm.dt, self.Tinf, c)
                worst = max(worst, float(c.view(6, nl)[4].abs().max()))
            return Lw, worst, bool(worst <= self.DECAY_TOL)          # NaN compares false -> not decayed
        # Deferred form first (its eligibility costs one pass over the flags and a host-side solve): every sharded-axis
        # line of every slab uniform -- solid, no Dirichlet cell -- and the homogeneous solution decayed across every slab.
        # Where the weights have NOT decayed across a slab (thin slabs: strong scaling) the same algebra holds with the
        # interface system of all ranks (all-gather) and per-line end corrections on the first / last rank: 'deferred_exact'.
        dfr, uniform = None, False
        if self._allow_deferred and hasattr(E, 'deferred_setup') and self.nxl >= 2 and prm.theta * gam > 0.0:
            # uniform lines need every cell of the slab in the mask: where the engine keeps the collective all-solid hint
            # (set_mask / set_mask_device) a slab that is not all-solid is not classified at all -- one kernel and one host
            # synchronisation less in every plan of a layer-birth loop
            if self._padded:
                # padded planes: the lines of the padding are off-mask, so the device classification would say "not uniform";
                # but they are identity rows of a state that is zero there for good, their boundary planes are zero, their
                # interface values and with them the correction come out as exact zeros -- what decides is the caller's box
                uniform = bool(getattr(self, '_solid_everywhere', False)) and self._bc['dir_mask'] is None
            else:
                uniform = (getattr(E, 'box_hint', 2) == 2) and bool(E.lines_all_uniform(self.Lint, fl, pk[1]))
            if uniform:
                dfr = E.deferred_setup(self.nxl, prm.theta, gam, self.DECAY_TOL)
        dflag = E.vec(3)
        dflag[0] = 1.0 if dfr is not None else 0.0
        dflag[1] = 1.0 if (self._allow_fused and hasattr(E, 'sweep0_fused')
                           and E.fused_supported(self.nxl, self.ny, self.nz, self.Lint.sx, False)) else 0.0
        dflag[2] = 1.0 if (dfr is not None and dfr['reach'] < self.nxl) else 0.0
        alld = E.vec(3 * self.world)
        self.comm.all_gather(alld, dflag)
        alld = alld.view(self.world, 3)
        if float(alld[:, 0].min()) >= 1.0:
            nl_ = self.nlines
            all_decayed = float(alld[:, 2].min()) >= 1.0 and not self._force_exact
            plan = dict(mode='deferred' if all_decayed else 'deferred_exact', K=dfr['reach'], dfr=dfr,
                        fused=bool(float(alld[:, 1].min()) >= 1.0), dots=False, keep_r0=False, chunks=[],
                        ulo=E.vec(nl_), uhi=E.vec(nl_))
            if all_decayed:
                plan.update(prev_last=E.vec(nl_), next_first=E.vec(nl_))
            elif hasattr(E, 'deferred_exact_setup') and self._allow_deferred_exact:
                plan['dfr'] = E.deferred_setup(self.nxl, prm.theta, gam, 0.0)          # no weight is cut
                mat, scal = E.vec(4 * nl_), E.vec(2)
                plan.update(g=E.vec(2 * nl_), g_all=E.vec(2 * nl_ * self.world), mat_all=E.vec(4 * nl_ * self.world),
                            scal_all=E.vec(2 * self.world), xlo=E.vec(nl_), xhi=E.vec(nl_))
                plan['dx'] = E.deferred_exact_setup(self.Lint, fl, pk, prm.theta, gam, prm.dt, plan['dfr'], mat, scal)
                self.comm.all_gather(plan['mat_all'], mat)          # the matrix entries travel once per plan
                self.comm.all_gather(plan['scal_all'], scal)
            else:
                plan = None
            if plan is not None:
                self._a0_key, self._a0 = key, plan
                self.axis0_mode = plan['mode']
                return plan
        # Lines that are not uniform (curved solids, voids, Dirichlet cells): the same algebra with the two homogeneous solutions
        # of EVERY line, where they have decayed across every slab ('deferred_lines', include/adi_hip.h ABI v17).  Collective.
        if (self._allow_deferred and self._allow_deferred_lines and hasattr(E, 'homogeneous_solution') and not self._force_exact
                and not short_lived and prm.theta * gam > 0.0):
            # (every term of this gate is the same on every rank: what depends on the rank's own slab -- its thickness, the decay
            # of its lines -- is decided inside, behind the all-gather; a rank-local `nxl >= 2` here hung slabs of [8, 1] planes)
            plan = self._plan_deferred_lines(fl, pk, gam, bool(float(alld[:, 1].min()) >= 1.0))
            if plan is not None:
                self._a0_key, self._a0 = key, plan
                self.axis0_mode = plan['mode']
                return plan
        K = self._window_guess(gam)
        cand = {}
        if not self._force_exact:
            # windows pay off when they are a small part of the slab (pass A on 2K planes instead of all of them)
            if self._allow_window and 4 * K <= self.nxl:
                cand['window'] = (K,) + decayed(K)
            cand['slab'] = (self.nxl,) + decayed(self.nxl)
        prop = E.vec(3)
        prop[0] = 1.0 if ('window' in cand and cand['window'][3]) else 0.0
        prop[1] = float(K)
        prop[2] = 1.0 if ('slab' in cand and cand['slab'][3]) else 0.0
        allp = E.vec(3 * self.world)
        self.comm.all_gather(allp, prop)
        allp = allp.view(self.world, 3)
        plan = None
        if float(allp[:, 0].min()) >= 1.0 and float(allp[:, 1].min()) == float(allp[:, 1].max()):
            k, Lw, worst, _ = cand['window']
            plan = dict(mode='window', K=k, Lw=Lw, worst=worst)
        elif float(allp[:, 2].min()) >= 1.0:
            k, Lw, worst, _ = cand['slab']
            plan = dict(mode='slab', K=k, Lw=Lw, worst=worst)
        flag = E.vec(1)
        allf = E.vec(self.world)
        if plan is None:
            plan = dict(mode='exact', K=self.nxl)
        plan['fused'] = self._fused_supported(plan['K'])     # the same on every rank (it depends on sizes only...
        if self.world > 1:                                   # ...but slabs may differ by two planes: make it collective)
            flag.fill_(1.0 if plan['fused'] else 0.0)
            self.comm.all_gather(allf, flag)
            plan['fused'] = bool(float(allf.min()) >= 1.0)
        # whole-slab pass A ('slab', 'exact'): it stores R0 and pass B is the plain sweep; 'window' condenses only the
        # boundary planes, so its pass B evaluates the explicit stage itself
        plan['keep_r0'] = bool(plan['fused'] and plan['mode'] != 'window' and self._keep_r0)
        # whole-slab pass A without a second read of the slab: the marching explicit kernel accumulates the dot products
        # pass A needs while it writes R0 (uniform lines; the others are condensed from R0); pass B is the plain sweep
        dots = bool(self._allow_dots and plan['mode'] != 'window' and hasattr(E, 'dots_setup')
                    and E.dots_supported(self.nxl, self.ny, self.nz, self.Lint.sx))
        flag.fill_(1.0 if dots else 0.0)
        self.comm.all_gather(allf, flag)
        plan['dots'] = bool(float(allf.min()) >= 1.0)
        if plan['dots']:
            # the dot products serve lines that are uniform along the axis; the others are condensed one thread per
            # line from R0.  On a curved solid most lines are of the second kind: there the tiled pass-A kernels (which
            # take surface segments, section 3.2b of DESIGN.md) are the better pass A.  Collective decision.
            dd = E.dots_setup(self.Lint, self.flags_int, self.packs_int[0][1], prm.theta, gam)
            flag.fill_(float(E.dots_nonuniform_fraction(dd)) if hasattr(E, 'dots_nonuniform_fraction') else 0.0)
            self.comm.all_gather(allf, flag)
            if float(allf.max()) > self.DOTS_MAX_NONUNIFORM:
                plan['dots'] = False
            else:
                plan['fused'] = plan['keep_r0'] = False
                plan['dd'] = dd
        # chunks of lines of the axis-0 pipeline.  Tiled pass A: the exchange of a chunk hides behind pass A of the next
        # (2 chunks, 4 for the all-gather form).  Dot-product pass A has no pass-A kernel to hide behind and every extra
        # exchange costs its fixed latency: ONE chunk (RCCL self-loop rehearsal at 512^3: 2.06 ms with two halves, 1.97 with
        # one chunk; finer first chunks were slower still).  A window's exchange hides behind the middle planes' explicit stage.
        if plan['dots'] or plan['mode'] == 'window':
            ranges = self._chunk_ranges(1)
        elif plan['mode'] == 'exact':
            # (small slabs: the kernels of a quarter of the lines take microseconds and the step is bound by the host's
            # launch path -- 64 x 256 x 320 in the layer-birth loop: 0.94 -> 0.5 ms of host time per step)
            ranges = self._chunk_ranges(4 if self.nxl * self.ny * self.nz >= (1 << 25) else 1)
        else:
            ranges = self._chunk_ranges(2, self.SLAB_CHUNK_EDGES)
        bufs = []
        for j0, j1 in ranges:
            n = (j1 - j0) * self.nz
            Lc = E.layout(plan['K'], j1 - j0, self.nz, self.Lint.sx)
            Lb = E.layout(self.nxl, j1 - j0, self.nz, self.Lint.sx)
            b = dict(j0=j0, j1=j1, nl=n, Lc=Lc, Lb=Lb, xlo=E.vec(n), xhi=E.vec(n))
            if plan['mode'] == 'exact':
                b.update(cond=E.vec(6 * n), cond_all=E.vec(6 * n * self.world))
            else:
                b['cond_hi'] = E.vec(6 * n)
                b['cond_lo'] = E.vec(6 * n) if plan['mode'] == 'window' else b['cond_hi']
                b['prev_hi'] = E.vec(3 * n)        # (gL, aL, cL) of the slab below
                b['next_lo'] = E.vec(2 * n)        # (gF, aF) of the slab above
            bufs.append(b)
        plan['chunks'] = bufs
        if plan['mode'] != 'window':
            self._dots_sticky = bool(plan['dots'])          # (a re-plan of a short-lived plan keeps this decision, _quick_plan)
        self._a0_key, self._a0 = key, plan
        self.axis0_mode = plan['mode']
        return plan

    # ---- re-planning inside an event loop ---------------------------------------------------------------------------------
    # A layer-birth loop changes the mask and the time step at every birth, i.e. the plan lives for a step or two, and the
    # full planner above costs eight host synchronisations per plan (the measured decay of the condensed entries, the
    # collective flags, the share of non-uniform lines): 0.5 ms per birth, 44 % of the per-rank loop of BASELINE.json
    # configs[4] on 64-plane slabs (scripts/waam_slab_profile.py).  While plans are short-lived they are made WITHOUT reading
    # anything back:
    #   * the interface form from the rigorous bound instead of the measurement: the coupling through k rows is at most rho^k,
    #     rho the per-row factor of solid interior rows (-tg, 1+2tg, -tg) -- every other row is more dominant, a row that lacks a
    #     neighbour or is a Dirichlet cell cuts the coupling altogether -- so rho^K <= DECAY_TOL with 4 K <= the thinnest slab
    #     proves 'window', rho^n <= DECAY_TOL proves 'slab', and 'exact' is always valid.  (The measurement can accept 'slab' where
    #     the bound cannot -- thin-walled parts whose lines never run through a whole slab; the next long-lived plan measures again.)
    #   * whether the fused kernels take the slab: a function of the extents, agreed collectively once per K;
    #   * dot-product or tiled pass A: what the last full plan found (a performance choice, both are exact);
    #   * the interface buffers of the previous plan of the same form are reused.
    # Everything above is the same on every rank, so the ranks keep agreeing on the form without talking.
    def _slab_planes_of_all_ranks(self):
        if self._nx_all is None:
            E = self.engine
            mine, allv = E.vec(1), E.vec(self.world)
            mine.fill_(float(self.nxl))
            self.comm.all_gather(allv, mine)
            self._nx_all = [int(round(float(x))) for x in allv.tolist()]
        return self._nx_all

    def _fused_collective(self, K):
        if K not in self._fused_by_K:
            E = self.engine
            flag, allf = E.vec(1), E.vec(self.world)
            flag.fill_(1.0 if self._fused_supported(K) else 0.0)
            self.comm.all_gather(allf, flag)
            self._fused_by_K[K] = bool(float(allf.min()) >= 1.0)
        return self._fused_by_K[K]

    def _quick_plan(self, gam):
        """plan for the current (dt, theta, mask) without a host synchronisation, or None (no full plan has been made yet, or
        the slab is an all-solid box: the deferred forms need the full planner)"""
        E, prm = self.engine, self.params
        if self._dots_sticky is None or getattr(self, '_solid_everywhere', False) or getattr(E, 'box_hint', 0) == 2:
            return None
        nmin = min(self._slab_planes_of_all_ranks())
        Kg = self._window_guess(gam)
        if self._force_exact or prm.theta * gam <= 0.0:
            mode, K = 'exact', self.nxl
        elif self._allow_window and 4 * Kg <= nmin:
            mode, K = 'window', Kg
        elif Kg <= nmin:
            mode, K = 'slab', self.nxl
        else:
            mode, K = 'exact', self.nxl
        plan = dict(mode=mode, K=K, quick=True)
        if mode != 'exact':
            plan.update(Lw=E.layout(K, self.ny, self.nz, self.Lint.sx), worst=None)
        plan['fused'] = self._fused_collective(K)
        plan['keep_r0'] = bool(plan['fused'] and mode != 'window' and self._keep_r0)
        plan['dots'] = bool(self._dots_sticky and self._allow_dots and mode != 'window' and hasattr(E, 'dots_setup')
                            and E.dots_supported(self.nxl, self.ny, self.nz, self.Lint.sx))
        if plan['dots']:
            plan['dd'] = E.dots_setup(self.Lint, self.flags_int, self.packs_int[0][1], prm.theta, gam)
            plan['fused'] = plan['keep_r0'] = False
        if plan['dots'] or mode == 'window':
            ranges = self._chunk_ranges(1)
        elif mode == 'exact':
            ranges = self._chunk_ranges(4 if self.nxl * self.ny * self.nz >= (1 << 25) else 1)
        else:
            ranges = self._chunk_ranges(2, self.SLAB_CHUNK_EDGES)
        ck = (mode, K, tuple(ranges))
        bufs = self._chunk_cache.get(ck)
        if bufs is None:
            bufs = []
            for j0, j1 in ranges:
                n = (j1 - j0) * self.nz
                b = dict(j0=j0, j1=j1, nl=n, Lc=E.layout(K, j1 - j0, self.nz, self.Lint.sx),
                         Lb=E.layout(self.nxl, j1 - j0, self.nz, self.Lint.sx), xlo=E.vec(n), xhi=E.vec(n))
                if mode == 'exact':
                    b.update(cond=E.vec(6 * n), cond_all=E.vec(6 * n * self.world))
                else:
                    b['cond_hi'] = E.vec(6 * n)
                    b['cond_lo'] = E.vec(6 * n) if mode == 'window' else b['cond_hi']
                    b['prev_hi'] = E.vec(3 * n)
                    b['next_lo'] = E.vec(2 * n)
                bufs.append(b)
            self._chunk_cache[ck] = bufs
        for b in bufs:
            b.pop('matrix_sent', None)            # a new plan: the matrix parts of the interface travel again
        plan['chunks'] = bufs
        return plan

    def _plan_deferred_lines(self, fl, pk, gam, fused_ok):
        """plan of the deferred form with per-line homogeneous solutions, or None (not decayed on some rank).  Per rank and
        plan: two axis-0 sweeps of a zero field with unit boundary values -> w_lo, w_hi of every line; K planes of each are
        kept (beyond them every entry must be below DECAY_TOL, checked on the device), the planes next to the interfaces
        travel to the neighbours once."""
        E, prm, n, nl_ = self.engine, self.params, self.nxl, self.nlines
        first, last = self.rank == 0, self.rank == self.world - 1
        # where to look: the reach of the uniform row's solution plus a margin (a run that ends in a line start reflects);
        # what decides is the check below
        dfr = E.deferred_setup(n, prm.theta, gam, self.DECAY_TOL)
        K = min(n, int(dfr['reach']) + 8)
        ok = n >= 2 and K < n
        om, sides = {}, {}
        dm = pk[1]                                  # Dirichlet mask of the slab (None: no Dirichlet cells)

        def sort_lines(Wr, fr, dr):
            """(uniform marks [nlines] uint8, cells of the flagged lines int32, their weights [K][nflag]) of one side.  Wr, fr,
            dr: the K planes of the homogeneous solution, the flags and the Dirichlet mask within reach of the interface, counted
            from it.  A line is uniform when every one of them is a solid interior row (bit 0 and both sharded-axis neighbour
            bits of its flags byte) without a Dirichlet cell -- its homogeneous solution is then the scalar w[i] there, up to
            the decay tolerance --, off when none is in the mask (identity rows: weight 0), flagged otherwise."""
            uni = ((fr & 7) == 7).all(dim=0)
            if dr is not None:
                uni &= ~(dr != 0).any(dim=0)
            off = ((fr & 1) == 0).all(dim=0)
            cells = (~(uni | off)).reshape(-1).nonzero().reshape(-1)
            wc = Wr.reshape(Wr.shape[0], -1).index_select(1, cells).contiguous()
            # rows of each flagged line that can carry a non-zero weight: everything beyond its first row outside the mask is an
            # exact zero (identity rows cut the coupling) and need not even be read
            rows = torch.arange(1, wc.shape[0] + 1, device=wc.device, dtype=torch.int32).reshape(-1, 1)
            nrows = ((wc != 0).to(torch.int32) * rows).amax(dim=0).to(torch.int32).contiguous() if cells.numel() else cells.to(torch.int32)
            return uni.reshape(-1).to(torch.uint8).contiguous(), cells.to(torch.int32).contiguous(), wc, nrows
        nflag = 0
        if ok and not first:
            W = E.homogeneous_solution(self.variant, self.Lint, fl, pk, prm.theta, gam, prm.dt, True)
            ok = bool(float(W[K:].abs().max()) <= self.DECAY_TOL)             # NaN compares false
            om['lo_own'] = W[0].clone()
            if ok:
                sides['lo'] = sort_lines(W[:K], fl[:K], None if dm is None else dm[:K])
                nflag += int(sides['lo'][1].numel())
            del W
        if ok and not last:
            W = E.homogeneous_solution(self.variant, self.Lint, fl, pk, prm.theta, gam, prm.dt, False)
            ok = bool(float(W[:n - K].abs().max()) <= self.DECAY_TOL)
            om['hi_own'] = W[n - 1].clone()
            if ok:
                sides['hi'] = sort_lines(W[n - K:].flip(0), fl[n - K:], None if dm is None else dm[n - K:])   # row r: plane n-1-r
                nflag += int(sides['hi'][1].numel())
            del W
        # ... and where it pays: the sparse pass moves 24 B per flagged line and row within reach (its weight, the value read
        # and written); what it replaces is pass A of the two-pass forms, 17 B per cell of the 2 K planes of a window or of the
        # whole slab.  A solid riddled with voids (every line flagged) is left to those; a part with a few cavities is not.
        ok = ok and 24.0 * nflag * K <= self._deferred_lines_cost_ratio * 17.0 * nl_ * min(2 * K, n)
        flag, allf = E.vec(1), E.vec(self.world)
        flag.fill_(1.0 if ok else 0.0)
        self.comm.all_gather(allf, flag)
        if float(allf.min()) < 1.0:
            return None
        # the planes next to the interfaces, once per plan: mine down / up, the neighbours' back
        dummy = E.vec(nl_).view(self.ny, self.nz)
        om['hi_prev'] = None if first else E.vec(nl_).view(self.ny, self.nz)
        om['lo_next'] = None if last else E.vec(nl_).view(self.ny, self.nz)
        self.comm.exchange_planes(om.get('lo_own', dummy), om.get('hi_own', dummy),
                                  dummy if first else om['hi_prev'], dummy if last else om['lo_next'])
        return dict(mode='deferred_lines', K=K, fused=fused_ok, dots=False, keep_r0=False, chunks=[], om=om, dfr=dfr, sides=sides,
                    nflag=nflag, ulo=E.vec(nl_), uhi=E.vec(nl_), ulo_uni=E.vec(nl_), uhi_uni=E.vec(nl_),
                    prev_last=E.vec(nl_), next_first=E.vec(nl_))

    def _condense_box(self, plan, L, src, p0, p1, j0, j1, cond):
        """pass A on planes [p0, p1), rows [j0, j1) of the slab.  src: the explicit stage's output (interior view) or,
        when the explicit stage is folded into the pass, the extended state itself."""
        E, prm, v = self.engine, self.params, self.variant
        cut = lambda t: None if t is None else t[p0:p1, j0:j1, :]
        fl, pk = cut(self.flags_int), tuple(cut(t) for t in self.packs_int[0])
        if plan['fused']:
            # a pass A over the whole slab leaves R0 in the scratch field for pass B (no second explicit evaluation)
            r0 = _interior(self._tmp[0])[p0:p1, j0:j1, :] if plan['keep_r0'] else None
            E.condense0_fused(v, L, src, 1 + p0, j0, fl, pk, self.dx, prm.dt, self._kappa, prm.theta, self.Tinf, cond, r0)
        else:
            E.condense(0, v, L, cut(src), fl, pk, prm.theta, self._gam, prm.dt, self.Tinf, cond)

    def _condense_windows(self, plan, Ai, b):
        """pass A on one chunk of lines: the condensations this rank's neighbours need"""
        K, n = plan['K'], self.nxl
        j0, j1 = b['j0'], b['j1']
        if plan['dots']:
            E, prm = self.engine, self.params
            E.dots_finish(self.variant, self.Lint, plan['dd'], Ai, self.flags_int, self.packs_int[0], prm.theta, self._gam,
                          prm.dt, self.Tinf, j0 * self.nz, j1 * self.nz, b['cond'] if plan['mode'] == 'exact' else b['cond_hi'])
            return
        if plan['mode'] == 'exact':
            self._condense_box(plan, b['Lb'], Ai, 0, n, j0, j1, b['cond'])
        elif plan['mode'] == 'slab':
            if self.world > 1:
                self._condense_box(plan, b['Lb'], Ai, 0, n, j0, j1, b['cond_hi'])
        else:
            if self.rank > 0:
                self._condense_box(plan, b['Lc'], Ai, 0, K, j0, j1, b['cond_lo'])
            if self.rank < self.world - 1:
                self._condense_box(plan, b['Lc'], Ai, n - K, n, j0, j1, b['cond_hi'])

    def _exchange_interface(self, plan, b):
        if plan['mode'] == 'exact':
            self.comm.all_gather(b['cond_all'], b['cond'])
        else:
            n = b['nl']
            if b.get('matrix_sent') and self._send_g_only:
                # aF and (aL, cL) are entries of the condensed MATRIX: they depend on dt, theta and the mask only (this
                # plan), so after the first exchange only the right-hand-side parts gF / gL travel (2 of 5 arrays)
                self.comm.exchange_planes(b['cond_lo'].view(6, n)[0:1], b['cond_hi'].view(6, n)[3:4],
                                          b['prev_hi'].view(3, n)[0:1], b['next_lo'].view(2, n)[0:1])
            else:
                self.comm.exchange_planes(b['cond_lo'].view(6, n)[0:2], b['cond_hi'].view(6, n)[3:6],
                                          b['prev_hi'].view(3, n), b['next_lo'].view(2, n))
                b['matrix_sent'] = True

    def _solve_and_sweep(self, plan, Ai, Bi, b):
        """interface values of one chunk of lines, then pass B: the local sweep with them injected"""
        E, prm, v = self.engine, self.params, self.variant
        j0, j1, n = b['j0'], b['j1'], self.nxl
        fl, pk = self.flags_int, self.packs_int[0]
        if plan['mode'] == 'exact':
            E.interface(b['cond_all'], self.world, self.rank, b['nl'], b['xlo'], b['xhi'])
        else:
            E.interface_pair(b['cond_lo'], b['cond_hi'], b['prev_hi'] if self.rank > 0 else None,
                             b['next_lo'] if self.rank < self.world - 1 else None, b['nl'], b['xlo'], b['xhi'])
        cut = lambda t: None if t is None else t[:, j0:j1, :]
        if plan['fused'] and plan['keep_r0']:
            E.sweep(0, v, b['Lb'], cut(_interior(self._tmp[0])), cut(fl), tuple(cut(t) for t in pk), prm.theta, self._gam,
                    prm.dt, self.Tinf, cut(Bi), b['xlo'], b['xhi'])
        elif plan['fused']:
            E.sweep0_fused(v, b['Lb'], Ai, 1, j0, cut(fl), tuple(cut(t) for t in pk), self.dx, prm.dt, self._kappa,
                           prm.theta, self.Tinf, cut(Bi), b['xlo'], b['xhi'])
        else:
            E.sweep(0, v, b['Lb'], cut(Ai), cut(fl), tuple(cut(t) for t in pk), prm.theta, self._gam, prm.dt, self.Tinf,
                    cut(Bi), b['xlo'], b['xhi'])

    def _axis0_pipeline(self, plan, Ai, Bi, condensed=False, before_chunk=None):
        """pass A, exchange, interface solve and pass B over the chunks of lines; with RCCL the exchange of chunk c
        runs on a second stream while chunk c+1 is condensed and chunk c-1 is solved.  before_chunk(b): work that has to
        precede pass A of chunk b on the main stream (the explicit stage of its rows in the dot-product form)."""
        bufs = plan['chunks']
        use_streams = self._use_streams
        main = torch.cuda.current_stream() if use_streams else None
        ev_x = []
        for b in bufs:
            if before_chunk is not None:
                before_chunk(b)
            if not condensed:
                self._condense_windows(plan, Ai, b)
            if use_streams:
                e = torch.cuda.Event(); e.record(main)
                with torch.cuda.stream(self._comm_stream):     # issued now: it overlaps the next chunk's main-stream work
                    self._comm_stream.wait_event(e)
                    self._exchange_interface(plan, b)
                    e2 = torch.cuda.Event(); e2.record(self._comm_stream); ev_x.append(e2)
        return ev_x

    def _axis0_finish(self, plan, Ai, Bi, ev_x):
        use_streams = self._use_streams
        main = torch.cuda.current_stream() if use_streams else None
        for i, b in enumerate(plan['chunks']):
            if use_streams:
                main.wait_event(ev_x[i])
            else:
                self._exchange_interface(plan, b)
            self._solve_and_sweep(plan, Ai, Bi, b)

    def self_check(self, T):
        """One step as configured (neighbour-only interface solve where the decay allows it, exchanges on the
        second stream) and one with the exact all-gather solve and every exchange on the main stream, from the same
        input; they must agree to rounding.  If they do not (a stream-ordering problem on this software stack),
        the conservative configuration is kept for the rest of the run.
        Returns (max relative difference, fast configuration enabled)."""
        t = T.t if hasattr(T, 't') and not isinstance(T, torch.Tensor) else T
        src = t.clone()
        a = self.step(src)
        a = (a.t if hasattr(a, 't') and not isinstance(a, torch.Tensor) else a).clone()
        keep = (self._no_overlap, self._force_exact, self._allow_deferred)
        self._no_overlap, self._force_exact, self._allow_deferred = True, True, False     # the two-pass all-gather form
        b = self.step(src)
        b = b.t if hasattr(b, 't') and not isinstance(b, torch.Tensor) else b
        den = float(b.abs().max())
        err = float((a - b).abs().max()) / (den if den > 0 else 1.0)
        ok = torch.tensor([1.0 if err <= 1e-12 else 0.0], dtype=torch.float64, device=self.engine.device)
        allok = self.engine.vec(self.world)
        self.comm.all_gather(allok, ok)                 # the ranks must not end up in different configurations
        self._allow_deferred = keep[2]
        self._plan_steps = None                   # the plans of this check do not count as short-lived ones
        if float(allok.min()) >= 1.0:
            self._no_overlap, self._force_exact = keep[:2]
        return err, not self._no_overlap

    def step(self, T, events=None, prefetch_halo=False):
        """One ADI step of the local slab.  prefetch_halo=True promises that the returned field is passed to the
        next step() unmodified (an nsub loop).  Used by the 'window' form, whose step starts with the planes next
        to the halos: the boundary planes of the result are then computed first and sent to the neighbours while
        the rest of the last sweep runs.  (The other forms start with the planes that need no halo, which hides
        the exchange just as well.)"""
        E = self.engine
        if hasattr(E, '_stream_ptr'):
            E._stream_ptr = E.hip._stream()        # every kernel of this step goes to the stream that is current now
        try:
            return self._step(T, events, prefetch_halo)
        finally:
            if hasattr(E, '_stream_ptr'):
                E._stream_ptr = None

    def _step(self, T, events, prefetch_halo):
        E, prm, mat = self.engine, self.params, self.mat
        kappa = self._kappa = mat.k / (mat.rho * mat.cp)         # adi3d_numba_coeff.py:292
        gam = self._gam = kappa * prm.dt / (self.dx * self.dx)
        kind = 'torch' if isinstance(T, torch.Tensor) else ('numpy' if isinstance(T, np.ndarray) else 'field')
        Text = self._load_state(T)
        nxt = self._ext_bufs[self._cur ^ 1]
        A, B = self._tmp
        Ai, Bi, Oi = _interior(A), _interior(B), _interior(nxt)
        v, Li, fl = self.variant, self.Lint, self.flags_int
        nl = self.nxl
        ne = 0

        def mark():
            nonlocal ne
            if events is not None:
                events[ne].record()
            ne += 1
        mark()
        plan = self._plan_axis0(_interior(Text), gam) if self.world > 1 else None
        fused = plan['fused'] if plan is not None else self._fused_supported(nl)
        streams = self._streams() and self.world > 1
        main = torch.cuda.current_stream() if streams else None
        ex = lambda b, e: E.explicit(self.Lext, Text, self.flags_ext, self.dx, prm.dt, kappa, prm.theta, A, b, e)

        # 1. state halos (zeros outside the global grid are never read: the flags carry no coupling there)
        halo_ev = None
        if self._halo_ready == self._cur and self._halo_ready is not None:
            halo_ev = self._halo_event                       # sent at the end of the previous step
        elif self.world > 1:
            lo, hi = Text[0], Text[-1]
            if streams:
                ev0 = torch.cuda.Event(); ev0.record(main)
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev0)                   # Text is complete
                    self.comm.exchange_planes(Text[1], Text[-2], lo, hi)
                    halo_ev = torch.cuda.Event(); halo_ev.record(self._comm_stream)
            else:
                self.comm.exchange_planes(Text[1], Text[-2], lo, hi)
        self._halo_ready = None

        # 2. explicit stage, 3. axis-0 sweep
        if plan is not None and plan['mode'].startswith('deferred'):
            # every line of the slab solved with ZERO boundary values by the single-domain kernels; the first and last plane
            # of that result go to the neighbours, the 2 x 2 interface systems give the two boundary values of every line, and
            # the axis-1 sweep adds  ulo * w[i] + uhi * w[n-1-i]  to what it loads (step 4)
            if fused:
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                E.sweep0_fused(v, Li, Text, 1, 0, fl, self.packs_int[0], self.dx, prm.dt, kappa, prm.theta, self.Tinf, Bi)
            else:
                if nl >= 4:                                   # the planes that touch no halo run while the halos travel
                    ex(2, nl)
                    if halo_ev is not None and streams:
                        main.wait_event(halo_ev)
                    ex(1, 2); ex(nl, nl + 1)
                else:
                    if halo_ev is not None and streams:
                        main.wait_event(halo_ev)
                    ex(1, nl + 1)
                mark()
                E.sweep(0, v, Li, Ai, fl, self.packs_int[0], prm.theta, gam, prm.dt, self.Tinf, Bi)
            mark()
            if plan['mode'] == 'deferred_exact':
                # no decay: all-gather of (plane 0, plane n-1) of x0 -- the right-hand sides of the interface system, whose matrix
                # is the plan's -- the interface solve over all ranks, then the per-line coefficients of w[i] and w[n-1-i]
                g2 = plan['g'].view(2, self.ny, self.nz)
                g2[0].copy_(Bi[0]); g2[1].copy_(Bi[nl - 1])
                if streams:
                    ev0 = torch.cuda.Event(); ev0.record(main)
                    with torch.cuda.stream(self._comm_stream):
                        self._comm_stream.wait_event(ev0)
                        self.comm.all_gather(plan['g_all'], plan['g'])
                        ev1 = torch.cuda.Event(); ev1.record(self._comm_stream)
                    main.wait_event(ev1)
                else:
                    self.comm.all_gather(plan['g_all'], plan['g'])
                E.interface_uniform(plan['g_all'], plan['mat_all'], plan['scal_all'], self.world, self.rank, self.nlines,
                                    plan['xlo'], plan['xhi'])
                E.deferred_exact_coef(plan['dx'], plan['xlo'], plan['xhi'], self.nlines, plan['ulo'], plan['uhi'])
            else:
                # issued on the main stream: nothing can run beside this exchange (the next kernel needs the planes), and
                # every hop through another stream is a cross-queue dependency of 10 - 20 us on the critical path
                self.comm.exchange_planes(Bi[0], Bi[nl - 1], plan['prev_last'].view(self.ny, self.nz),
                                          plan['next_first'].view(self.ny, self.nz))
                first, last = self.rank == 0, self.rank == self.world - 1
                if plan['mode'] == 'deferred_lines':
                    # the interface values of every line, and the same masked to the lines that are uniform within reach (the
                    # scalar weights of the axis-1 sweep multiply those); the flagged lines get their own weights here, in memory
                    sd = plan['sides']
                    E.interface_deferred_lines(Bi[0], Bi[nl - 1], None if first else plan['prev_last'],
                                               None if last else plan['next_first'], plan['om'], self.nlines, plan['ulo'],
                                               plan['uhi'], sd['lo'][0] if 'lo' in sd else None, sd['hi'][0] if 'hi' in sd else None,
                                               plan['ulo_uni'], plan['uhi_uni'])
                    if 'lo' in sd:
                        E.deferred_lines_apply(Li, Bi, sd['lo'][1], sd['lo'][2], plan['ulo'], False, sd['lo'][3])
                    if 'hi' in sd:
                        E.deferred_lines_apply(Li, Bi, sd['hi'][1], sd['hi'][2], plan['uhi'], True, sd['hi'][3])
                else:
                    E.interface_deferred(Bi[0], Bi[nl - 1], None if first else plan['prev_last'],
                                         None if last else plan['next_first'], plan['dfr']['omega'], self.nlines, plan['ulo'],
                                         plan['uhi'])
        elif fused:
            # R0 never reaches HBM: both passes of the axis-0 sweep evaluate it from the state (halo planes included,
            # so they must have landed; in an nsub loop they were sent while the previous step's last sweep ran)
            if self.world == 1:
                E.sweep0_fused(v, Li, Text, 1, 0, fl, self.packs_int[0], self.dx, prm.dt, kappa, prm.theta, self.Tinf, Bi)
            else:
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                ev_x = self._axis0_pipeline(plan, Text, Bi)
                self._axis0_finish(plan, Text, Bi, ev_x)
        elif self.world == 1:
            ex(1, nl + 1)
            mark()
            E.sweep(0, v, Li, Ai, fl, self.packs_int[0], prm.theta, gam, prm.dt, self.Tinf, Bi)
        elif plan['dots']:
            # one pass over the slab: R0 and, per line, the two dot products of pass A (halos must have landed)
            # (running the explicit stage chunk by chunk of LINES, so that a chunk's interface exchange travels behind
            # the next chunk's explicit stage, was measured over the RCCL self-loop: no gain, 1.99 -> 2.02 ms)
            exd = lambda b, e: E.explicit_dots(self.Lext, Text, self.flags_ext, self.dx, prm.dt, kappa, prm.theta, A, b, e,
                                               plan['dd'], 1, nl)
            ich = E.dots_ichunk(nl) if hasattr(E, 'dots_ichunk') else nl
            if nl >= 3 * ich and nl % ich == 0:
                # the chunks of planes that touch no halo run while the halo planes are in flight
                exd(1 + ich, nl + 1 - ich)
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                exd(1, 1 + ich); exd(nl + 1 - ich, nl + 1)
            else:
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                exd(1, nl + 1)
            mark()
            ev_x = self._axis0_pipeline(plan, Ai, Bi)
            self._axis0_finish(plan, Ai, Bi, ev_x)
        elif plan['mode'] == 'window':
            # the boundary windows first: their condensation travels while the middle planes are computed
            K = plan['K']
            if halo_ev is not None and streams:
                main.wait_event(halo_ev)
            if 2 * K < nl:
                ex(1, K + 1); ex(nl - K + 1, nl + 1)
            else:
                ex(1, nl + 1)
            ev_x = self._axis0_pipeline(plan, Ai, Bi)
            if 2 * K < nl:
                ex(K + 1, nl - K + 1)
            mark()
            self._axis0_finish(plan, Ai, Bi, ev_x)
        else:
            # the planes that touch no halo run while the halo planes are in flight
            if nl >= 4:
                ex(2, nl)
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                ex(1, 2); ex(nl, nl + 1)
            else:
                if halo_ev is not None and streams:
                    main.wait_event(halo_ev)
                ex(1, nl + 1)
            mark()
            ev_x = self._axis0_pipeline(plan, Ai, Bi)
            self._axis0_finish(plan, Ai, Bi, ev_x)
        # 4. local sweeps
        if plan is not None and plan['mode'] == 'deferred_lines':
            E.sweep_corrected(v, Li, Bi, fl, self.packs_int[1], prm.theta, gam, prm.dt, self.Tinf, Ai,
                              None if self.rank == 0 else plan['ulo_uni'], None if self.rank == self.world - 1 else plan['uhi_uni'],
                              plan['dfr']['w'])
        elif plan is not None and plan['mode'].startswith('deferred'):
            ex_ = plan['mode'] == 'deferred_exact'        # (there the end ranks carry a Sherman-Morrison term on both vectors)
            E.sweep_corrected(v, Li, Bi, fl, self.packs_int[1], prm.theta, gam, prm.dt, self.Tinf, Ai,
                              None if (self.rank == 0 and not ex_) else plan['ulo'],
                              None if (self.rank == self.world - 1 and not ex_) else plan['uhi'], plan['dfr']['w'])
        else:
            mark()
            E.sweep(1, v, Li, Bi, fl, self.packs_int[1], prm.theta, gam, prm.dt, self.Tinf, Ai)
        mark()
        pk2 = self.packs_int[2]
        sw2 = lambda p0, p1: E.sweep(2, v, E.layout(p1 - p0, self.ny, self.nz, Li.sx), Ai[p0:p1], fl[p0:p1],
                                     tuple(None if t is None else t[p0:p1] for t in pk2), prm.theta, gam, prm.dt,
                                     self.Tinf, Oi[p0:p1])
        # the fused passes and the 'window' form start with planes that need the halos: send them early
        if prefetch_halo and self.world > 1 and nl >= 4 and (plan['mode'] == 'window' or fused):
            sw2(0, 1); sw2(nl - 1, nl)                        # the two planes the neighbours need
            if streams:
                ev0 = torch.cuda.Event(); ev0.record(main)
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev0)
                    self.comm.exchange_planes(nxt[1], nxt[-2], nxt[0], nxt[-1])
                    self._halo_event = torch.cuda.Event(); self._halo_event.record(self._comm_stream)
            else:
                self.comm.exchange_planes(nxt[1], nxt[-2], nxt[0], nxt[-1])
                self._halo_event = None
            self._halo_ready = self._cur ^ 1
            sw2(1, nl - 1)
        else:
            sw2(0, nl)
        mark()
        self._cur ^= 1
        Oi = self._logical(Oi)
        if kind == 'numpy':
            return Oi.cpu().contiguous().numpy()
        if kind == 'torch':
            return Oi
        return type(T)(Oi)
