"""dist_slab -- the Cartesian ADI step on a grid cut into slabs along memory axis 0, one slab per GPU.

The reference is single-process (no NCCL/MPI anywhere); this module is the multi-GPU
part of the north star: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Per step and rank (nxl local planes, halo planes at both ends of every extended array):
  1. halo exchange of the state: one (ny, nz) plane to each neighbour (send/recv, 2 MiB at 512^2);
     the mask halo travels only when the mask changes (pack rebuild).
  2. explicit stage on the extended slab (interior result used).
  3. axis-0 sweep, whose lines span all ranks, as a reduced-interface solve:
       pass A  adi_sweep_condense: every local line -> 6 numbers (first/last unknown as affine functions of
               the neighbours' adjacent unknowns),
       all_gather of the (6, ny*nz) block (12.6 MB per rank at 512^2),
       adi_interface_solve: each rank merges the slabs below/above it and solves a 2x2 system per line,
       pass B  adi_sweep with the boundary values injected: the ordinary local sweep.
     Exact (not iterative): the result equals the single-domain sweep to rounding.
  4. axis-1 and axis-2 sweeps: lines are local, no communication.

The numerical work is behind an `engine` (HIP kernels through the C ABI in production) and the exchange
behind a `comm`, so the algebra of the decomposition can be tested on CPU ranks over gloo with a
reference engine (tests/) and, on one GPU, with several in-process ranks.
"""
import ctypes
import threading

import numpy as np
import torch

__all__ = ['SlabStepper', 'TorchDistComm', 'LocalComm', 'HipEngine', 'split_planes']


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
    """torch.distributed (nccl = RCCL on ROCm, or gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def exchange_planes(self, send_lo, send_hi, recv_lo, recv_hi):
        """send_lo -> rank-1 (received there as recv_hi), send_hi -> rank+1 (received there as recv_lo)."""
        dist = self.dist
        ops = []
        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, send_lo, self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.rank - 1, self.group))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, send_hi, self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.rank + 1, self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def all_gather(self, out, inp):
        self.dist.all_gather_into_tensor(out, inp, group=self.group)


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


# ------------------------------------------------------------------------------------------ engine
class HipEngine:
    """The product engine: hand-written HIP kernels through the C ABI (include/adi_hip.h)."""

    def __init__(self):
        from . import adi3d_hip_coeff as hip
        from . import _lib
        self.hip, self._lib, self.lib, self.check = hip, _lib, _lib.lib, _lib.check
        self.device = hip._device()

    def layout(self, nx, ny, nz, sx=None):
        return self.hip.Layout(nx, ny, nz, sx)

    def vec(self, n):
        return torch.empty(n, dtype=torch.float64, device=self.device)

    def build_flags(self, L, mask_ext):
        flags = L.empty(torch.uint8, zero=True)
        self.check(self.lib.adi_build_nbr_flags(self.hip._p(mask_ext), L.nx, L.ny, L.nz, L.sx, self.hip._p(flags),
                                                self.hip._stream()))
        return flags

    def build_packs(self, L, mask_ext, flags_ext, dx, mat, dir_mask, dir_value, neumann, robin_h):
        """precompute_coeff_packs_unified on the extended slab; returns packs whose arrays are extended too."""
        g = self.hip.Grid3D.__new__(self.hip.Grid3D)
        g.nx, g.ny, g.nz, g.dx, g.layout = L.nx, L.ny, L.nz, float(dx), L
        g._mask, g._d_mask, g._d_flags, g._scratch, g.mask_version = None, mask_ext, flags_ext, None, 0
        g.sync_mask = lambda: mask_ext            # the device mask (with halos) is authoritative here
        return self.hip.precompute_coeff_packs_unified(g, mat, dir_mask=dir_mask, dir_value=dir_value,
                                                       neumann=neumann, robin_h=robin_h)

    def explicit(self, L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin=0, i_end=None):
        h = self.hip
        i_end = L.nx if i_end is None else i_end
        self.check(self.lib.adi_explicit_rhs_planes(h._p(T_ext), h._p(flags_ext), L.nx, L.ny, L.nz, L.sx, dx, dt, kappa,
                                                    theta, h._p(out_ext), i_begin, i_end, h._stream()))

    def _args(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf):
        h = self.hip
        return (axis, variant, h._p(t_in), h._p(flags), h._p(pack[0]), h._p(pack[1]), h._p(pack[2]), h._p(pack[3]),
                Li.nx, Li.ny, Li.nz, Li.sx, 1, theta, gam, dt, float(Tinf))   # packs come from adi_build_coeffs: sparse ok

    def _workspace(self, Li):
        """unit queue of the FAST/GENERAL kernel pair (sized for the largest box seen)"""
        need = 0
        for ax in range(3):
            b = ctypes.c_size_t(0)
            self.check(self.lib.adi_sweep_workspace_bytes(ax, Li.nx, Li.ny, Li.nz, Li.sx, ctypes.byref(b)))
            need = max(need, b.value)
        if getattr(self, '_work', None) is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._work

    def sweep(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, t_out, xlo=None, xhi=None):
        h = self.hip
        w = self._workspace(Li)
        self.check(self.lib.adi_sweep(*self._args(axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf),
                                      h._p(t_out), h._p(xlo), h._p(xhi), h._p(w), w.numel(), h._stream()))

    def condense(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, cond):
        h = self.hip
        w = self._workspace(Li)
        self.check(self.lib.adi_sweep_condense(*self._args(axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf),
                                               h._p(cond), h._p(w), w.numel(), h._stream()))

    def interface(self, cond_all, world, rank, nlines, xlo, xhi):
        h = self.hip
        self.check(self.lib.adi_interface_solve(h._p(cond_all), world, rank, nlines, h._p(xlo), h._p(xhi),
                                                h._stream()))


def _interior(t_ext):
    """planes 1..n-2 of an extended array: same strides, pointer advanced by one plane"""
    return t_ext[1:-1]


class SlabStepper:
    """One rank's slab.  step(T) takes / returns the LOCAL (nxl, ny, nz) field (DeviceField, torch tensor or
    NumPy array); the state stays in HBM when fed back what step() returned."""

    stage_names = ['halo+explicit', 'sweep_axis0_distributed', 'sweep_axis1', 'sweep_axis2_contig']

    def __init__(self, mask_local, dx, mat, params, Tinf=0.0, dir_mask=None, dir_value=None, neumann=None,
                 robin_h=None, comm=None, engine=None):
        self.engine = engine or HipEngine()
        self.comm = comm or TorchDistComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.mat, self.params, self.Tinf, self.dx = mat, params, float(Tinf), float(dx)
        self._bc = dict(dir_mask=dir_mask, dir_value=dir_value, neumann=neumann, robin_h=robin_h)
        mask_local = np.asarray(mask_local).astype(np.bool_)
        self.nxl, self.ny, self.nz = mask_local.shape
        E = self.engine
        self.Lext = E.layout(self.nxl + 2, self.ny, self.nz)
        self.Lint = E.layout(self.nxl, self.ny, self.nz)
        assert self.Lint.sx == self.Lext.sx
        self.nlines = self.ny * self.nz
        self._ext_bufs = [self.Lext.empty(), self.Lext.empty()]
        self._cur = 0
        self._tmp = [self.Lext.empty(), self.Lext.empty()]
        self._chunk_list = None
        self.set_mask(mask_local)

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
        self.flags_ext = E.build_flags(L, d_mask)

        def ext(a, fill):
            if a is None or np.isscalar(a):
                return a
            e = np.full((self.nxl + 2, self.ny, self.nz), fill, dtype=np.asarray(a).dtype)
            e[1:-1] = a
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
        bpc = [getattr(p, 'bytes_per_cell', float(_lib.SWEEP_BYTES_PER_CELL[self.variant])) for p in self.packs_ext]
        self.stage_bytes_per_cell = [float(_lib.EXPLICIT_BYTES_PER_CELL),
                                     (2 * (bpc[0] - 8) + 8) if self.world > 1 else bpc[0],   # pass A re-reads the inputs
                                     bpc[1], bpc[2]]

        def interior_pack(p):
            return tuple(None if t is None else _interior(t) for t in (p.d_coeff, p.d_dir_mask, p.d_dir_val, p.d_qflux))
        self.packs_int = [interior_pack(p) for p in self.packs_ext]
        self.flags_int = _interior(self.flags_ext)

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
        src = t if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(t), dtype=np.float64))
        _interior(buf).copy_(src)
        return buf

    def _chunks(self):
        """(j0, j1) ranges: the lines of the sharded sweep are cut along j so that each chunk is a sub-box"""
        if getattr(self, '_chunk_list', None) is None:
            nch = 4 if (self.ny >= 32) else 1
            edges = [round(i * self.ny / nch) for i in range(nch + 1)]
            self._chunk_list = [(edges[i], edges[i + 1]) for i in range(nch) if edges[i + 1] > edges[i]]
            E = self.engine
            self._chunk_bufs = []
            for j0, j1 in self._chunk_list:
                nl = (j1 - j0) * self.nz
                self._chunk_bufs.append(dict(L=E.layout(self.nxl, j1 - j0, self.nz, self.Lint.sx), nl=nl, cond=E.vec(6 * nl),
                                             cond_all=E.vec(6 * nl * self.world), xlo=E.vec(nl), xhi=E.vec(nl)))
            self._use_streams = (E.device.type == 'cuda') and isinstance(self.comm, TorchDistComm) \
                and not getattr(self, '_no_overlap', False)
            if self._use_streams:
                self._comm_stream = torch.cuda.Stream(device=E.device)
        return self._chunk_list

    def _axis0_distributed(self, Ai, Bi, gam):
        E, prm, v = self.engine, self.params, self.variant
        chunks = self._chunks()
        fl = self.flags_int
        pk = self.packs_int[0]

        def sub(t, j0, j1):
            return None if t is None else t[:, j0:j1, :]
        use_streams = self._use_streams
        main = torch.cuda.current_stream() if use_streams else None
        ev_cond, ev_ag = [], []
        for (j0, j1), cb in zip(chunks, self._chunk_bufs):           # pass A for every chunk
            E.condense(0, v, cb['L'], sub(Ai, j0, j1), sub(fl, j0, j1), tuple(sub(t, j0, j1) for t in pk),
                       prm.theta, gam, prm.dt, self.Tinf, cb['cond'])
            if use_streams:
                e = torch.cuda.Event(); e.record(main); ev_cond.append(e)
        if use_streams:
            with torch.cuda.stream(self._comm_stream):                 # all-gathers on the second stream
                for cb, e in zip(self._chunk_bufs, ev_cond):
                    self._comm_stream.wait_event(e)
                    self.comm.all_gather(cb['cond_all'], cb['cond'])
                    e2 = torch.cuda.Event(); e2.record(self._comm_stream); ev_ag.append(e2)
        for i, ((j0, j1), cb) in enumerate(zip(chunks, self._chunk_bufs)):   # interface + pass B per chunk
            if use_streams:
                main.wait_event(ev_ag[i])
            else:
                self.comm.all_gather(cb['cond_all'], cb['cond'])
            E.interface(cb['cond_all'], self.world, self.rank, cb['nl'], cb['xlo'], cb['xhi'])
            E.sweep(0, v, cb['L'], sub(Ai, j0, j1), sub(fl, j0, j1), tuple(sub(t, j0, j1) for t in pk), prm.theta, gam,
                    prm.dt, self.Tinf, sub(Bi, j0, j1), cb['xlo'], cb['xhi'])

    def self_check(self, T):
        """One step with the second-stream pipeline and one with every exchange on the main stream, from the same
        input; they must agree to rounding.  If they do not (a stream-ordering problem on this software stack), the
        pipeline is switched off for the rest of the run.  Returns (max relative difference, overlap enabled)."""
        t = T.t if hasattr(T, 't') and not isinstance(T, torch.Tensor) else T
        src = t.clone()
        a = self.step(src)
        a = (a.t if hasattr(a, 't') and not isinstance(a, torch.Tensor) else a).clone()
        self._no_overlap = True
        self._chunk_list = None
        b = self.step(src)
        b = b.t if hasattr(b, 't') and not isinstance(b, torch.Tensor) else b
        den = float(b.abs().max())
        err = float((a - b).abs().max()) / (den if den > 0 else 1.0)
        if err <= 1e-12:
            self._no_overlap = False
            self._chunk_list = None
        return err, not self._no_overlap

    def step(self, T, events=None):
        E, prm, mat = self.engine, self.params, self.mat
        kappa = mat.k / (mat.rho * mat.cp)                       # adi3d_numba_coeff.py:292
        gam = kappa * prm.dt / (self.dx * self.dx)
        kind = 'torch' if isinstance(T, torch.Tensor) else ('numpy' if isinstance(T, np.ndarray) else 'field')
        Text = self._load_state(T)
        nxt = self._ext_bufs[self._cur ^ 1]
        A, B = self._tmp

        def mark(i):
            if events is not None:
                events[i].record()
        mark(0)
        # 1. state halos (zeros outside the global grid are never read: the flags carry no coupling there)
        # 2. explicit stage: the planes that do not touch a halo run while the halo planes are in flight
        lo = Text[0]; hi = Text[-1]
        self._chunks()
        nl = self.nxl
        overlap = self._use_streams and self.world > 1
        ev1 = None
        if overlap:
            main = torch.cuda.current_stream()
            ev0 = torch.cuda.Event(); ev0.record(main)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev0)                       # Text is complete
                self.comm.exchange_planes(Text[1], Text[-2], lo, hi)
                ev1 = torch.cuda.Event(); ev1.record(self._comm_stream)
        else:
            self.comm.exchange_planes(Text[1], Text[-2], lo, hi)
        ex = lambda b, e: E.explicit(self.Lext, Text, self.flags_ext, self.dx, prm.dt, kappa, prm.theta, A, b, e)
        if self.world > 1 and nl >= 4:
            ex(2, nl)                        # planes that touch no halo
            if ev1 is not None:
                main.wait_event(ev1)
            ex(1, 2)                         # first and last local plane need the neighbours' planes
            ex(nl, nl + 1)
        else:
            if ev1 is not None:
                main.wait_event(ev1)
            ex(1, nl + 1)
        mark(1)
        Ai, Bi, Oi = _interior(A), _interior(B), _interior(nxt)
        v, Li, fl = self.variant, self.Lint, self.flags_int
        # 3. distributed axis-0 sweep, pipelined over chunks of lines (ranges of j): the all-gather of chunk c
        #    runs on a second stream while chunk c+1 is condensed and chunk c-1 is solved
        if self.world > 1:
            self._axis0_distributed(Ai, Bi, gam)
        else:
            E.sweep(0, v, Li, Ai, fl, self.packs_int[0], prm.theta, gam, prm.dt, self.Tinf, Bi)
        mark(2)
        # 4. local sweeps
        E.sweep(1, v, Li, Bi, fl, self.packs_int[1], prm.theta, gam, prm.dt, self.Tinf, Ai)
        mark(3)
        E.sweep(2, v, Li, Ai, fl, self.packs_int[2], prm.theta, gam, prm.dt, self.Tinf, Oi)
        mark(4)
        self._cur ^= 1
        if kind == 'numpy':
            return Oi.cpu().contiguous().numpy()
        if kind == 'torch':
            return Oi
        return type(T)(Oi)
