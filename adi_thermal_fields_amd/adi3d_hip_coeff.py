"""adi3d_hip_coeff -- MI355X drop-in for the reference's Cartesian ADI backends.

Same operator surface as `adi3d_numba_coeff` / `adi3d_gpu_coeff` (the reference picks a backend by
module import, waam_from_stl_v7_mm.py:321-335), so a driver switches with

    import adi_thermal_fields_amd.adi3d_hip_coeff as adi

Names and meaning follow adi3d_numba_coeff.py:14-36, :38-55, :57-118, :290-302:
    Grid3D, Material, Params, AxisCoeffPack, exposed_mask, precompute_coeff_packs_unified,
    adi_step_hip_coeff  (also exported as adi_step_numba_coeff and adi_step_gpu_coeff).

Host code is Python; every number is computed by hand-written HIP kernels reached through the
ctypes C ABI of include/adi_hip.h.  PyTorch is used only for device memory and streams.
There is no CPU fallback.

Residency: a NumPy `Tn` is uploaded, stepped and downloaded (exact reference semantics: new array
out, input untouched).  Pass a `DeviceField` (see `to_device`) to keep the state in HBM across
steps -- the step then returns a new DeviceField, like the CuPy backend returns CuPy arrays.

Mask semantics: drivers rebind `grid.mask` and then rebuild the packs
(single_track_on_plate.py:159-163, waam_from_stl_v7_mm.py:494-495, :534).  The device copy of the
mask is refreshed on every `grid.mask = ...` assignment AND on every
precompute_coeff_packs_unified(grid, ...) call, which is the documented synchronisation point.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr_array, FACES

__all__ = ['Grid3D', 'Material', 'Params', 'AxisCoeffPack', 'exposed_mask', 'precompute_coeff_packs_unified',
           'adi_step_hip_coeff', 'adi_step_numba_coeff', 'adi_step_gpu_coeff', 'DeviceField', 'to_device',
           'adi_explicit_rhs', 'adi_sweep_axis', 'StagedStepper']


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("adi3d_hip_coeff needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _upload(a, dtype):
    """host array -> contiguous device tensor (bool masks travel as uint8)."""
    if isinstance(a, DeviceField):
        a = a.t
    if isinstance(a, torch.Tensor):
        t = a.to(device=_device())
        if dtype == torch.uint8 and t.dtype == torch.bool:
            t = t.view(torch.uint8) if t.is_contiguous() else t.contiguous().view(torch.uint8)
        return t.to(dtype).contiguous()
    arr = np.asarray(a)
    if dtype == torch.uint8:
        arr = np.ascontiguousarray(arr.astype(np.bool_, copy=False)).view(np.uint8)
    else:
        arr = np.ascontiguousarray(arr, dtype=np.float64)
    return torch.from_numpy(arr).to(_device())


class DeviceField:
    """A fp64 (nx, ny, nz) field resident in HBM.  Enough of the ndarray surface for the reference's
    drivers: indexing returns NumPy data, item assignment writes through, `np.asarray(f)` downloads."""

    def __init__(self, tensor):
        assert tensor.dtype == torch.float64 and tensor.is_cuda and tensor.is_contiguous()
        self.t = tensor

    shape = property(lambda self: tuple(self.t.shape))
    ndim = property(lambda self: self.t.dim())
    size = property(lambda self: self.t.numel())
    dtype = np.dtype(np.float64)

    def get(self):
        return self.t.cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def copy(self):
        return DeviceField(self.t.clone())

    def astype(self, dtype, copy=True):
        return self.get().astype(dtype, copy=False)

    @staticmethod
    def _idx(idx):
        def conv(i):
            if isinstance(i, np.ndarray):
                return torch.from_numpy(np.ascontiguousarray(i)).to(_device())
            return i
        return tuple(conv(i) for i in idx) if isinstance(idx, tuple) else conv(idx)

    def __getitem__(self, idx):
        r = self.t[self._idx(idx)]
        return r.item() if r.dim() == 0 else r.cpu().numpy()

    def __setitem__(self, idx, value):
        if isinstance(idx, np.ndarray) and idx.dtype == np.bool_ and idx.shape == self.shape and np.isscalar(value):
            sel = _upload(idx, torch.uint8)   # T[newborn] = Ts (waam_from_stl_v7_mm.py:491-493)
            check(lib.adi_masked_fill(_p(self.t), _p(sel), self.t.numel(), float(value), _stream()))
            return
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float64)).to(_device())
        elif isinstance(value, DeviceField):
            value = value.t
        self.t[self._idx(idx)] = value

    def min(self):
        return self.t.min().item()

    def max(self):
        return self.t.max().item()

    def sum(self):
        return self.t.sum().item()

    def mean(self):
        return self.t.mean().item()


def to_device(T):
    """NumPy (nx, ny, nz) field -> DeviceField (a copy, like the constructors of the reference)."""
    if isinstance(T, DeviceField):
        return T.copy()
    return DeviceField(_upload(T, torch.float64).clone() if isinstance(T, torch.Tensor) else _upload(T, torch.float64))


class Grid3D:
    """adi3d_numba_coeff.py:14-19.  `mask` is a property: assigning uploads it."""

    def __init__(self, nx, ny, nz, dx, mask):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.dx = float(dx)
        self._mask = None
        self._d_mask = None
        self._d_flags = None
        self._scratch = None
        self.mask_version = 0
        self.mask = np.asarray(mask).astype(np.bool_, copy=True, order='C')

    @property
    def shape(self):
        return (self.nx, self.ny, self.nz)

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, m):
        if isinstance(m, (torch.Tensor, DeviceField)):
            m = np.asarray(m.cpu() if isinstance(m, torch.Tensor) else m.get()).astype(np.bool_)
        m = np.asarray(m)
        assert m.shape == (self.nx, self.ny, self.nz)   # adi3d_numba_coeff.py:19
        self._mask = m                                   # rebinding keeps the caller's object, as in the reference
        self.sync_mask()

    def sync_mask(self):
        """(Re)upload the host mask; called on assignment and by precompute_coeff_packs_unified."""
        self._d_mask = _upload(self._mask, torch.uint8)
        self._d_flags = torch.empty_like(self._d_mask)
        check(lib.adi_build_nbr_flags(_p(self._d_mask), self.nx, self.ny, self.nz, _p(self._d_flags), _stream()))
        self.mask_version += 1
        return self._d_mask

    @property
    def d_mask(self):
        return self._d_mask

    @property
    def d_flags(self):
        """neighbour-flags digest of the mask (adi_build_nbr_flags), what the step kernels read"""
        return self._d_flags

    def scratch(self, n):
        """n cached scratch fields + the long-line workspace (None when not needed)."""
        if self._scratch is None or len(self._scratch[0]) < n or self._scratch[0][0].device != _device():
            fields = [torch.empty(self.shape, dtype=torch.float64, device=_device()) for _ in range(n)]
            wb = 0
            for ax in range(3):
                b = ctypes.c_size_t(0)
                check(lib.adi_sweep_workspace_bytes(ax, self.nx, self.ny, self.nz, ctypes.byref(b)))
                wb = max(wb, b.value)
            work = torch.empty(wb, dtype=torch.uint8, device=_device()) if wb else None
            self._scratch = (fields, work, wb)
        return self._scratch


class Material:  # adi3d_numba_coeff.py:21-23
    def __init__(self, rho, cp, k):
        self.rho = float(rho); self.cp = float(cp); self.k = float(k)


class Params:  # adi3d_numba_coeff.py:25-27
    def __init__(self, dt, theta=0.5):
        self.dt = float(dt); self.theta = float(theta)


class AxisCoeffPack:
    """adi3d_numba_coeff.py:29-36.  Arrays live in HBM (`d_*` tensors); the reference's attribute names
    `.coeff / .dir_mask / .dir_val / .qflux` return host copies (drivers read `packs[2].qflux`,
    quick_compare_neumann_robin.py:104)."""

    def __init__(self, coeff, dir_mask, dir_val, qflux=None, _has_dir=None, _has_q=None):
        self.d_coeff = _upload(coeff, torch.float64)
        self.d_dir_mask = None if dir_mask is None else _upload(dir_mask, torch.uint8)
        self.d_dir_val = None if dir_val is None else _upload(dir_val, torch.float64)
        self.d_qflux = None if qflux is None else _upload(qflux, torch.float64)
        if _has_dir is None:
            _has_dir = self.d_dir_mask is not None and bool(self.d_dir_mask.any().item())
        if _has_q is None:
            _has_q = self.d_qflux is not None and bool((self.d_qflux != 0).any().item())
        self.has_dir, self.has_q = bool(_has_dir), bool(_has_q)
        if self.has_dir and self.d_dir_val is None:
            self.d_dir_val = torch.zeros_like(self.d_coeff)

    @property
    def variant(self):
        if self.has_dir:
            return _lib.SWEEP_GENERAL if self.has_q else _lib.SWEEP_NO_Q
        return _lib.SWEEP_NO_DIR if self.has_q else _lib.SWEEP_LEAN

    @property
    def coeff(self):
        return self.d_coeff.cpu().numpy()

    @property
    def qflux(self):
        if self.d_qflux is None:
            return np.zeros(tuple(self.d_coeff.shape), dtype=np.float64)
        return self.d_qflux.cpu().numpy()

    @property
    def dir_mask(self):
        if self.d_dir_mask is None:
            return np.zeros(tuple(self.d_coeff.shape), dtype=np.bool_)
        return self.d_dir_mask.cpu().numpy().astype(np.bool_)

    @property
    def dir_val(self):
        if self.d_dir_val is None:
            return np.zeros(tuple(self.d_coeff.shape), dtype=np.float64)
        return self.d_dir_val.cpu().numpy()


def exposed_mask(mask, face):
    """adi3d_numba_coeff.py:38-55; ValueError("bad face") for an unknown face."""
    if face not in FACES:
        raise ValueError("bad face")
    host = not isinstance(mask, (torch.Tensor, DeviceField))
    d = _upload(mask, torch.uint8)
    assert d.dim() == 3
    out = torch.empty_like(d)
    nx, ny, nz = d.shape
    check(lib.adi_exposed_mask(_p(d), nx, ny, nz, FACES.index(face), _p(out), _stream()))
    return out.cpu().numpy().astype(np.bool_) if host else out.to(torch.bool)


def _face_spec(spec, shape, keep):
    """scalar / array / None -> (mode, scalar, device tensor or None)"""
    if spec is None:
        return (_lib.FACE_NONE, 0.0, None)
    if np.isscalar(spec):
        return (_lib.FACE_SCALAR, float(spec), None)
    t = _upload(spec, torch.float64)
    assert tuple(t.shape) == shape
    keep.append(t)
    return (_lib.FACE_FIELD, 0.0, t)


def precompute_coeff_packs_unified(grid, mat, dir_mask=None, dir_value=None, neumann=None,
                                   robin_h=None, robin_Tinf=None):
    """adi3d_numba_coeff.py:57-118: one HIP pass builds the Robin coefficient and Neumann flux fields
    of the three axes on the device.  `robin_Tinf` is accepted and ignored, as in the reference (the
    ambient enters at step time)."""
    shape = grid.shape
    d_mask = grid.sync_mask()
    keep = []
    h_specs, q_specs = [], []
    for f in FACES:
        if robin_h is None:
            h_specs.append((_lib.FACE_NONE, 0.0, None))
        elif isinstance(robin_h, dict):
            h_specs.append(_face_spec(robin_h.get(f, 0.0), shape, keep))
        else:
            h_specs.append(_face_spec(robin_h, shape, keep))
        q_specs.append(_face_spec(neumann.get(f) if neumann is not None else None, shape, keep))
    if neumann is not None:
        for f in neumann:
            if f not in FACES:
                raise ValueError("bad face")   # exposed_mask(grid.mask, f) raises in the reference (:106)

    dev = _device()
    coeff = [torch.empty(shape, dtype=torch.float64, device=dev) for _ in range(3)]
    qflux = [torch.empty(shape, dtype=torch.float64, device=dev) for _ in range(3)]
    hm = (ctypes.c_int * 6)(*[s[0] for s in h_specs])
    hs = (ctypes.c_double * 6)(*[s[1] for s in h_specs])
    hf = ptr_array([s[2].data_ptr() if s[2] is not None else None for s in h_specs])
    qm = (ctypes.c_int * 6)(*[s[0] for s in q_specs])
    qs = (ctypes.c_double * 6)(*[s[1] for s in q_specs])
    qf = ptr_array([s[2].data_ptr() if s[2] is not None else None for s in q_specs])
    check(lib.adi_build_coeffs(_p(d_mask), grid.nx, grid.ny, grid.nz, grid.dx, mat.rho, mat.cp,
                               hm, hs, hf, qm, qs, qf,
                               ptr_array([c.data_ptr() for c in coeff]), ptr_array([q.data_ptr() for q in qflux]),
                               _stream()))
    has_q = any(s[0] != _lib.FACE_NONE for s in q_specs)
    d_dm = d_dv = None
    has_dir = False
    if dir_mask is not None:
        d_dm = _upload(dir_mask, torch.uint8)
        assert tuple(d_dm.shape) == shape
        has_dir = bool(d_dm.any().item())
        if dir_value is None:
            d_dv = torch.zeros(shape, dtype=torch.float64, device=dev)          # :75-76
        elif np.isscalar(dir_value):
            d_dv = torch.full(shape, float(dir_value), dtype=torch.float64, device=dev)  # :77-78
        else:
            d_dv = _upload(dir_value, torch.float64)
    packs = tuple(AxisCoeffPack(coeff[a], d_dm, d_dv, qflux[a], _has_dir=has_dir, _has_q=has_q) for a in range(3))
    for p in packs:
        p.mask_version = grid.mask_version
    return packs


def _gam(grid, mat, params):
    kappa = mat.k / (mat.rho * mat.cp)                    # adi3d_numba_coeff.py:292
    return kappa, kappa * params.dt / (grid.dx * grid.dx)


def _as_state(Tn, grid):
    """-> (device tensor fp64 contiguous, kind) with kind in {'numpy', 'field', 'torch'}"""
    if isinstance(Tn, DeviceField):
        t, kind = Tn.t, 'field'
    elif isinstance(Tn, torch.Tensor):
        t, kind = Tn.to(device=_device(), dtype=torch.float64).contiguous(), 'torch'
    else:
        t, kind = _upload(np.asarray(Tn), torch.float64), 'numpy'   # fp32 fields are up-cast (waam --precision float32)
    assert tuple(t.shape) == grid.shape
    return t, kind


def _wrap(t, kind):
    if kind == 'field':
        return DeviceField(t)
    if kind == 'torch':
        return t
    return t.cpu().numpy()


def adi_explicit_rhs(Tn, grid, mat, params):
    """R0 of adi3d_numba_coeff.py:292-298 (stage entry point for per-stage parity tests / benchmarks)."""
    t, kind = _as_state(Tn, grid)
    kappa, _ = _gam(grid, mat, params)
    out = torch.empty_like(t)
    check(lib.adi_explicit_rhs(_p(t), _p(grid.d_flags), grid.nx, grid.ny, grid.nz, grid.dx, params.dt, kappa,
                               params.theta, _p(out), _stream()))
    return _wrap(out, kind)


def _sweep_into(axis, t_in, t_out, grid, mat, params, pack, Tinf, variant=None):
    _, gam = _gam(grid, mat, params)
    _, work, wb = grid.scratch(2)
    v = pack.variant if variant is None else variant
    check(lib.adi_sweep(axis, v, _p(t_in), _p(grid.d_flags), _p(pack.d_coeff), _p(pack.d_dir_mask),
                        _p(pack.d_dir_val), _p(pack.d_qflux), grid.nx, grid.ny, grid.nz,
                        params.theta, gam, params.dt, float(Tinf), _p(t_out), _p(work), wb, _stream()))


def adi_sweep_axis(axis, stage_in, grid, mat, params, pack, Tinf=0.0, variant=None):
    """sweep_axis0/1/2 of adi3d_numba_coeff.py:133-237 for one axis (stage entry point).
    variant=None picks the leanest kernel the pack allows; pass _lib.SWEEP_GENERAL to force the
    42 B/cell general-pack kernel."""
    t, kind = _as_state(stage_in, grid)
    if variant == _lib.SWEEP_GENERAL or (variant is None and pack.variant == _lib.SWEEP_GENERAL):
        _ensure_general(pack)
    out = torch.empty_like(t)
    _sweep_into(axis, t, out, grid, mat, params, pack, Tinf, variant)
    return _wrap(out, kind)


def _ensure_general(pack):
    """materialise the arrays a forced general-pack sweep reads (zeros, like the reference's packs)"""
    if pack.d_dir_mask is None:
        pack.d_dir_mask = torch.zeros(tuple(pack.d_coeff.shape), dtype=torch.uint8, device=pack.d_coeff.device)
    if pack.d_dir_val is None:
        pack.d_dir_val = torch.zeros_like(pack.d_coeff)
    if pack.d_qflux is None:
        pack.d_qflux = torch.zeros_like(pack.d_coeff)


def adi_step_hip_coeff(Tn, grid, mat, params, packs, Tinf=0.0):
    """adi3d_numba_coeff.py:290-302 / adi3d_gpu_coeff.py:213-230: explicit stage, then the three
    implicit sweeps in the order axis 0, 1, 2.  Returns a NEW array of the kind it was given;
    `Tn` is never modified."""
    t, kind = _as_state(Tn, grid)
    packx, packy, packz = packs
    (ta, tb), _, _ = grid.scratch(2)
    kappa, _ = _gam(grid, mat, params)
    out = torch.empty_like(t)
    check(lib.adi_explicit_rhs(_p(t), _p(grid.d_flags), grid.nx, grid.ny, grid.nz, grid.dx, params.dt, kappa,
                               params.theta, _p(ta), _stream()))
    _sweep_into(0, ta, tb, grid, mat, params, packx, Tinf)
    _sweep_into(1, tb, ta, grid, mat, params, packy, Tinf)
    _sweep_into(2, ta, out, grid, mat, params, packz, Tinf)
    return _wrap(out, kind)


# the reference's backend-specific names, so its drivers run unchanged on this module
adi_step_numba_coeff = adi_step_hip_coeff
adi_step_gpu_coeff = adi_step_hip_coeff


class StagedStepper:
    """The step of adi_step_hip_coeff with its arguments resolved once, for tight loops over a
    device-resident field (drivers call the step `nsub` times with the same packs and dt,
    quick_compare_dirichlet_robin.py:169-178).  `events`: optional list of 5 torch.cuda.Event recorded on
    the launch stream before/between/after the four stage kernels (per-stage HIP-event timing)."""

    stage_names = ['explicit', 'sweep_axis0', 'sweep_axis1', 'sweep_axis2_contig']

    def __init__(self, grid, mat, params, packs, Tinf=0.0):
        self.grid, self.mat, self.params, self.packs, self.Tinf = grid, mat, params, packs, float(Tinf)
        self.stage_bytes_per_cell = [_lib.EXPLICIT_BYTES_PER_CELL] + \
            [_lib.SWEEP_BYTES_PER_CELL[p.variant] for p in packs]

    def sweep_into(self, axis, t_in, t_out, variant=None):
        if variant == _lib.SWEEP_GENERAL:
            _ensure_general(self.packs[axis])
        _sweep_into(axis, t_in, t_out, self.grid, self.mat, self.params, self.packs[axis], self.Tinf, variant)

    def step(self, T, events=None):
        g, prm = self.grid, self.params
        t = T.t
        (ta, tb), _, _ = g.scratch(2)
        kappa, _ = _gam(g, self.mat, prm)
        out = torch.empty_like(t)
        if events is not None:
            events[0].record()
        check(lib.adi_explicit_rhs(_p(t), _p(g.d_flags), g.nx, g.ny, g.nz, g.dx, prm.dt, kappa, prm.theta,
                                   _p(ta), _stream()))
        if events is not None:
            events[1].record()
        self.sweep_into(0, ta, tb)
        if events is not None:
            events[2].record()
        self.sweep_into(1, tb, ta)
        if events is not None:
            events[3].record()
        self.sweep_into(2, ta, out)
        if events is not None:
            events[4].record()
        return DeviceField(out)
