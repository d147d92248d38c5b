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

Device layout: fields live in HBM with a padded plane stride (`Layout.sx`, chosen by
adi_recommended_plane_stride) so that the rows of an axis-0 line do not alias on the HBM channel
interleave; the C-order (nx, ny, nz) shape of the reference is what every host-visible view has.

Mask semantics: drivers rebind `grid.mask` and then rebuild the packs
(single_track_on_plate.py:159-163, waam_from_stl_v7_mm.py:494-495, :534).  The device copy of the
mask (and its neighbour-flags digest) is refreshed on every `grid.mask = ...` assignment AND on every
precompute_coeff_packs_unified(grid, ...) call, which is the documented synchronisation point.
"""
import ctypes
import itertools

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr_array, FACES

__all__ = ['Grid3D', 'Material', 'Params', 'AxisCoeffPack', 'exposed_mask', 'precompute_coeff_packs_unified',
           'adi_step_hip_coeff', 'adi_step_numba_coeff', 'adi_step_gpu_coeff', 'DeviceField', 'to_device',
           'adi_explicit_rhs', 'adi_sweep_axis', 'StagedStepper', 'Layout', 'apply_surface_impulse_Q',
           'exposed_faces_per_layer', 'count_exposed_faces', 'perimeter_ratio', 'birth_planes', 'BirthPacks']


# Mask versions come from ONE process-wide counter: a pack remembers the version of the mask it was built for, and a
# version is never shared by two grids (a per-grid counter starting at 0 let packs of grid A pass for fresh on a
# same-shape grid B and inherit A's no-fallback promise).
_MASK_VERSIONS = itertools.count(1)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("adi3d_hip_coeff needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def recommended_dims(nx, ny, nz):
    """physical extents the kernels want for a logical (nx, ny, nz) grid (adi_recommended_dims)"""
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(lib.adi_recommended_dims(int(nx), int(ny), int(nz), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


class Layout:
    """Device layout of a (nx, ny, nz) grid: element (i, j, k) at i*sx + j*pz + k inside a PHYSICAL box (px, py, pz) >= the
    logical one.  Ragged extents (257 rows, nz = 250) would send every line to the GENERAL kernels, so fields are allocated
    with the extents adi_recommended_dims() picks; the cells outside the logical box are off-mask (identity rows, never
    read by an in-mask cell: the reference treats the edge of the domain and an off-mask neighbour alike,
    adi3d_numba_coeff.py:38-55), the kernels are launched on the physical box (`pd`), and everything the caller sees --
    shapes, NumPy arrays, DeviceField indexing -- is the logical box.  Layouts made with an explicit plane stride `sx`
    (slabs, views of foreign tensors) are never padded unless `phys` says so."""

    def __init__(self, nx, ny, nz, sx=None, phys=None):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        if phys is None:
            phys = recommended_dims(nx, ny, nz) if sx is None else (self.nx, self.ny, self.nz)
        self.px, self.py, self.pz = (int(v) for v in phys)
        assert self.px >= self.nx and self.py >= self.ny and self.pz >= self.nz
        self.sx = int(lib.adi_recommended_plane_stride(self.py, self.pz)) if sx is None else int(sx)
        assert self.sx >= self.py * self.pz

    @property
    def shape(self):
        return (self.nx, self.ny, self.nz)

    @property
    def strides(self):
        return (self.sx, self.pz, 1)

    @property
    def pd(self):
        """(nx, ny, nz, plane_stride) as the C ABI takes them: the physical box"""
        return (self.px, self.py, self.pz, self.sx)

    @property
    def padded(self):
        return (self.px, self.py, self.pz) != (self.nx, self.ny, self.nz)

    @property
    def numel_padded(self):
        return self.px * self.sx

    def empty(self, dtype=torch.float64, zero=False):
        buf = (torch.zeros if zero else torch.empty)(self.numel_padded, dtype=dtype, device=_device())
        return buf.as_strided(self.shape, self.strides)

    def is_native(self, t):
        return (isinstance(t, torch.Tensor) and t.is_cuda and tuple(t.shape) == self.shape
                and tuple(t.stride()) == self.strides and t.storage_offset() == 0
                and t.untyped_storage().nbytes() >= self.numel_padded * t.element_size())

    def _fresh(self, dtype):
        # masks / flags are read over the whole physical box and the plane padding; fields over the physical box
        return self.empty(dtype, zero=(self.padded or (dtype == torch.uint8 and self.sx != self.ny * self.nz)))

    def to_layout(self, a, dtype):
        """host array / tensor / DeviceField -> tensor in this layout (a copy unless already native)."""
        if isinstance(a, DeviceField):
            a = a.t
        if isinstance(a, torch.Tensor):
            if self.is_native(a) and a.dtype == dtype:
                return a
            src = a.to(device=_device())
            if dtype == torch.uint8 and src.dtype == torch.bool:
                src = src.to(torch.uint8)
            assert tuple(src.shape) == self.shape, (tuple(src.shape), self.shape)
            out = self._fresh(dtype)
            out.copy_(src.to(dtype) if src.dtype != dtype else src)
            return out
        arr = np.asarray(a)
        if dtype == torch.uint8:
            arr = np.ascontiguousarray(arr.astype(np.bool_, copy=False)).view(np.uint8)
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float64)   # fp32 fields are up-cast (waam --precision float32)
        assert tuple(arr.shape) == self.shape, (tuple(arr.shape), self.shape)
        out = self._fresh(dtype)
        es = arr.itemsize
        if self.pz == self.nz:
            # every plane of the dense host array into its (padded) device plane: ONE 2-D DMA, no staging tensor
            check(lib.adi_copy_planes(_p(out), self.sx * es, ctypes.c_void_p(arr.ctypes.data), self.ny * self.nz * es,
                                      self.ny * self.nz * es, self.nx, 1, _stream()))
        else:
            # padded rows: one dense DMA into a staging tensor, rows spread on the device
            stage = torch.empty(self.shape, dtype=dtype, device=_device())
            n = arr.size * es
            check(lib.adi_copy_planes(_p(stage), n, ctypes.c_void_p(arr.ctypes.data), n, n, 1, 1, _stream()))
            out.copy_(stage)
        torch.cuda.current_stream().synchronize()     # the caller may touch its array as soon as we return
        return out

    def to_host(self, t):
        """native-layout device tensor -> C-order NumPy array (a new array, as the reference's step returns one).  The
        array lives in page-locked memory from torch's caching host allocator: the transfer is one 2-D DMA at PCIe
        rate, and in a driver's `T = step(T, ...)` loop the block of the array dropped a step ago is reused, so no
        fresh pages are faulted in (1 GiB of first-touch page faults cost more than the transfer itself)."""
        assert self.is_native(t)
        host = torch.empty(self.shape, dtype=t.dtype, pin_memory=True)
        es = t.element_size()
        if self.pz == self.nz:
            check(lib.adi_copy_planes(ctypes.c_void_p(host.data_ptr()), self.ny * self.nz * es, _p(t), self.sx * es,
                                      self.ny * self.nz * es, self.nx, 0, _stream()))
        else:
            stage = t.contiguous()                    # padded rows: gathered on the device, one dense DMA
            n = stage.numel() * es
            check(lib.adi_copy_planes(ctypes.c_void_p(host.data_ptr()), n, _p(stage), n, n, 1, 0, _stream()))
        torch.cuda.current_stream().synchronize()
        return host.numpy()

    @staticmethod
    def of(t):
        """the layout a 3-D device tensor is in, or None when it is not of this family (rows contiguous, offset 0)"""
        if not (isinstance(t, torch.Tensor) and t.dim() == 3 and t.storage_offset() == 0 and t.stride(2) == 1
                and t.stride(1) >= t.shape[2] and t.stride(0) >= t.stride(1) * t.shape[1]):
            return None
        nx, ny, nz = t.shape
        sx, pz = t.stride(0), t.stride(1)
        py = max(ny, min(sx // pz, recommended_dims(nx, ny, nz)[1])) if pz else ny
        px = max(nx, t.untyped_storage().nbytes() // t.element_size() // sx) if sx else nx
        return Layout(nx, ny, nz, sx=sx, phys=(px, py, pz))


class DeviceField:
    """A fp64 (nx, ny, nz) field resident in HBM.  Enough of the ndarray surface for the reference's
    drivers: indexing returns NumPy data, item assignment writes through, `np.asarray(f)` downloads."""

    def __init__(self, tensor):
        assert tensor.dtype == torch.float64 and tensor.is_cuda and tensor.dim() == 3
        self.t = tensor

    shape = property(lambda self: tuple(self.t.shape))
    ndim = property(lambda self: self.t.dim())
    size = property(lambda self: self.t.numel())
    dtype = np.dtype(np.float64)

    def get(self):
        L = Layout.of(self.t)
        if L is not None and L.is_native(self.t):
            return L.to_host(self.t)
        return self.t.cpu().contiguous().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def copy(self):
        if self.t.is_contiguous() or self.t.storage_offset() != 0:
            return DeviceField(self.t.clone())
        # the whole storage: plane padding and the cells of the physical box outside the logical one travel along
        n = self.t.untyped_storage().nbytes() // self.t.element_size()
        flat = torch.empty(n, dtype=self.t.dtype, device=self.t.device)
        flat.copy_(self.t.as_strided((n,), (1,)))
        return DeviceField(flat.as_strided(self.t.shape, self.t.stride()))

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
        # T[newborn] = Ts / T[idx] = Ts (waam_from_stl_v7_mm.py:489-493) and general slices
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float64)).to(_device())
        elif isinstance(value, DeviceField):
            value = value.t
        self.t[self._idx(idx)] = value

    def fill_where(self, d_sel_native, value):
        """T[sel] = value with `sel` a uint8 tensor in the same device layout (HIP kernel, no host round trip)."""
        assert tuple(d_sel_native.stride()) == tuple(self.t.stride())
        n = self.t.shape[0] * self.t.stride(0)
        check(lib.adi_masked_fill(_p(self.t), _p(d_sel_native), n, float(value), _stream()))

    def min(self):
        return self.t.min().item()

    def max(self):
        return self.t.max().item()

    def sum(self):
        return self.t.sum().item()

    def mean(self):
        return self.t.mean().item()


def to_device(T):
    """NumPy (nx, ny, nz) field -> DeviceField in the library's device layout (a copy)."""
    if isinstance(T, DeviceField):
        return T.copy()
    t = Layout(*tuple(T.shape)).to_layout(T, torch.float64)
    return DeviceField(t).copy() if (isinstance(T, torch.Tensor) and t is T) else DeviceField(t)


class Grid3D:
    """adi3d_numba_coeff.py:14-19.  `mask` is a property: assigning uploads it."""

    def __init__(self, nx, ny, nz, dx, mask):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.dx = float(dx)
        self.layout = Layout(self.nx, self.ny, self.nz)
        self._mask = None
        self._d_mask = None
        self._d_flags = None
        self._all_solid = None
        self._scratch = None
        self.mask_version = next(_MASK_VERSIONS)
        self.mask = np.asarray(mask).astype(np.bool_, copy=True, order='C')

    @property
    def shape(self):
        return (self.nx, self.ny, self.nz)

    @property
    def sx(self):
        return self.layout.sx

    @property
    def mask(self):
        if self._mask is None and self._d_mask is not None:       # device-side mask (set_mask_device): download on demand
            self._mask = self._d_mask.cpu().contiguous().numpy().astype(np.bool_)
        return self._mask

    @mask.setter
    def mask(self, m):
        if isinstance(m, (torch.Tensor, DeviceField)):
            m = np.asarray(m.cpu() if isinstance(m, torch.Tensor) else m.get()).astype(np.bool_)
        m = np.asarray(m)
        assert m.shape == (self.nx, self.ny, self.nz)   # adi3d_numba_coeff.py:19
        self._mask = m                                   # rebinding keeps the caller's object, as in the reference
        self.sync_mask()

    def _rebuild_flags(self, k_begin=0, k_end=None):
        """neighbour flags of the device mask, in place (the buffer is zero-filled once: plane padding)"""
        if self._d_flags is None:
            self._d_flags = self.layout.empty(torch.uint8, zero=True)
        k_end = self.layout.pz if k_end is None else k_end
        check(lib.adi_build_nbr_flags_planes(_p(self._d_mask), *self.layout.pd, _p(self._d_flags),
                                             int(k_begin), int(k_end), _stream()))
        self.mask_version = next(_MASK_VERSIONS)
        self._all_solid = None

    @property
    def all_solid(self):
        """hint for the kernels (no surface inside the box); evaluated on demand -- it costs a host synchronisation"""
        if self._all_solid is None:
            self._all_solid = (not self.layout.padded and bool((self._d_mask != 0).all().item())) \
                if self._d_mask is not None else False       # (a padded box has off-mask cells: never "all solid")
        return self._all_solid

    @all_solid.setter
    def all_solid(self, v):
        self._all_solid = None if v is None else (bool(v) and not self.layout.padded)

    def set_mask_device(self, d_mask, k_begin=0, k_end=None, all_solid=None):
        """`grid.mask = ...` for a mask that already lives on the device (uint8 tensor in the grid's layout, e.g.
        updated in place by a birth): no host round trip; the neighbour flags are rebuilt from it (only the planes
        [k_begin, k_end) of axis 2 when the caller knows nothing else changed) and the host copy is downloaded only if
        somebody reads `grid.mask`.  `all_solid`: the caller's knowledge of the box hint (None: evaluated on demand).
        The device loop of waam.run_layer_birth uses this."""
        assert self.layout.is_native(d_mask) and d_mask.dtype == torch.uint8
        self._d_mask = d_mask
        self._mask = None
        self._device_mask = True
        self._rebuild_flags(k_begin, k_end)
        self._all_solid = None if all_solid is None else (bool(all_solid) and not self.layout.padded)
        return self._d_mask

    def sync_mask(self):
        """(Re)upload the host mask and rebuild its neighbour-flags digest; called on assignment and by
        precompute_coeff_packs_unified.  A device-side mask (set_mask_device) is authoritative as it is."""
        if getattr(self, '_device_mask', False) and self._mask is None:
            return self._d_mask
        self._device_mask = False
        new = self.layout.to_layout(self._mask, torch.uint8)
        if self._d_mask is not None and self._d_flags is not None and tuple(new.shape) == tuple(self._d_mask.shape) \
                and torch.equal(new, self._d_mask):
            # the same mask again (a second precompute_coeff_packs_unified on an unchanged grid: Dirichlet-vs-Robin
            # comparisons, packs rebuilt with another h): flags and version stand, packs built earlier stay fresh
            return self._d_mask
        self._d_mask = new
        self._rebuild_flags()
        self._all_solid = bool(np.asarray(self._mask).all()) and not self.layout.padded   # hint for the kernels: no surface inside the (physical) box
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
            # The scratch fields sit 2 KiB / 4 KiB past a 2 MiB boundary, not on it: a sweep whose input and output are both
            # 2 MiB-aligned (what separate allocations give) runs into the same channels with its reads and its writes.  Same
            # stepper, scratch as allocated against skewed, alternated (scripts/skew_probe.py, profiles/r04_q_skew_probe.txt):
            # axis-1 sweep 0.418 -> 0.409 ms, and the fused kernel's launch-to-launch spread 0.0077 -> 0.0009 ms.
            L = self.layout
            fields = []
            for i in range(n):
                skew = 256 * (i + 1)
                raw = torch.empty(L.numel_padded + skew, dtype=torch.float64, device=_device())
                fields.append(raw[skew:skew + L.numel_padded].as_strided(L.shape, L.strides))
            wb = 0
            for ax in range(3):
                b = ctypes.c_size_t(0)
                check(lib.adi_sweep_workspace_bytes(ax, *self.layout.pd, ctypes.byref(b)))
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
    """adi3d_numba_coeff.py:29-36.  Arrays live in HBM (`d_*` tensors, device layout); the reference's
    attribute names `.coeff / .dir_mask / .dir_val / .qflux` return host copies (drivers read
    `packs[2].qflux`, quick_compare_neumann_robin.py:104)."""

    def __init__(self, coeff, dir_mask, dir_val, qflux=None, _has_dir=None, _has_q=None, _layout=None):
        self.layout = _layout or Layout(*tuple(coeff.shape))
        L = self.layout
        self.d_coeff = L.to_layout(coeff, torch.float64)
        self.d_dir_mask = None if dir_mask is None else L.to_layout(dir_mask, torch.uint8)
        self.d_dir_val = None if dir_val is None else L.to_layout(dir_val, torch.float64)
        self.d_qflux = None if qflux is None else L.to_layout(qflux, torch.float64)
        if _has_dir is None:
            _has_dir = self.d_dir_mask is not None and bool(self.d_dir_mask.any().item())
        if _has_q is None:
            _has_q = self.d_qflux is not None and bool((self.d_qflux != 0).any().item())
        self.has_dir, self.has_q = bool(_has_dir), bool(_has_q)
        if self.has_dir and self.d_dir_val is None:
            self.d_dir_val = L.empty(zero=True)
        # Set by precompute_coeff_packs_unified only: coeff/qflux are non-zero just on cells exposed along the
        # pack's axis, so the sweep may skip loading them elsewhere.  Hand-built packs are read densely.
        self.sparse_ok = False
        self.face_consts = None        # (c-, c+, q-, q+) when built from per-face scalars (precompute_coeff_packs_unified)
        self._fractions = None         # (grid, axis, mask version) the byte accounting is evaluated from, on demand
        self._exposed_fraction = 1.0   # fraction of cells exposed along the axis
        self._dir_fraction = 1.0

    def _eval_fractions(self):
        if self._fractions is not None:
            grid, a, ver = self._fractions
            self._fractions = None
            if grid.mask_version != ver:       # the mask moved on before anybody asked: keep the dense figures
                return
            fl = grid.d_flags
            L = self.layout
            ncell = float(L.nx * L.ny * L.nz)
            inmask = (fl & 1) == 1
            self._exposed_fraction = float((inmask & (((fl >> (1 + 2 * a)) & 3) != 3)).sum().item()) / ncell
            self._dir_fraction = float(self.d_dir_mask.sum().item()) / ncell if self.has_dir else 0.0

    @property
    def exposed_fraction(self):
        self._eval_fractions()
        return self._exposed_fraction

    @property
    def dir_fraction(self):
        self._eval_fractions()
        return self._dir_fraction

    @property
    def variant(self):
        if self.has_dir:
            return _lib.SWEEP_GENERAL if self.has_q else _lib.SWEEP_NO_Q
        return _lib.SWEEP_NO_DIR if self.has_q else _lib.SWEEP_LEAN

    def _host(self, t, dtype):
        if t is None:
            return np.zeros(self.layout.shape, dtype=dtype)
        return t.cpu().contiguous().numpy().astype(dtype, copy=False)

    @property
    def bytes_per_cell(self):
        """HBM bytes per cell the sweep of this pack must move (its inputs + the output): the byte count of
        the roofline (SURVEY.md 8(d) variant rule)."""
        fe = self.exposed_fraction if self.sparse_ok else 1.0
        if self.sparse_ok and self.face_consts is not None:
            fe = 0.0                                      # per-face scalars: coeff / qflux are not read at all
        b = 8.0 + 1.0 + 8.0 + 8.0 * fe                    # in, flags, out, coeff
        if self.has_q:
            b += 8.0 * fe
        if self.has_dir:
            b += 1.0 + 8.0 * (self.dir_fraction if self.sparse_ok else 1.0)
        return b

    coeff = property(lambda self: self._host(self.d_coeff, np.float64))
    qflux = property(lambda self: self._host(self.d_qflux, np.float64))
    dir_mask = property(lambda self: self._host(self.d_dir_mask, np.bool_))
    dir_val = property(lambda self: self._host(self.d_dir_val, np.float64))


def exposed_mask(mask, face):
    """adi3d_numba_coeff.py:38-55; ValueError("bad face") for an unknown face."""
    if face not in FACES:
        raise ValueError("bad face")
    host = not isinstance(mask, (torch.Tensor, DeviceField))
    shape = tuple(mask.shape)
    assert len(shape) == 3
    L = Layout(*shape, sx=shape[1] * shape[2])
    d = L.to_layout(mask, torch.uint8)
    out = L.empty(torch.uint8)
    check(lib.adi_exposed_mask(_p(d), shape[0], shape[1], shape[2], 0, FACES.index(face), _p(out), _stream()))
    return out.cpu().numpy().astype(np.bool_) if host else out.to(torch.bool)


def _face_spec(spec, L, keep):
    """scalar / array / None -> (mode, scalar, device tensor or None)"""
    if spec is None:
        return (_lib.FACE_NONE, 0.0, None)
    if np.isscalar(spec):
        return (_lib.FACE_SCALAR, float(spec), None)
    t = L.to_layout(spec, torch.float64)
    keep.append(t)
    return (_lib.FACE_FIELD, 0.0, t)


def _face_constants(grid, mat, h_modes, h_scalars, q_modes, q_scalars):
    """per axis (c-, c+, q-, q+) as a ctypes array of 4 doubles, or None where a face of the axis carries a per-voxel field:
    what the sweeps take as `h_face_consts` instead of loading coeff / qflux at the exposed cells (adi_face_constants)"""
    consts = (ctypes.c_double * 12)()
    valid = (ctypes.c_int * 3)()
    check(lib.adi_face_constants(grid.dx, mat.rho, mat.cp, (ctypes.c_int * 6)(*h_modes), (ctypes.c_double * 6)(*h_scalars),
                                 (ctypes.c_int * 6)(*q_modes), (ctypes.c_double * 6)(*q_scalars), consts, valid))
    return [(ctypes.c_double * 4)(*consts[4 * a:4 * a + 4]) if valid[a] else None for a in range(3)]


def _fc_arg(grid, pack, sp):
    """the h_face_consts argument for a sweep of `pack` under the `sparse` word `sp`: only with sparse reads (fresh packs)"""
    fc = getattr(pack, 'face_consts', None)
    return fc if (fc is not None and (sp & 1)) else None


def precompute_coeff_packs_unified(grid, mat, dir_mask=None, dir_value=None, neumann=None,
                                   robin_h=None, robin_Tinf=None):
    """adi3d_numba_coeff.py:57-118: one HIP pass builds the Robin coefficient and Neumann flux fields
    of the three axes on the device.  `robin_Tinf` is accepted and ignored, as in the reference (the
    ambient enters at step time)."""
    L = grid.layout
    d_mask = grid.sync_mask()
    keep = []
    h_specs, q_specs = [], []
    for f in FACES:
        if robin_h is None:
            h_specs.append((_lib.FACE_NONE, 0.0, None))
        elif isinstance(robin_h, dict):
            h_specs.append(_face_spec(robin_h.get(f, 0.0), L, keep))
        else:
            h_specs.append(_face_spec(robin_h, L, keep))
        q_specs.append(_face_spec(neumann.get(f) if neumann is not None else None, L, keep))
    if neumann is not None:
        for f in neumann:
            if f not in FACES:
                raise ValueError("bad face")   # exposed_mask(grid.mask, f) raises in the reference (:106)

    coeff = [L.empty() for _ in range(3)]
    qflux = [L.empty() for _ in range(3)]
    hm = (ctypes.c_int * 6)(*[s[0] for s in h_specs])
    hs = (ctypes.c_double * 6)(*[s[1] for s in h_specs])
    hf = ptr_array([s[2].data_ptr() if s[2] is not None else None for s in h_specs])
    qm = (ctypes.c_int * 6)(*[s[0] for s in q_specs])
    qs = (ctypes.c_double * 6)(*[s[1] for s in q_specs])
    qf = ptr_array([s[2].data_ptr() if s[2] is not None else None for s in q_specs])
    check(lib.adi_build_coeffs(_p(d_mask), *grid.layout.pd, grid.dx, mat.rho, mat.cp,
                               hm, hs, hf, qm, qs, qf,
                               ptr_array([c.data_ptr() for c in coeff]), ptr_array([q.data_ptr() for q in qflux]),
                               _stream()))
    has_q = any(s[0] != _lib.FACE_NONE for s in q_specs)
    d_dm = d_dv = None
    has_dir = False
    if dir_mask is not None:
        d_dm = L.to_layout(dir_mask, torch.uint8)
        has_dir = bool(d_dm.any().item())
        if dir_value is None:
            d_dv = L.empty(zero=True)                                   # :75-76
        elif np.isscalar(dir_value):
            d_dv = L.empty(zero=L.padded)
            d_dv.fill_(float(dir_value))                                # :77-78
        else:
            d_dv = L.to_layout(dir_value, torch.float64)
    packs = tuple(AxisCoeffPack(coeff[a], d_dm, d_dv, qflux[a], _has_dir=has_dir, _has_q=has_q, _layout=L)
                  for a in range(3))
    fcs = _face_constants(grid, mat, [s[0] for s in h_specs], [s[1] for s in h_specs], [s[0] for s in q_specs],
                          [s[1] for s in q_specs])
    for a, p in enumerate(packs):
        p.mask_version = grid.mask_version
        p.sparse_ok = True
        p.face_consts = fcs[a]                           # per-face scalars: the sweeps need not load coeff / qflux
        p._fractions = (grid, a, grid.mask_version)      # exposed / Dirichlet fractions: evaluated when somebody asks
    return packs


def _gam(grid, mat, params):
    kappa = mat.k / (mat.rho * mat.cp)                    # adi3d_numba_coeff.py:292
    return kappa, kappa * params.dt / (grid.dx * grid.dx)


def _as_state(Tn, grid):
    """-> (device tensor in the grid's layout, kind) with kind in {'numpy', 'field', 'torch'}"""
    kind = 'field' if isinstance(Tn, DeviceField) else ('torch' if isinstance(Tn, torch.Tensor) else 'numpy')
    if kind == 'numpy':
        Tn = np.asarray(Tn)
        # Called exactly like the reference (host arrays in, host array out) the step also READS grid.mask like the
        # reference does (adi3d_numba_coeff.py:294-301 pass grid.mask to every stage): the host mask is re-uploaded and
        # compared with the device copy, so a mask mutated in place since the last assignment / pack build is seen
        # (1 B/cell next to the 16 B/cell this path moves anyway; an unchanged mask keeps flags, version and packs).
        # Device-resident stepping (DeviceField / tensor state) is the opt-in fast path: there `grid.mask = ...`,
        # grid.sync_mask() or a pack rebuild is the synchronisation point.
        if getattr(grid, '_mask', None) is not None and hasattr(grid, 'sync_mask'):
            grid.sync_mask()
    assert tuple(Tn.shape) == grid.shape
    return grid.layout.to_layout(Tn, torch.float64), kind


def _wrap(t, kind):
    if kind == 'field':
        return DeviceField(t)
    if kind == 'torch':
        return t
    return DeviceField(t).get()


def adi_explicit_rhs(Tn, grid, mat, params):
    """R0 of adi3d_numba_coeff.py:292-298 (stage entry point for per-stage parity tests / benchmarks)."""
    t, kind = _as_state(Tn, grid)
    kappa, _ = _gam(grid, mat, params)
    out = grid.layout.empty()
    check(lib.adi_explicit_rhs(_p(t), _p(grid.d_flags), *grid.layout.pd, grid.dx, params.dt,
                               kappa, params.theta, _p(out), _stream()))
    return _wrap(out, kind)


def _sparse_arg(grid, pack, dense):
    """the `sparse` argument of the sweep entry points: bit 0 = the pack arrays may be read only where the flags say a
    cell is exposed, bit 1 = all-solid box hint.  Bit 0 needs packs built for the mask the flags describe: after
    `grid.mask = new` WITHOUT a pack rebuild the reference pairs the stale packs with the live mask
    (adi3d_numba_coeff.py:150-162 reads coeff at every in-mask cell), so stale packs are read densely here too."""
    fresh = getattr(pack, 'mask_version', None) == grid.mask_version
    return int(pack.sparse_ok and fresh and not dense) | (2 if getattr(grid, 'all_solid', False) else 0)


class _NoFallback:
    """Bit 2 of `sparse` (include/adi_hip.h): skip the queue reset and the GENERAL launch behind a FAST kernel that is known
    to take every unit.  Which units a FAST kernel queues depends on the flags, the Dirichlet mask, the variant, the
    `sparse` bits and the shape -- not on the field -- so the first sweep of a configuration runs without the bit, the
    number of queued units is read back from the first word of the workspace (one 4-byte copy, once per mask / pack
    version), and later sweeps of the same configuration carry the bit when that number was zero.  The read-back is a
    host synchronisation, so it waits for the third sweep of a configuration: a layer-birth loop that changes the mask every
    two or three steps (waam.run_layer_birth) never pays it."""
    LEARN_AFTER = 3

    def __init__(self, grid, pack, entry, axis, v, sp, work):
        self.cache = pack.__dict__.setdefault('_nofb', {})
        self.key = (entry, axis, v, sp, grid.mask_version, getattr(pack, 'mask_version', None), grid.shape, grid.sx,
                    None if pack.d_dir_mask is None else pack.d_dir_mask.data_ptr())
        self.work = work
        self.state = self.cache.get(self.key, 0) if ((sp & 1) and work is not None and work.numel() >= 4) else False

    @property
    def bit(self):
        return 4 if self.state is True else 0

    def learn(self):
        st = self.state
        if st is True or st is False:
            return
        st += 1                                   # uses of this configuration so far
        if st >= self.LEARN_AFTER and not torch.cuda.is_current_stream_capturing():
            st = int(self.work[:4].view(torch.int32)[0].item()) == 0
        if self.key not in self.cache:            # a new configuration: entries of older mask versions are dead (a
            cur = self.key[4]                     # layer-birth run would otherwise add three keys per birth for good)
            for k in [k for k in self.cache if k[4] != cur]:
                del self.cache[k]
        self.cache[self.key] = st


def _sweep_into(axis, t_in, t_out, grid, mat, params, pack, Tinf, variant=None, xlo=None, xhi=None, dense=False):
    _, gam = _gam(grid, mat, params)
    _, work, wb = grid.scratch(2)
    v = pack.variant if variant is None else variant
    sp = _sparse_arg(grid, pack, dense)
    nf = _NoFallback(grid, pack, 'sweep', axis, v, sp, work)
    check(lib.adi_sweep(axis, v, _p(t_in), _p(grid.d_flags), _p(pack.d_coeff), _p(pack.d_dir_mask),
                        _p(pack.d_dir_val), _p(pack.d_qflux), *grid.layout.pd,
                        sp | nf.bit, params.theta,
                        gam, params.dt, float(Tinf), _p(t_out),
                        _p(xlo), _p(xhi), _fc_arg(grid, pack, sp),
                        _p(work), wb, _stream()))
    nf.learn()


def fused_supported(grid, cond_pass=False):
    """explicit stage folded into the axis-0 sweep (adi_explicit_sweep0, ABI v7) available for this grid"""
    return bool(lib.adi_explicit_fused_supported(*grid.layout.pd, 1 if cond_pass else 0))


def valid_range(t):
    """element offsets relative to t's first element that lie inside its storage: the [valid_lo, valid_hi) of
    adi_explicit_sweep0 / adi_explicit_condense0"""
    off = t.storage_offset()
    return -off, t.untyped_storage().nbytes() // t.element_size() - off


def _explicit_sweep0_into(t, t_out, grid, mat, params, pack, Tinf, variant=None, dense=False):
    """stages 1+2 of adi_step_numba_coeff (adi3d_numba_coeff.py:292-299) in one pass: R0 is evaluated inside the
    loads of the axis-0 sweep.  Neighbours are read anywhere inside t's storage."""
    kappa, _ = _gam(grid, mat, params)
    _, work, wb = grid.scratch(2)
    v = pack.variant if variant is None else variant
    vlo, vhi = valid_range(t)
    sp = _sparse_arg(grid, pack, dense)
    nf = _NoFallback(grid, pack, 'fused', 0, v, sp, work)
    check(lib.adi_explicit_sweep0(v, _p(t), vlo, vhi, _p(grid.d_flags), _p(pack.d_coeff), _p(pack.d_dir_mask),
                                  _p(pack.d_dir_val), _p(pack.d_qflux), *grid.layout.pd,
                                  sp | nf.bit, grid.dx, params.dt, kappa, params.theta,
                                  float(Tinf), _p(t_out), None, None, _fc_arg(grid, pack, sp), _p(work), wb, _stream()))
    nf.learn()


def adi_explicit_sweep_axis0(Tn, grid, mat, params, pack, Tinf=0.0, variant=None, dense=False):
    """U of adi3d_numba_coeff.py:299 straight from Tn (stage entry point of the fused kernel)."""
    t, kind = _as_state(Tn, grid)
    if variant == _lib.SWEEP_GENERAL or (variant is None and pack.variant == _lib.SWEEP_GENERAL):
        _ensure_general(pack)
    out = grid.layout.empty()
    _explicit_sweep0_into(t, out, grid, mat, params, pack, Tinf, variant, dense)
    return _wrap(out, kind)


def adi_sweep_axis(axis, stage_in, grid, mat, params, pack, Tinf=0.0, variant=None, dense=False):
    """sweep_axis0/1/2 of adi3d_numba_coeff.py:133-237 for one axis (stage entry point).
    variant=None picks the leanest kernel the pack allows; variant=_lib.SWEEP_GENERAL with dense=True forces
    the 42 B/cell general-pack kernel that reads every pack array in full, like the reference does."""
    t, kind = _as_state(stage_in, grid)
    if variant == _lib.SWEEP_GENERAL or (variant is None and pack.variant == _lib.SWEEP_GENERAL):
        _ensure_general(pack)
    out = grid.layout.empty()
    _sweep_into(axis, t, out, grid, mat, params, pack, Tinf, variant, dense=dense)
    return _wrap(out, kind)


def _ensure_general(pack):
    """materialise the arrays a forced general-pack sweep reads (zeros, like the reference's packs)"""
    L = pack.layout
    if pack.d_dir_mask is None:
        pack.d_dir_mask = L.empty(torch.uint8, zero=True)
    if pack.d_dir_val is None:
        pack.d_dir_val = L.empty(zero=True)
    if pack.d_qflux is None:
        pack.d_qflux = L.empty(zero=True)


def adi_step_hip_coeff(Tn, grid, mat, params, packs, Tinf=0.0):
    """adi3d_numba_coeff.py:290-302 / adi3d_gpu_coeff.py:213-230: explicit stage, then the three
    implicit sweeps in the order axis 0, 1, 2.  Returns a NEW array of the kind it was given;
    `Tn` is never modified."""
    t, kind = _as_state(Tn, grid)
    packx, packy, packz = packs
    (ta, tb), _, _ = grid.scratch(2)
    kappa, _ = _gam(grid, mat, params)
    out = grid.layout.empty()
    if fused_supported(grid):
        _explicit_sweep0_into(t, tb, grid, mat, params, packx, Tinf)
    else:
        check(lib.adi_explicit_rhs(_p(t), _p(grid.d_flags), *grid.layout.pd, grid.dx, params.dt,
                                   kappa, params.theta, _p(ta), _stream()))
        _sweep_into(0, ta, tb, grid, mat, params, packx, Tinf)
    _sweep_into(1, tb, ta, grid, mat, params, packy, Tinf)
    _sweep_into(2, ta, out, grid, mat, params, packz, Tinf)
    return _wrap(out, kind)


def apply_surface_impulse_Q(T, grid, mat, Q, face='z-'):
    """adi3d_numba_coeff.py:304-320: add dT = Q/(rho cp dx) to the in-mask cells of the domain-boundary plane of
    `face`, IN PLACE (NumPy array or DeviceField).  ValueError("bad face") for an unknown face."""
    if face not in FACES:
        raise ValueError("bad face")
    dT = Q / (mat.rho * mat.cp * grid.dx)
    ax, plus = FACES.index(face) // 2, FACES.index(face) % 2
    sl = [slice(None)] * 3
    sl[ax] = -1 if plus else 0
    sl = tuple(sl)
    if isinstance(T, DeviceField):
        sel = (grid.d_mask[sl] != 0)
        plane = T.t[sl]
        plane[sel] = plane[sel] + dT
    else:
        sel = np.asarray(grid.mask)[sl]
        T[sl][sel] += dT


# ---- per-voxel Robin correction: perimeter ratio --------------------------------------------
_LATERAL = ('x-', 'x+', 'y-', 'y+')


def exposed_faces_per_layer(mask_or_grid, faces=_LATERAL):
    """Number of exposed faces (in-mask cell whose neighbour across the face is outside the mask or the box) per plane
    k of axis 2, counted on the device from the neighbour-flags digest: int64 array of length nz.  With the four
    lateral faces this is the digital perimeter of every layer divided by dx."""
    for f in faces:
        if f not in FACES:
            raise ValueError("bad face")
    if isinstance(mask_or_grid, Grid3D):
        g = mask_or_grid
    else:
        m = np.asarray(mask_or_grid)
        g = Grid3D(m.shape[0], m.shape[1], m.shape[2], 1.0, m)
    bits = sum(1 << FACES.index(f) for f in set(faces))
    counts = torch.empty(g.layout.pz, dtype=torch.int64, device=_device())
    check(lib.adi_count_exposed_faces(_p(g.d_flags), *g.layout.pd, bits, _p(counts), _stream()))
    return counts[:g.nz].cpu().numpy()                # (the planes of the physical box beyond nz hold no in-mask cell)


def count_exposed_faces(mask2d):
    """quick_compare_layer_birth_robin_v3.py:97-108: exposed x-/x+/y-/y+ faces of a 2-D section (an integer)."""
    m = np.asarray(mask2d).astype(bool)
    assert m.ndim == 2
    return int(exposed_faces_per_layer(m[:, :, None])[0])


def perimeter_ratio(mask2d, dx, perim_true):
    """gamma = true perimeter / digital perimeter of the voxelised section (quick_compare_layer_birth_robin_v3.py:109-112);
    the lateral Robin coefficient of a staircase surface is scaled by it: h_side_eff = h_side * gamma (pi/4 for a
    large disk)."""
    faces = count_exposed_faces(mask2d)
    if faces == 0:
        raise ValueError("empty section")
    return float(perim_true) / (faces * float(dx))


# ---- layer birth on the device --------------------------------------------------------------
def birth_planes(T, d_active, d_full, grid, k_begin, k_end, Ts, count=None):
    """activate_layer (waam_from_stl_v7_mm.py:487-495) on the planes [k_begin, k_end) of axis 2, one kernel:
    newborn = full & ~active; T[newborn] = Ts; active |= full.  T: DeviceField; d_active / d_full: uint8 tensors in the
    grid's layout.  `count`: optional int64 device tensor (1 element) that receives the number of newborn cells."""
    L = grid.layout
    assert L.is_native(d_active) and L.is_native(d_full) and L.is_native(T.t)
    if count is None:
        count = torch.empty(1, dtype=torch.int64, device=T.t.device)
    check(lib.adi_birth_planes(_p(T.t), _p(d_active), _p(d_full), *grid.layout.pd, int(k_begin),
                               int(k_end), float(Ts), _p(count), _stream()))
    return count


class BirthPacks:
    """The coefficient packs of a part that grows by layer births along axis 2 (precompute_coeff_packs_unified with
    scalar / None face specifications, no Dirichlet cells), kept in HBM and UPDATED IN PLACE after a birth: only the
    planes of the new layer and the one below / above it change exposure, so the flags and the six coefficient arrays
    are rebuilt on [k_begin - 1, k_end + 1) instead of the whole box -- the same arrays the reference's full rebuild
    (waam_from_stl_v7_mm.py:534) would give."""

    def __init__(self, grid, mat, robin_h=None, neumann=None):
        self.grid, self.mat = grid, mat
        L = grid.layout

        def spec(v):
            if v is None:
                return (_lib.FACE_NONE, 0.0)
            if not np.isscalar(v):
                raise TypeError("BirthPacks takes scalar face specifications (use precompute_coeff_packs_unified for fields)")
            return (_lib.FACE_SCALAR, float(v))
        hs = [spec(None if robin_h is None else (robin_h.get(f, 0.0) if isinstance(robin_h, dict) else robin_h)) for f in FACES]
        qs = [spec(neumann.get(f) if neumann is not None else None) for f in FACES]
        self._args = ((ctypes.c_int * 6)(*[h[0] for h in hs]), (ctypes.c_double * 6)(*[h[1] for h in hs]), ptr_array([None] * 6),
                      (ctypes.c_int * 6)(*[q[0] for q in qs]), (ctypes.c_double * 6)(*[q[1] for q in qs]), ptr_array([None] * 6))
        self.coeff = [L.empty(zero=True) for _ in range(3)]
        self.qflux = [L.empty(zero=True) for _ in range(3)]
        has_q = any(q[0] != _lib.FACE_NONE for q in qs)
        self.packs = tuple(AxisCoeffPack(self.coeff[a], None, None, self.qflux[a], _has_dir=False, _has_q=has_q, _layout=L)
                           for a in range(3))
        fcs = _face_constants(grid, mat, [h[0] for h in hs], [h[1] for h in hs], [q[0] for q in qs], [q[1] for q in qs])
        for a, p in enumerate(self.packs):
            p.sparse_ok = True
            p.face_consts = fcs[a]
        self.update(0, grid.nz)

    def update(self, k_begin, k_end):
        """rebuild the planes [k_begin, k_end) of axis 2 from the grid's current device mask"""
        g, m = self.grid, self.mat
        k0, k1 = max(0, int(k_begin)), min(g.nz, int(k_end))
        hm, hs, hf, qm, qs, qf = self._args
        check(lib.adi_build_coeffs_planes(_p(g.d_mask), *g.layout.pd, g.dx, m.rho, m.cp, hm, hs, hf, qm, qs, qf,
                                          ptr_array([c.data_ptr() for c in self.coeff]),
                                          ptr_array([q.data_ptr() for q in self.qflux]), k0, k1, _stream()))
        for a, p in enumerate(self.packs):
            p.mask_version = g.mask_version
            p._fractions = (g, a, g.mask_version)
        return self.packs


# the reference's backend-specific names, so its drivers run unchanged on this module
adi_step_numba_coeff = adi_step_hip_coeff
adi_step_gpu_coeff = adi_step_hip_coeff


class StagedStepper:
    """The step of adi_step_hip_coeff with its arguments resolved once, for tight loops over a
    device-resident field (drivers call the step `nsub` times with the same packs and dt,
    quick_compare_dirichlet_robin.py:169-178).  `events`: optional list of 5 torch.cuda.Event recorded on
    the launch stream before/between/after the four stage kernels (per-stage HIP-event timing)."""

    def __init__(self, grid, mat, params, packs, Tinf=0.0, fused=None):
        self.grid, self.mat, self.params, self.packs, self.Tinf = grid, mat, params, packs, float(Tinf)
        self.fused = fused_supported(grid) if fused is None else (bool(fused) and fused_supported(grid))
        if self.fused:
            # the fused kernel reads T + flags (+ the pack arrays of the axis-0 sweep) and writes U: the sweep's own
            # byte count (SURVEY.md 8(d) variant rule); R0 never reaches HBM
            self.stage_names = ['explicit+sweep_axis0', 'sweep_axis1', 'sweep_axis2_contig']
            self.stage_bytes_per_cell = [p.bytes_per_cell for p in packs]
        else:
            self.stage_names = ['explicit', 'sweep_axis0', 'sweep_axis1', 'sweep_axis2_contig']
            self.stage_bytes_per_cell = [float(_lib.EXPLICIT_BYTES_PER_CELL)] + [p.bytes_per_cell for p in packs]

    def sweep_into(self, axis, t_in, t_out, variant=None, dense=False):
        if variant == _lib.SWEEP_GENERAL:
            _ensure_general(self.packs[axis])
        _sweep_into(axis, t_in, t_out, self.grid, self.mat, self.params, self.packs[axis], self.Tinf, variant,
                    dense=dense)

    def _step_into(self, t, out):
        """one step t -> out (both native-layout device tensors), no allocation: what a HIP graph captures"""
        g, prm = self.grid, self.params
        (ta, tb), _, _ = g.scratch(2)
        kappa, _ = _gam(g, self.mat, prm)
        if self.fused:
            _explicit_sweep0_into(t, tb, g, self.mat, prm, self.packs[0], self.Tinf)
        else:
            check(lib.adi_explicit_rhs(_p(t), _p(g.d_flags), *g.layout.pd, g.dx, prm.dt, kappa, prm.theta,
                                       _p(ta), _stream()))
            self.sweep_into(0, ta, tb)
        self.sweep_into(1, tb, ta)
        self.sweep_into(2, ta, out)

    def run(self, T, nsteps, graph=True):
        """The drivers' `nsub` loop (quick_compare_dirichlet_robin.py:169-178, waam_from_stl_v7_mm.py:525-528): `nsteps`
        steps with the same packs and dt on a device-resident field, returned as a new DeviceField.  The launches of
        two steps (X -> Y -> X) are captured once into a HIP graph and replayed, so small grids are not bound by the
        host's launch path (64^3: 3 kernels + 3 memsets per step, each a ctypes call); the graph is rebuilt when dt,
        theta, Tinf, the mask or the packs change.  graph=False: plain launches."""
        g, prm = self.grid, self.params
        nsteps = int(nsteps)
        key = (float(prm.dt), float(prm.theta), self.Tinf, g.mask_version, tuple(id(p) for p in self.packs),
               tuple(getattr(p, 'mask_version', None) for p in self.packs),
               tuple(None if p.d_coeff is None else p.d_coeff.data_ptr() for p in self.packs), self.fused)
        st = getattr(self, '_graph', None)
        if st is None or st['key'] != key:
            X, Y = g.layout.empty(), g.layout.empty()
            st = self._graph = dict(key=key, X=X, Y=Y, g=None)
        X, Y = st['X'], st['Y']
        X.copy_(g.layout.to_layout(T, torch.float64))
        if graph and nsteps >= 2 and st['g'] is None:
            g.scratch(2)                                   # every buffer exists before the capture
            self._step_into(X, Y); self._step_into(Y, X)   # warm-up outside the capture (lazy module loads, and the
            self._step_into(X, Y); self._step_into(Y, X)   # no-fallback promise is learnt on the third step); harmless:
            X.copy_(g.layout.to_layout(T, torch.float64))  # X is restored
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                self._step_into(X, Y)
                self._step_into(Y, X)
            st['g'] = cg
        done = 0
        if graph and st['g'] is not None:
            for _ in range(nsteps // 2):
                st['g'].replay()
            done = 2 * (nsteps // 2)
        cur, oth = X, Y
        for _ in range(nsteps - done):
            self._step_into(cur, oth)
            cur, oth = oth, cur
        out = g.layout.empty()
        out.copy_(cur)
        return DeviceField(out)

    def step(self, T, events=None):
        g, prm = self.grid, self.params
        t = g.layout.to_layout(T, torch.float64)
        (ta, tb), _, _ = g.scratch(2)
        kappa, _ = _gam(g, self.mat, prm)
        out = g.layout.empty()
        ne = 0

        def mark():
            nonlocal ne
            if events is not None:
                events[ne].record()
            ne += 1
        mark()
        if self.fused:
            _explicit_sweep0_into(t, tb, g, self.mat, prm, self.packs[0], self.Tinf)
        else:
            check(lib.adi_explicit_rhs(_p(t), _p(g.d_flags), *g.layout.pd, g.dx, prm.dt, kappa, prm.theta,
                                       _p(ta), _stream()))
            mark()
            self.sweep_into(0, ta, tb)
        mark()
        self.sweep_into(1, tb, ta)
        mark()
        self.sweep_into(2, ta, out)
        mark()
        return DeviceField(out)
