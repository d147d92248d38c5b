"""frame_io -- frame output for the WAAM drivers.

The reference writes legacy-VTK STRUCTURED_POINTS files in ASCII, one formatted value at a time in Python
(`vtk_writer.write_vtk_structured_points`, vtk_writer.py:12-30; `waam_from_stl_v7_mm.write_vtk_structured_points`,
:191-216) -- minutes per frame at 256 x 256 x 320 once a step takes milliseconds.  Here:

  write_vtk_structured_points(path, T, dx, origin, field_name, mask, binary=True)
      same signature and header as vtk_writer.py; BINARY payload (big-endian float32, x fastest) by default, packed on the
      device from the HBM-resident field (`adi_pack_frame_f32be`); `binary=False` reproduces the reference's ASCII text
      byte for byte.
  write_vtk_structured_points_mm(...)   the waam driver's variant (mm units, origin as given, "Mask" block)
  write_npy(path, T)                    raw fp64 .npy of the field (C order), for restart / post-processing
"""
import ctypes

import numpy as np
import torch

from . import adi3d_hip_coeff as _hip
from ._lib import check, lib

__all__ = ['write_vtk_structured_points', 'write_vtk_structured_points_mm', 'write_npy', 'pack_frame_f32be']


def pack_frame_f32be(T):
    """DeviceField / device tensor / NumPy (nx, ny, nz) fp64 -> bytes of the big-endian float32 values in VTK point
    order (x fastest), packed on the device."""
    t = T.t if isinstance(T, _hip.DeviceField) else T
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(t), dtype=np.float64))
    t = t.to(device=_hip._device(), dtype=torch.float64)
    assert t.ndim == 3
    nx, ny, nz = t.shape
    if not (t.stride(2) == 1 and t.stride(1) == nz and t.stride(0) >= ny * nz):
        t = t.contiguous()
    out = torch.empty(nx * ny * nz, dtype=torch.int32, device=t.device)
    check(lib.adi_pack_frame_f32be(ctypes.c_void_p(t.data_ptr()), nx, ny, nz, t.stride(0), ctypes.c_void_p(out.data_ptr()),
                                   _hip._stream()))
    return out.cpu().numpy().tobytes()


def _ascii_scalars9(f, name, flat):
    f.write(f"SCALARS {name} float 1\n")
    f.write("LOOKUP_TABLE default\n")
    for i in range(0, flat.size, 9):
        f.write(" ".join(f"{float(v):.6e}" for v in flat[i:i + 9]) + "\n")


def _host(T):
    return np.asarray(T.get() if isinstance(T, _hip.DeviceField) else (T.cpu().numpy() if isinstance(T, torch.Tensor) else T))


def _mask_field(mask):
    m = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    return np.ascontiguousarray(m.astype(np.float32).astype(np.float64))


def write_vtk_structured_points(path, T, dx, origin=(0.0, 0.0, 0.0), field_name="Temperature", mask=None, binary=True):
    """vtk_writer.py:12-30 -- same header (origin shifted to the cell centre), point order and block names."""
    nx, ny, nz = T.shape
    ox, oy, oz = origin
    oc = (ox + dx * 0.5, oy + dx * 0.5, oz + dx * 0.5)
    head = ("# vtk DataFile Version 3.0\n" "Uniform grid with Temperature and mask\n" "%s\n"
            "DATASET STRUCTURED_POINTS\n" f"DIMENSIONS {nx} {ny} {nz}\n"
            f"ORIGIN {oc[0]:.9e} {oc[1]:.9e} {oc[2]:.9e}\n" f"SPACING {dx:.9e} {dx:.9e} {dx:.9e}\n"
            f"POINT_DATA {nx*ny*nz}\n")
    if not binary:
        with open(path, 'w', encoding='utf-8') as f:
            f.write(head % "ASCII")
            _ascii_scalars9(f, field_name, _host(T).reshape(-1, order='F'))
            if mask is not None:
                _ascii_scalars9(f, "mask", np.asarray(_host(mask), dtype=np.float32).reshape(-1, order='F'))
        return
    with open(path, 'wb') as f:
        f.write((head % "BINARY").encode('ascii'))
        f.write(f"SCALARS {field_name} float 1\nLOOKUP_TABLE default\n".encode('ascii'))
        f.write(pack_frame_f32be(T)); f.write(b"\n")
        if mask is not None:
            f.write(b"SCALARS mask float 1\nLOOKUP_TABLE default\n")
            f.write(pack_frame_f32be(_mask_field(mask))); f.write(b"\n")


def write_vtk_structured_points_mm(path, T, dx_mm, origin_mm=(0.0, 0.0, 0.0), field_name="Temperature", mask=None,
                                   binary=True):
    """waam_from_stl_v7_mm.py:191-216 -- mm units, origin as given, second block named "Mask"."""
    nx, ny, nz = T.shape
    ox, oy, oz = map(float, origin_mm)
    dx = float(dx_mm)
    head = ("# vtk DataFile Version 3.0\n" "WAAM Structured Points (mm)\n" "%s\n" "DATASET STRUCTURED_POINTS\n"
            f"DIMENSIONS {nx} {ny} {nz}\n" f"ORIGIN {ox:.9g} {oy:.9g} {oz:.9g}\n" f"SPACING {dx:.9g} {dx:.9g} {dx:.9g}\n"
            f"POINT_DATA {nx*ny*nz}\n")
    if not binary:
        Th = _host(T)
        with open(path, 'w', encoding='utf-8') as f:
            f.write(head % "ASCII")

            def block(name, A):
                f.write(f"SCALARS {name} float 1\n")
                f.write("LOOKUP_TABLE default\n")
                for k in range(nz):
                    for j in range(ny):
                        f.write(" ".join(f"{float(A[i, j, k]):.6g}" for i in range(nx)) + "\n")
            block(field_name, Th)
            if mask is not None:
                block("Mask", np.asarray(_host(mask), dtype=np.float32))
        return
    with open(path, 'wb') as f:
        f.write((head % "BINARY").encode('ascii'))
        f.write(f"SCALARS {field_name} float 1\nLOOKUP_TABLE default\n".encode('ascii'))
        f.write(pack_frame_f32be(T)); f.write(b"\n")
        if mask is not None:
            f.write(b"SCALARS Mask float 1\nLOOKUP_TABLE default\n")
            f.write(pack_frame_f32be(_mask_field(mask))); f.write(b"\n")


def write_npy(path, T):
    """the field as a C-order fp64 .npy (downloaded once, no text formatting)"""
    np.save(path, np.ascontiguousarray(_host(T), dtype=np.float64))
