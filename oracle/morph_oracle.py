"""ORACLE (test infrastructure, never shipped): NumPy restatement of the reference's 6-connectivity voxel morphology
and of its two ASCII VTK writers, the callers on the mask side of the hot path.

  dilate6               waam_from_stl_v7_mm.py:73-82
  erode6                waam_from_stl_v7_mm.py:84-96
  closing6              waam_from_stl_v7_mm.py:98-104
  flood_fill_outside    waam_from_stl_v7_mm.py:106-134
  solidify_mask         waam_from_stl_v7_mm.py:136-188
  vtk_ascii_points      vtk_writer.py:4-30           (cell-centre origin, Fortran-order flatten, 9 values per line)
  vtk_ascii_waam        waam_from_stl_v7_mm.py:191-216 (mm, one x-row per line, %.6g)

Pinned bit-for-bit to outputs of the imported reference (tests/golden/make_golden_io.py -> tests/golden/io_*.npz).

Reference defect D8 (reproduced only on request): flood_fill_outside pads the SOLID with True, so the padding layer is
not air, no seed exists, the function returns all-False and solidify_mask(mode='flood'/'close_flood') fills the whole
box.  `reference_defect=True` restates exactly that; the default is the behaviour its comment describes (everything
outside the box is air), which the product implements."""
import io

import numpy as np


def dilate6(a):
    a = np.asarray(a).astype(bool, copy=False)
    b = a.copy()
    b[1:, :, :] |= a[:-1, :, :]
    b[:-1, :, :] |= a[1:, :, :]
    b[:, 1:, :] |= a[:, :-1, :]
    b[:, :-1, :] |= a[:, 1:, :]
    b[:, :, 1:] |= a[:, :, :-1]
    b[:, :, :-1] |= a[:, :, 1:]
    return b


def erode6(a):
    a = np.asarray(a).astype(bool, copy=False)
    b = np.zeros_like(a, dtype=bool)
    b[1:-1, 1:-1, 1:-1] = (a[1:-1, 1:-1, 1:-1] & a[:-2, 1:-1, 1:-1] & a[2:, 1:-1, 1:-1] & a[1:-1, :-2, 1:-1]
                           & a[1:-1, 2:, 1:-1] & a[1:-1, 1:-1, :-2] & a[1:-1, 1:-1, 2:])
    return b


def closing6(a, iters=1):
    x = np.asarray(a).astype(bool, copy=False)
    for _ in range(max(0, iters)):
        x = dilate6(x)
    for _ in range(max(0, iters)):
        x = erode6(x)
    return x


def flood_fill_outside(solid, max_iters=None, reference_defect=False):
    s = np.asarray(solid).astype(bool, copy=False)
    air = ~np.pad(s, 1, mode='constant', constant_values=bool(reference_defect))
    outside = np.zeros_like(air, dtype=bool)
    outside[0, :, :] |= air[0, :, :]; outside[-1, :, :] |= air[-1, :, :]
    outside[:, 0, :] |= air[:, 0, :]; outside[:, -1, :] |= air[:, -1, :]
    outside[:, :, 0] |= air[:, :, 0]; outside[:, :, -1] |= air[:, :, -1]
    if max_iters is None:
        max_iters = sum(s.shape) + 10
    for _ in range(max_iters):
        new = outside | (dilate6(outside) & air)
        if new.sum() == outside.sum():
            break
        outside = new
    return outside[1:-1, 1:-1, 1:-1]


def solidify_mask(mask_surface, mode='auto', close_iters=2, reference_defect=False):
    m = np.asarray(mask_surface).astype(bool, copy=False)

    def is_shell_like(a):
        if a.sum() == 0:
            return True
        ratio = erode6(a).sum() / float(a.sum())
        return (ratio < 0.25) or (a.mean() < 0.02)

    if mode in ('off', 'fill'):
        return m
    if mode == 'flood':
        outside = flood_fill_outside(m, reference_defect=reference_defect)
        return m | ((~m) & (~outside))
    if mode == 'close_flood':
        closed = closing6(m, iters=int(close_iters))
        outside = flood_fill_outside(closed, reference_defect=reference_defect)
        return closed | ((~closed) & (~outside))
    if mode == 'auto':
        return solidify_mask(m, 'close_flood', close_iters, reference_defect) if is_shell_like(m) else m
    return m


def vtk_ascii_points(T, dx, origin=(0.0, 0.0, 0.0), field_name="Temperature", mask=None):
    """the text vtk_writer.write_vtk_structured_points writes (returned as str)"""
    T = np.asarray(T)
    nx, ny, nz = T.shape
    ox, oy, oz = origin
    oc = (ox + dx * 0.5, oy + dx * 0.5, oz + dx * 0.5)
    f = io.StringIO()

    def scalars(name, flat):
        f.write(f"SCALARS {name} float 1\n")
        f.write("LOOKUP_TABLE default\n")
        for i in range(0, flat.size, 9):
            f.write(" ".join(f"{float(v):.6e}" for v in flat[i:i + 9]) + "\n")
    f.write("# vtk DataFile Version 3.0\n")
    f.write("Uniform grid with Temperature and mask\n")
    f.write("ASCII\n")
    f.write("DATASET STRUCTURED_POINTS\n")
    f.write(f"DIMENSIONS {nx} {ny} {nz}\n")
    f.write(f"ORIGIN {oc[0]:.9e} {oc[1]:.9e} {oc[2]:.9e}\n")
    f.write(f"SPACING {dx:.9e} {dx:.9e} {dx:.9e}\n")
    f.write(f"POINT_DATA {nx*ny*nz}\n")
    scalars(field_name, T.reshape(-1, order='F'))
    if mask is not None:
        scalars("mask", np.asarray(mask, dtype=np.float32).reshape(-1, order='F'))
    return f.getvalue()


def vtk_ascii_waam(T, dx_mm, origin_mm=(0.0, 0.0, 0.0), field_name="Temperature", mask=None):
    """the text waam_from_stl_v7_mm.write_vtk_structured_points writes (returned as str)"""
    T = np.asarray(T)
    nx, ny, nz = T.shape
    ox, oy, oz = map(float, origin_mm)
    dx = float(dx_mm)
    f = io.StringIO()
    f.write("# vtk DataFile Version 3.0\n")
    f.write("WAAM Structured Points (mm)\n")
    f.write("ASCII\n")
    f.write("DATASET STRUCTURED_POINTS\n")
    f.write(f"DIMENSIONS {nx} {ny} {nz}\n")
    f.write(f"ORIGIN {ox:.9g} {oy:.9g} {oz:.9g}\n")
    f.write(f"SPACING {dx:.9g} {dx:.9g} {dx:.9g}\n")
    f.write(f"POINT_DATA {nx*ny*nz}\n")

    def block(name, A):
        f.write(f"SCALARS {name} float 1\n")
        f.write("LOOKUP_TABLE default\n")
        for k in range(nz):
            for j in range(ny):
                f.write(" ".join(f"{float(A[i, j, k]):.6g}" for i in range(nx)) + "\n")
    block(field_name, T)
    if mask is not None:
        block("Mask", np.asarray(mask, dtype=np.float32))
    return f.getvalue()
