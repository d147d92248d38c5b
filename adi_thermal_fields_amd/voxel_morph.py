"""voxel_morph -- the reference's 6-connectivity voxel morphology on the MI355X.

Same names and argument meaning as waam_from_stl_v7_mm.py:73-188 (`dilate6`, `erode6`, `closing6`,
`flood_fill_outside`, `solidify_mask`), so the voxeliser's post-processing drops in unchanged; masks go in and come
out as NumPy bool arrays (or stay on the device when given / asked for torch tensors).  HIP kernels through the C ABI
(`adi_morph6`, `adi_flood_outside`); no CPU fallback.

Reference defect D8: `flood_fill_outside` pads the *solid* with True, finds no seed and returns all-False,
so `solidify_mask(mode='flood'/'close_flood')` fills the whole box.  Here the flood fill does what its comment says
(everything outside the box is air); `reference_defect=True` reproduces the reference's output instead.
"""
import ctypes

import numpy as np
import torch

from ._lib import check, lib

__all__ = ['dilate6', 'erode6', 'closing6', 'flood_fill_outside', 'solidify_mask']


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("voxel_morph needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device())


def _to_dev(a):
    """-> (dense uint8 device tensor (nx, ny, nz), was_tensor)"""
    if isinstance(a, torch.Tensor):
        t = a.to(device=_dev())
        t = (t != 0).to(torch.uint8).contiguous()
        assert t.ndim == 3
        return t, True
    arr = np.ascontiguousarray(np.asarray(a).astype(np.bool_, copy=False)).view(np.uint8)
    assert arr.ndim == 3
    return torch.from_numpy(arr).to(_dev()), False


def _back(t, was_tensor):
    return t if was_tensor else t.cpu().numpy().astype(np.bool_)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _morph(op, t):
    out = torch.empty_like(t)
    nx, ny, nz = t.shape
    check(lib.adi_morph6(op, _p(t), _p(out), nx, ny, nz, _stream()))
    return out


def dilate6(a):
    """waam_from_stl_v7_mm.py:73-82"""
    t, wt = _to_dev(a)
    return _back(_morph(0, t), wt)


def erode6(a):
    """waam_from_stl_v7_mm.py:84-96 (the box boundary is always eroded away)"""
    t, wt = _to_dev(a)
    return _back(_morph(1, t), wt)


def _closing(t, iters):
    for _ in range(max(0, iters)):
        t = _morph(0, t)
    for _ in range(max(0, iters)):
        t = _morph(1, t)
    return t


def closing6(a, iters=1):
    """waam_from_stl_v7_mm.py:98-104"""
    t, wt = _to_dev(a)
    return _back(_closing(t, int(iters)), wt)


def _flood(t, reference_defect):
    out = torch.empty_like(t)
    if reference_defect:
        return out.zero_()                         # D8: no seed is ever found
    nx, ny, nz = t.shape
    flag = torch.zeros(1, dtype=torch.int32, device=t.device)
    rounds = ctypes.c_int(0)
    check(lib.adi_flood_outside(_p(t), _p(out), nx, ny, nz, _p(flag), ctypes.byref(rounds), _stream()))
    return out


def flood_fill_outside(solid, max_iters=None, reference_defect=False):
    """waam_from_stl_v7_mm.py:106-134: True where the air is connected to the outside of the box.  `max_iters` is
    accepted for signature compatibility; the device version scans whole lines per pass and runs to the fixed point."""
    t, wt = _to_dev(solid)
    return _back(_flood(t, reference_defect), wt)


def solidify_mask(mask_surface, mode='auto', close_iters=2, verbose=False, reference_defect=False):
    """waam_from_stl_v7_mm.py:136-188: 'off' / 'fill' (unchanged), 'flood', 'close_flood', 'auto' (shell test:
    erosion ratio < 0.25 or fill fraction < 0.02 -> close_flood)."""
    t, wt = _to_dev(mask_surface)

    def fill(m):
        outside = _flood(m, reference_defect)
        return m | ((1 - m) & (1 - outside))
    if mode == 'flood':
        t = fill(t)
    elif mode == 'close_flood':
        t = fill(_closing(t, int(close_iters)))
    elif mode == 'auto':
        n = int(t.sum())
        shell = True
        if n > 0:
            ratio = int(_morph(1, t).sum()) / float(n)
            shell = (ratio < 0.25) or (n / float(t.numel()) < 0.02)
        if verbose:
            print("[solidify] auto: %s" % ('SHELL -> close_flood' if shell else 'SOLID -> unchanged'))
        if shell:
            t = fill(_closing(t, int(close_iters)))
    return _back(t, wt)
