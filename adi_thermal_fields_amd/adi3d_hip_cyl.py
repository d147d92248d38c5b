"""adi3d_hip_cyl -- MI355X drop-in for the reference's cylindrical backend `adi3d_cyl_phi_v3`.

Operator surface of adi3d_cyl_phi_v3.py:33-68, :332-350:
    GridCyl, Material, Params, RobinR, ZBC, adi_step(Tn, grid, mat, prm, robin_r, zbc, S=None, theta=None)
plus adi_step_masked (quick_spiral_deposition_gif_v5.py:31-70).

Only the backward-Euler scheme is served: the reference's scheme="douglas" branch reads
uninitialised memory and omits the diffusivity, so there is nothing valid to match;
requesting it raises NotImplementedError instead of silently computing something else.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check
from .adi3d_hip_coeff import DeviceField, Layout, to_device, _device, _stream, _p, _wrap

__all__ = ['StagedCylStepper', 'GridCyl', 'Material', 'Params', 'RobinR', 'ZBC', 'adi_step', 'adi_step_masked', 'DeviceField', 'to_device']


class GridCyl:  # adi3d_cyl_phi_v3.py:33-43
    def __init__(self, nr, nphi, nz, dr, dphi, dz, R, R_in=0.0):
        """R_in: inner radius of an annular grid, r_i = R_in + (i + 1/2) dr.  The reference's own drivers and its only test
        pass it (quick_spiral_deposition_gif_v5.py:80, tests/test_spiral_vs_analytic.py:18) to a constructor that does not
        take it (TypeError at HEAD); here it is accepted and every formula is the reference's with r shifted."""
        self.nr = int(nr); self.nphi = int(nphi); self.nz = int(nz)
        self.dr = float(dr); self.dphi = float(dphi); self.dz = float(dz)
        self.R = float(R)
        self.R_in = float(R_in)
        self.r = self.R_in + (np.arange(self.nr, dtype=np.float64) + 0.5) * self.dr
        self.r_imh = self.r - 0.5 * self.dr
        self.r_iph = self.r + 0.5 * self.dr
        self.r_outer_face = self.r_iph[-1]
        self.layout = Layout(self.nr, self.nphi, self.nz, phys=(self.nr, self.nphi, self.nz))   # (no padded extents: the phi lines are periodic)
        self._plans = {}
        self._scratch = None

    @property
    def shape(self):
        return (self.nr, self.nphi, self.nz)

    def scratch(self):
        if self._scratch is None or self._scratch[0].device != _device():
            self._scratch = [self.layout.empty() for _ in range(2)]
        return self._scratch


class Material:  # :45-50
    def __init__(self, rho, cp, k):
        self.rho = float(rho); self.cp = float(cp); self.k = float(k)

    @property
    def alpha(self):
        return self.k / (self.rho * self.cp)


class Params:  # :52-54
    def __init__(self, dt, theta=0.5, scheme="be"):
        self.dt = float(dt); self.theta = float(theta); self.scheme = str(scheme).lower()


class RobinR:  # :56-58
    def __init__(self, h, T_inf):
        self.h = float(h); self.T_inf = float(T_inf)


class ZBC:  # :60-68
    def __init__(self, kind_bot='neumann0', kind_top='robin', h_bot=0.0, h_top=0.0,
                 T_inf_bot=20.0, T_inf_top=20.0, T_bot=20.0, T_top=20.0):
        self.kind_bot = kind_bot; self.kind_top = kind_top
        self.h_bot = float(h_bot); self.h_top = float(h_top)
        self.T_inf_bot = float(T_inf_bot); self.T_inf_top = float(T_inf_top)
        self.T_bot = float(T_bot); self.T_top = float(T_top)


class _Plan:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            lib.adi_cyl_plan_destroy(self.handle)
        except Exception:
            pass


def _plan(grid, mat, dt, robin_r, zbc):
    if zbc.kind_bot not in _lib.ZBC_KINDS:
        raise ValueError("unknown zbc.kind_bot")     # adi3d_cyl_phi_v3.py:283
    if zbc.kind_top not in _lib.ZBC_KINDS:
        raise ValueError("unknown zbc.kind_top")     # :296
    key = (torch.cuda.current_device(), grid.dr, grid.dphi, grid.dz, getattr(grid, 'R_in', 0.0), mat.rho, mat.cp, mat.k, dt, robin_r.h,
           robin_r.T_inf, zbc.kind_bot, zbc.kind_top, zbc.h_bot, zbc.h_top, zbc.T_inf_bot, zbc.T_inf_top,
           zbc.T_bot, zbc.T_top)
    pl = grid._plans.get(key)
    if pl is None:
        _device()
        h = ctypes.c_void_p()
        check(lib.adi_cyl_plan_create_annular(grid.nr, grid.nphi, grid.nz, grid.layout.sx, grid.dr, grid.dphi, grid.dz,
                                              getattr(grid, 'R_in', 0.0), mat.rho, mat.cp, mat.k, dt, robin_r.h, robin_r.T_inf,
                                              _lib.ZBC_KINDS[zbc.kind_bot], _lib.ZBC_KINDS[zbc.kind_top], zbc.h_bot,
                                              zbc.h_top, zbc.T_inf_bot, zbc.T_inf_top, zbc.T_bot, zbc.T_top, ctypes.byref(h)))
        pl = _Plan(h)
        if len(grid._plans) > 16:      # drivers vary dt between segments; keep the cache bounded
            grid._plans.clear()
        grid._plans[key] = pl
    return pl


def _state(Tn, grid):
    kind = 'field' if isinstance(Tn, DeviceField) else ('torch' if isinstance(Tn, torch.Tensor) else 'numpy')
    if kind == 'numpy':
        Tn = np.asarray(Tn)
    assert tuple(Tn.shape) == grid.shape
    return grid.layout.to_layout(Tn, torch.float64), kind


def _run(Tn, grid, mat, prm, robin_r, zbc, S, active, T_void, T_inner):
    # adi3d_cyl_phi_v3.py:335: `scheme = prm.scheme if prm.scheme in ('be', 'douglas') else 'be'` -- every string but
    # 'douglas' is backward Euler in the reference, and so it is here
    if prm.scheme == "douglas":
        raise NotImplementedError("adi3d_hip_cyl does not serve scheme='douglas': the reference's branch is numerically "
                                  "broken (reads uninitialised memory, omits alpha), so it has no valid oracle")
    t, kind = _state(Tn, grid)
    pl = _plan(grid, mat, prm.dt, robin_r, zbc)
    out = grid.layout.empty()
    d_S = None if S is None else grid.layout.to_layout(S, torch.float64)
    d_act = None if active is None else grid.layout.to_layout(active, torch.uint8)
    check(lib.adi_cyl_step(pl.handle, _p(t), _p(out), None, None, _p(d_S), _p(d_act),
                           float(T_void), float(T_inner), _stream()))
    return _wrap(out, kind)


class StagedCylStepper:
    """adi_step (BE) with its arguments resolved once, for loops over a device-resident field and per-sweep timing:
    `events`: optional list of 4 torch.cuda.Event recorded on the launch stream before / between / after the r, phi
    and z sweeps (adi_cyl_sweep of the C ABI)."""
    stage_names = ['sweep_r', 'sweep_phi', 'sweep_z_contig']
    stage_bytes_per_cell = [16.0, 16.0, 16.0]          # field in + field out per sweep

    def __init__(self, grid, mat, prm, robin_r, zbc):
        if prm.scheme == "douglas":
            raise NotImplementedError("scheme='douglas' has no valid oracle (see adi_step)")
        self.grid = grid
        self.plan = _plan(grid, mat, prm.dt, robin_r, zbc)

    def step(self, T, events=None):
        """r sweep T -> out, then the phi and z sweeps IN PLACE on out (every sweep kernel reads only the rows it writes):
        the input is untouched, as in the reference, and two of the three sweeps work on one field instead of two"""
        g = self.grid
        t = g.layout.to_layout(T, torch.float64)
        out = g.layout.empty()
        seq = ((0, t, out), (1, out, out), (2, out, out)) if g.nphi > 1 else ((0, t, out), (2, out, out))
        if events is not None:
            events[0].record()
        for ax, a, b in seq:
            check(lib.adi_cyl_sweep(self.plan.handle, ax, _p(a), _p(b), None, None, 0.0, 0.0, _stream()))
            if events is not None:
                events[ax + 1].record()
                if ax == 0 and g.nphi == 1:
                    events[2].record()         # no phi sweep (phi_solve_spectral copies, :319-320): an empty interval
        return DeviceField(out)

    def _step_inplace(self, x, events=None):
        """one step on x in place (native-layout device tensor), no allocation: what a HIP graph captures.
        events: as in step()"""
        if events is not None:
            events[0].record()
        for ax in ((0, 1, 2) if self.grid.nphi > 1 else (0, 2)):
            check(lib.adi_cyl_sweep(self.plan.handle, ax, _p(x), _p(x), None, None, 0.0, 0.0, _stream()))
            if events is not None:
                events[ax + 1].record()
                if ax == 0 and self.grid.nphi == 1:
                    events[2].record()

    def run(self, T, nsteps, graph=True):
        """`nsteps` BE steps with the same plan on a device-resident field (the drivers' inner loops,
        quick_compare_layer_birth_robin_cyl_v3.py), returned as a new DeviceField.  The loop owns its field, so all three
        sweeps run IN PLACE: the working set is one field (134 MB at 128 x 256 x 512, inside the 256 MB Infinity Cache)
        instead of two -- 0.160 -> 0.144 ms per step.  The step is three kernels of ~45 us, at the edge of launch-bound: the
        launches of one step are captured once into a HIP graph and replayed.  Bit-identical to calling step() nsteps
        times.  graph=False: plain launches."""
        g = self.grid
        nsteps = int(nsteps)
        st = getattr(self, '_graph', None)
        if st is None:
            st = self._graph = dict(X=g.layout.empty(), g=None)
        X = st['X']
        X.copy_(g.layout.to_layout(T, torch.float64))
        if graph and nsteps >= 2 and st['g'] is None:
            self._step_inplace(X)                          # warm-up outside the capture (lazy module loads)
            X.copy_(g.layout.to_layout(T, torch.float64))
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                self._step_inplace(X)
            st['g'] = cg
        for _ in range(nsteps):
            if graph and st['g'] is not None:
                st['g'].replay()
            else:
                self._step_inplace(X)
        out = g.layout.empty()
        out.copy_(X)
        return DeviceField(out)


def adi_step(Tn, grid, mat, prm, robin_r, zbc, S=None, theta=None):
    """adi3d_cyl_phi_v3.py:332-350 (BE branch: r -> phi -> z with theta = 1; `theta` is unused there too)."""
    return _run(Tn, grid, mat, prm, robin_r, zbc, S, None, 0.0, 0.0)


def adi_step_masked(Tn, grid, mat, prm, robin_outer, zbc, active, robin_inner=None, robin_void=None):
    """quick_spiral_deposition_gif_v5.py:31-70: void cells clamped to robin_void.T_inf before and after
    the step, inactive axis-row cells to robin_inner.T_inf; the clamps are fused into the r-sweep load
    and the z-sweep store."""
    robin_inner = robin_inner or robin_outer
    robin_void = robin_void or robin_outer
    return _run(Tn, grid, mat, prm, robin_outer, zbc, None, active, robin_void.T_inf, robin_inner.T_inf)
