"""Reference engine for the slab decomposition (tests only): NumPy/oracle restatements of the five engine
calls of adi_thermal_fields_amd.dist_slab, on CPU tensors, so the exchange logic and the algebra of the
reduced interface system can be run on gloo ranks without a GPU.  The interface solve here assembles the
dense (2R x 2R) system per line instead of merging, as an independent check of the product's method."""
import numpy as np
import torch

from oracle import adi_oracle as orc


class CpuLayout:
    def __init__(self, nx, ny, nz):
        self.nx, self.ny, self.nz = nx, ny, nz
        self.sx = ny * nz

    shape = property(lambda self: (self.nx, self.ny, self.nz))

    def empty(self, dtype=torch.float64, zero=False):
        return torch.zeros(self.shape, dtype=dtype)

    def to_layout(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(dtype).contiguous()
        arr = np.asarray(a)
        if dtype == torch.uint8:
            arr = arr.astype(np.bool_).astype(np.uint8)
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)


class _Pack:
    def __init__(self, coeff, dm, dv, q, variant):
        self.d_coeff, self.d_dir_mask, self.d_dir_val, self.d_qflux, self.variant = coeff, dm, dv, q, variant


def _flags(mask):
    m = mask.astype(bool)
    f = m.astype(np.uint8)
    for ax in range(3):
        lo = np.zeros_like(m); hi = np.zeros_like(m)
        sl_hi = [slice(None)] * 3; sl_lo = [slice(None)] * 3
        sl_hi[ax] = slice(1, None); sl_lo[ax] = slice(None, -1)
        lo[tuple(sl_hi)] = m[tuple(sl_lo)]
        hi[tuple(sl_lo)] = m[tuple(sl_hi)]
        f |= ((m & lo).astype(np.uint8) << (1 + 2 * ax)) | ((m & hi).astype(np.uint8) << (2 + 2 * ax))
    return f


def _line_systems(axis, t_in, flags, pack, theta, gam, dt, Tinf):
    """full-length rows (a, b, c, d) of every line along `axis`, arrays shaped like the field
    (adi3d_gpu_coeff.py:173-187 with the neighbour tests taken from the flags, halo bits included)"""
    R = t_in.numpy(); f = flags.numpy()
    co = pack[0].numpy()
    dm = pack[1].numpy().astype(bool) if pack[1] is not None else np.zeros(R.shape, bool)
    dv = pack[2].numpy() if pack[2] is not None else np.zeros(R.shape)
    q = pack[3].numpy() if pack[3] is not None else np.zeros(R.shape)
    m = (f & 1).astype(bool)
    L = ((f >> (1 + 2 * axis)) & 1).astype(bool)
    H = ((f >> (2 + 2 * axis)) & 1).astype(bool)
    fr = m & ~dm
    tg = theta * gam
    a = np.where(fr & L, -tg, 0.0); c = np.where(fr & H, -tg, 0.0)
    b = np.where(fr, 1.0 + tg * (L.astype(float) + H.astype(float)) + dt * co, 1.0)
    d = np.where(fr, R + dt * q + dt * co * Tinf, np.where(m & dm, dv, R))
    return a, b, c, d


def _dense(a, b, c):
    n = len(b)
    A = np.diag(b)
    for i in range(1, n):
        A[i, i - 1] = a[i]; A[i - 1, i] = c[i - 1]
    return A


class CpuEngine:
    device = torch.device('cpu')

    def layout(self, nx, ny, nz, sx=None):
        return CpuLayout(nx, ny, nz)

    def vec(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def build_flags(self, L, mask_ext):
        self._flags_ext = torch.from_numpy(_flags(mask_ext.numpy()))
        return self._flags_ext

    def build_packs(self, L, mask_ext, flags_ext, dx, mat, dir_mask, dir_value, neumann, robin_h):
        grid = orc.Grid3D(L.nx, L.ny, L.nz, dx, mask_ext.numpy().astype(bool))
        packs = orc.precompute_coeff_packs_unified(grid, orc.Material(mat.rho, mat.cp, mat.k), dir_mask=dir_mask,
                                                   dir_value=dir_value, neumann=neumann, robin_h=robin_h)
        has_dir = dir_mask is not None and bool(np.any(dir_mask))
        has_q = neumann is not None and any(v is not None for v in neumann.values())
        variant = (0 if has_q else 2) if has_dir else (1 if has_q else 3)
        t = torch.from_numpy
        return tuple(_Pack(t(p.coeff), t(p.dir_mask.view(np.uint8)), t(p.dir_val), t(p.qflux), variant) for p in packs)

    def explicit(self, L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin=0, i_end=None):
        mask = (flags_ext.numpy() & 1).astype(bool)
        grid = orc.Grid3D(L.nx, L.ny, L.nz, dx, mask)
        out_ext.copy_(torch.from_numpy(orc.explicit_rhs(np.nan_to_num(T_ext.numpy()), grid, orc.Material(1.0, 1.0, kappa),
                                                        orc.Params(dt, theta))))

    # fused explicit + axis-0 passes of the product engine: here simply the explicit stage on the whole extended
    # slab followed by the plain pass on the box
    def fused_supported(self, nx, ny, nz, sx, cond_pass):
        return True

    def _r0_box(self, L, T_ext, i0, j0, flags, dx, dt, kappa, theta):
        ne = T_ext.shape[0]
        # flags of the box carry the halo coupling bits; rebuild a mask for the whole extended slab from T_ext's shape
        fl_ext = self._flags_ext
        Le = CpuLayout(ne, T_ext.shape[1], T_ext.shape[2])
        r0 = torch.zeros(Le.shape, dtype=torch.float64)
        self.explicit(Le, T_ext, fl_ext, dx, dt, kappa, theta, r0)
        return r0[i0:i0 + L.nx, j0:j0 + L.ny, :].contiguous()

    def sweep0_fused(self, variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf, t_out, xlo=None, xhi=None):
        r0 = self._r0_box(L, T_ext, i0, j0, flags, dx, dt, kappa, theta)
        tmp = torch.zeros(L.shape, dtype=torch.float64)
        self.sweep(0, variant, L, r0, flags, pack, theta, kappa * dt / (dx * dx), dt, Tinf, tmp, xlo, xhi)
        t_out.copy_(tmp)

    def condense0_fused(self, variant, L, T_ext, i0, j0, flags, pack, dx, dt, kappa, theta, Tinf, cond, r0_out=None):
        r0 = self._r0_box(L, T_ext, i0, j0, flags, dx, dt, kappa, theta)
        if r0_out is not None:
            r0_out.copy_(r0)
        self.condense(0, variant, L, r0, flags, pack, theta, kappa * dt / (dx * dx), dt, Tinf, cond)

    # pass A folded into the explicit stage (product: dot products in the marching kernel): here the explicit stage
    # followed by the dense condensation of the requested lines
    def dots_supported(self, nxl, ny, nz, sx):
        return True

    def dots_setup(self, Li, flags_int, dmask_int, theta, gam):
        return {}

    def explicit_dots(self, L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin, i_end, dd, i_org, n_line):
        self.explicit(L, T_ext, flags_ext, dx, dt, kappa, theta, out_ext, i_begin, i_end)

    def dots_finish(self, variant, Li, dd, r0, flags, pack, theta, gam, dt, Tinf, line_begin, line_end, cond):
        j0, j1 = line_begin // Li.nz, line_end // Li.nz
        assert j0 * Li.nz == line_begin and j1 * Li.nz == line_end
        cut = lambda t: None if t is None else t[:, j0:j1, :]
        self.condense(0, variant, CpuLayout(Li.nx, j1 - j0, Li.nz), cut(r0), cut(flags), tuple(cut(t) for t in pack), theta,
                      gam, dt, Tinf, cond)

    # deferred form of the sharded-axis sweep: dense restatements, independent of the product's kernels
    def lines_all_uniform(self, Li, flags_int, dmask_int):
        f = flags_int.numpy()
        if dmask_int is not None and bool(dmask_int.numpy().any()):
            return False
        n = f.shape[0]
        if n < 2:
            return False
        ok = ((f[0] & 5) == 5).all() and ((f[-1] & 3) == 3).all() and ((f[1:-1] & 7) == 7).all()
        both_ends = (((f[0] & 2) == 0) & ((f[-1] & 4) == 0)).any()     # a line with two modified end rows is not uniform
        return bool(ok and not both_ends)

    def deferred_setup(self, n, theta, gam, tol):
        tg = theta * gam
        A = _dense(np.full(n, -tg), np.full(n, 1.0 + 2.0 * tg), np.full(n, -tg))
        w = tg * np.linalg.solve(A, np.eye(n)[:, 0])
        reach = int((w > tol).sum())
        w[w <= tol] = 0.0
        return dict(w=torch.from_numpy(w), omega=float(w[0]), reach=reach)

    def interface_deferred(self, first, last, prev_last, next_first, omega, nlines, ulo, uhi):
        lo = ulo.numpy(); hi = uhi.numpy()
        lo[:] = 0.0; hi[:] = 0.0
        M = np.array([[1.0, -omega], [-omega, 1.0]])         # unknowns (last of the slab below, first of the slab above)
        if prev_last is not None:
            g = np.stack([prev_last.numpy().reshape(-1), first.numpy().reshape(-1)])
            lo[:] = np.linalg.solve(M, g)[0]
        if next_first is not None:
            g = np.stack([last.numpy().reshape(-1), next_first.numpy().reshape(-1)])
            hi[:] = np.linalg.solve(M, g)[1]

    # the deferred form without decay: homogeneous solutions of every line by a dense solve with the line's actual end rows,
    # and the two coefficients by least squares on (w, reversed w) -- independent of the product's Sherman-Morrison formulas
    def interface_uniform(self, g_all, mat_all, scal_all, world, rank, nlines, xlo, xhi):
        G = g_all.view(world, 2, nlines).numpy(); M = mat_all.view(world, 4, nlines).numpy()
        S = scal_all.view(world, 2).numpy()
        C = np.zeros((world, 6, nlines))
        for r in range(world):
            C[r, 0] = G[r, 0]; C[r, 3] = G[r, 1]
            if r in (0, world - 1):
                C[r, 1], C[r, 2], C[r, 4], C[r, 5] = M[r]
            else:
                C[r, 1] = -S[r, 0]; C[r, 2] = -S[r, 1]; C[r, 4] = -S[r, 1]; C[r, 5] = -S[r, 0]
        self.interface(torch.from_numpy(C.reshape(-1)), world, rank, nlines, xlo, xhi)

    def deferred_exact_setup(self, Li, flags_int, pack, theta, gam, dt, dfr, mat, scal):
        tg = theta * gam
        n, nl = Li.nx, Li.ny * Li.nz
        zero = torch.zeros(Li.shape, dtype=torch.float64)
        a, b, c, _ = _line_systems(0, zero, flags_int, pack, theta, gam, dt, 0.0)
        mv = lambda v: v.reshape(n, nl).T
        a, b, c = mv(a), mv(b), mv(c)
        psi_lo, psi_hi = np.zeros((nl, n)), np.zeros((nl, n))
        C = np.zeros((6, nl))
        w = dfr['w'].numpy()
        scal[0] = float(w[0]); scal[1] = float(w[n - 1])
        for l in range(nl):
            Ainv = np.linalg.inv(_dense(a[l], b[l], c[l]))
            if a[l, 0] != 0.0:
                psi_lo[l] = tg * Ainv[:, 0]
            if c[l, -1] != 0.0:
                psi_hi[l] = tg * Ainv[:, -1]
            C[1, l] = -psi_lo[l, 0]; C[2, l] = -psi_hi[l, 0]; C[4, l] = -psi_lo[l, -1]; C[5, l] = -psi_hi[l, -1]
        mat.view(4, nl).numpy()[:] = C[[1, 2, 4, 5]]
        return dict(psi_lo=psi_lo, psi_hi=psi_hi, w=dfr['w'].numpy().copy())

    def deferred_exact_coef(self, dx_, xlo, xhi, nlines, clo, chi):
        w = dx_['w']
        B = np.stack([w, w[::-1]], axis=1)                         # (n, 2)
        t = xlo.numpy()[:, None] * dx_['psi_lo'] + xhi.numpy()[:, None] * dx_['psi_hi']      # (nlines, n)
        sol, res, _, _ = np.linalg.lstsq(B, t.T, rcond=None)
        assert np.abs(B @ sol - t.T).max() <= 1e-9 * max(1.0, np.abs(t).max())           # the two-vector form is exact
        clo.numpy()[:] = sol[0]; chi.numpy()[:] = sol[1]

    def sweep_corrected(self, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, t_out, ulo, uhi, w_corr):
        w = w_corr.numpy()
        x = t_in.numpy().copy()
        if ulo is not None:
            x += w[:, None, None] * ulo.numpy().reshape(1, Li.ny, Li.nz)
        if uhi is not None:
            x += w[::-1][:, None, None] * uhi.numpy().reshape(1, Li.ny, Li.nz)
        self.sweep(1, variant, Li, torch.from_numpy(x), flags, pack, theta, gam, dt, Tinf, t_out)

    def deferred_lines_apply(self, Li, x, cells, wc, u, from_high_end, nrows=None):
        """the flagged lines of one side get their own weights, in place (include/adi_hip.h, adi_deferred_lines_apply)"""
        if cells.numel() == 0:
            return
        xv = x.numpy().reshape(Li.nx, -1)                 # (a view: x is a dense (nx, ny, nz) tensor on CPU ranks)
        assert np.shares_memory(xv, x.numpy())
        c = cells.numpy().astype(np.int64)
        K = wc.shape[0]
        uu = u.numpy().reshape(-1)[c]
        rows = np.arange(K)
        planes = (Li.nx - 1 - rows) if from_high_end else rows
        w = wc.numpy()
        if nrows is not None:             # (rows beyond nrows[q] carry exact zeros: the product kernel does not read them)
            assert not np.any(w * (np.arange(K)[:, None] >= nrows.numpy()[None, :]))
        xv[planes[:, None], c[None, :]] += w * uu[None, :]

    # the deferred form with per-line homogeneous solutions (include/adi_hip.h, ABI v17), restated with dense solves
    def homogeneous_solution(self, variant, Li, flags, pack, theta, gam, dt, lower):
        z = torch.zeros(Li.nx, Li.ny, Li.nz, dtype=torch.float64)
        out = torch.empty_like(z)
        one = torch.ones(Li.ny * Li.nz, dtype=torch.float64)
        self.sweep(0, variant, Li, z, flags, (pack[0], pack[1], None, None), theta, gam, dt, 0.0, out,
                   xlo=one if lower else None, xhi=None if lower else one)
        return out

    def interface_deferred_lines(self, first, last, prev_last, next_first, om, nlines, ulo, uhi, uni_lo=None, uni_hi=None,
                                 ulo_uni=None, uhi_uni=None):
        f, l = first.numpy().reshape(-1), last.numpy().reshape(-1)
        lo = np.zeros(nlines); hi = np.zeros(nlines)
        if prev_last is not None:
            gL = prev_last.numpy().reshape(-1)
            wl, wh = om['lo_own'].numpy().reshape(-1), om['hi_prev'].numpy().reshape(-1)
            F = (f + wl * gL) / (1.0 - wl * wh)
            lo = gL + wh * F
        if next_first is not None:
            wh, wl = om['hi_own'].numpy().reshape(-1), om['lo_next'].numpy().reshape(-1)
            hi = (next_first.numpy().reshape(-1) + wl * l) / (1.0 - wl * wh)
        ulo.copy_(torch.from_numpy(lo)); uhi.copy_(torch.from_numpy(hi))
        if ulo_uni is not None:
            ulo_uni.copy_(torch.from_numpy(lo * (uni_lo.numpy().reshape(-1) != 0) if uni_lo is not None else 0.0 * lo))
        if uhi_uni is not None:
            uhi_uni.copy_(torch.from_numpy(hi * (uni_hi.numpy().reshape(-1) != 0) if uni_hi is not None else 0.0 * hi))

    def sweep(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, t_out, xlo=None, xhi=None):
        a, b, c, d = _line_systems(axis, t_in, flags, pack, theta, gam, dt, Tinf)
        a = np.moveaxis(a, axis, -1).copy(); b = np.moveaxis(b, axis, -1).copy()
        c = np.moveaxis(c, axis, -1).copy(); d = np.moveaxis(d, axis, -1).copy()
        shp = d.shape
        n = shp[-1]
        a = a.reshape(-1, n); b = b.reshape(-1, n); c = c.reshape(-1, n); d = d.reshape(-1, n)
        x = np.empty_like(d)
        for l in range(d.shape[0]):
            dd = d[l].copy()
            if xlo is not None:
                dd[0] -= a[l, 0] * float(xlo[l])
            if xhi is not None:
                dd[-1] -= c[l, -1] * float(xhi[l])
            x[l] = np.linalg.solve(_dense(a[l], b[l], c[l]), dd)
        t_out.copy_(torch.from_numpy(np.moveaxis(x.reshape(shp), -1, axis).copy()))

    def condense(self, axis, variant, Li, t_in, flags, pack, theta, gam, dt, Tinf, cond):
        a, b, c, d = _line_systems(axis, t_in, flags, pack, theta, gam, dt, Tinf)
        mv = lambda v: np.moveaxis(v, axis, -1).reshape(-1, v.shape[axis])
        a, b, c, d = mv(a), mv(b), mv(c), mv(d)
        nl, n = d.shape
        out = cond.view(6, nl).numpy()
        for l in range(nl):
            Binv = np.linalg.inv(_dense(a[l], b[l], c[l]))
            g = Binv @ d[l]
            out[0, l] = g[0]; out[1, l] = a[l, 0] * Binv[0, 0]; out[2, l] = c[l, -1] * Binv[0, -1]
            out[3, l] = g[-1]; out[4, l] = a[l, 0] * Binv[-1, 0]; out[5, l] = c[l, -1] * Binv[-1, -1]

    def interface(self, cond_all, world, rank, nlines, xlo, xhi):
        C = cond_all.view(world, 6, nlines).numpy()
        lo = xlo.numpy(); hi = xhi.numpy()
        for l in range(nlines):
            # unknowns (f_0, l_0, f_1, l_1, ...):  f_r + aF_r l_{r-1} + cF_r f_{r+1} = gF_r ; same for l_r
            A = np.eye(2 * world); rhs = np.zeros(2 * world)
            for r in range(world):
                gF, aF, cF, gL, aL, cL = C[r, :, l]
                rhs[2 * r] = gF; rhs[2 * r + 1] = gL
                if r > 0:
                    A[2 * r, 2 * r - 1] += aF; A[2 * r + 1, 2 * r - 1] += aL
                if r < world - 1:
                    A[2 * r, 2 * r + 2] += cF; A[2 * r + 1, 2 * r + 2] += cL
            z = np.linalg.solve(A, rhs)
            lo[l] = z[2 * rank - 1] if rank > 0 else 0.0
            hi[l] = z[2 * rank + 2] if rank < world - 1 else 0.0

    def interface_pair(self, my_lo, my_hi, prev_hi, next_lo, nlines, xlo, xhi):
        """neighbour-only interface values: per line the 2x2 system between the last window of one slab and the
        first window of the next, written out with a dense solve (independent of the product's closed form)"""
        lo = xlo.numpy(); hi = xhi.numpy()
        lo[:] = 0.0; hi[:] = 0.0
        if prev_hi is not None:
            P = prev_hi.view(3, nlines).numpy(); C = my_lo.view(6, nlines).numpy()
            for l in range(nlines):       # unknowns (last of slab below, my first)
                M = np.array([[1.0, P[2, l]], [C[1, l], 1.0]])
                lo[l] = np.linalg.solve(M, np.array([P[0, l], C[0, l]]))[0]
        if next_lo is not None:
            N = next_lo.view(2, nlines).numpy(); C = my_hi.view(6, nlines).numpy()
            for l in range(nlines):       # unknowns (my last, first of slab above)
                M = np.array([[1.0, C[5, l]], [N[1, l], 1.0]])
                hi[l] = np.linalg.solve(M, np.array([C[3, l], N[0, l]]))[1]
