"""Shared, reference-free definitions of the parity cases.

Every case is a plain dict of inputs built by formula or a seeded NumPy RNG.  The same
definitions are consumed by
  * tests/golden/make_golden.py  (imports the REFERENCE in the build container and stores
    its outputs next to the inputs as tests/golden/<case>.npz), and
  * the parity tests (oracle vs golden on CPU; HIP vs oracle / golden on the GPU).
Nothing here reads /root/reference.
"""
import numpy as np

STEEL = dict(rho=7800.0, cp=490.0, k=54.0)  # SURVEY.md 8(c) KAT material
FACES = ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')


def _t0_formula(shape):
    i, j, k = np.meshgrid(*[np.arange(n) for n in shape], indexing='ij')
    return 20.0 + 100.0 * np.sin(0.3 * i) * np.cos(0.2 * j) + 3.0 * k


def _alpha(mat):
    return mat['k'] / (mat['rho'] * mat['cp'])


# ------------------------------------------------------------------ Cartesian
def cart_case(name):
    """Return dict(shape, dx, mat, mask, T0, dir_mask, dir_value, neumann, robin_h,
    Tinf, dt, theta, nsteps [, births])."""
    mat = dict(STEEL)
    c = dict(name=name, mat=mat, dir_mask=None, dir_value=None, neumann=None, robin_h=None,
             Tinf=20.0, theta=0.5, births=None)
    if name == 'kat1':
        s = (16, 16, 16)
        c.update(shape=s, dx=1e-3, mask=np.ones(s, bool), T0=_t0_formula(s), robin_h=500.0,
                 dt=0.05, nsteps=3)
    elif name == 'kat2':  # disk, Dirichlet top, Neumann bottom, side Robin
        s = (12, 12, 20)
        dx = 0.02 / 6
        X = (np.arange(12) + 0.5 - 6) * dx
        disk = np.sqrt(X[:, None] ** 2 + X[None, :] ** 2) <= 0.02 + 1e-12
        mask = np.repeat(disk[:, :, None], 20, axis=2)
        dm = np.zeros(s, bool); dm[:, :, -1] = mask[:, :, -1]
        c.update(shape=s, dx=dx, mask=mask, T0=np.full(s, 20.0), dir_mask=dm, dir_value=20.0,
                 neumann={'z-': 2e6}, robin_h={'x-': 500.0, 'x+': 500.0, 'y-': 500.0, 'y+': 500.0},
                 dt=0.5, nsteps=3)
    elif name == 'holes_mixed':  # 30 % random holes, every BC kind, array- and scalar-valued, theta=1
        s = (13, 11, 17)
        rng = np.random.default_rng(7)
        mask = rng.random(s) > 0.3
        dm = (rng.random(s) > 0.93) & mask
        dv = rng.uniform(50.0, 300.0, s)
        dx = 5e-4
        c.update(shape=s, dx=dx, mask=mask, T0=rng.uniform(20.0, 1000.0, s), dir_mask=dm, dir_value=dv,
                 neumann={'x+': rng.uniform(0.0, 1e6, s), 'y-': 3e5, 'z+': None},
                 robin_h={'x-': 400.0, 'y+': rng.uniform(10.0, 900.0, s), 'z-': 50.0, 'z+': 1200.0},
                 theta=1.0, dt=200.0 * dx * dx / _alpha(mat), nsteps=4)
    elif name == 'robin_array_stiff':  # one 3-D h array for all faces, gamma = 3000
        s = (10, 12, 9)
        rng = np.random.default_rng(11)
        dx = 1e-3
        c.update(shape=s, dx=dx, mask=np.ones(s, bool), T0=rng.uniform(20.0, 1000.0, s),
                 robin_h=rng.uniform(0.0, 2000.0, s), Tinf=35.0,
                 dt=3000.0 * dx * dx / _alpha(mat), nsteps=3)
    elif name == 'edge_shapes':  # unit-length axis, isolated cells, empty lines
        s = (1, 7, 9)
        rng = np.random.default_rng(3)
        mask = rng.random(s) > 0.5
        mask[0, 3, :] = False          # an empty line along z
        mask[0, 5, :] = False; mask[0, 5, 4] = True   # an isolated cell
        dx = 2e-3
        c.update(shape=s, dx=dx, mask=mask, T0=rng.uniform(0.0, 500.0, s), robin_h=150.0,
                 neumann={'z-': 1e5, 'x-': 2e4}, dt=0.7 * dx * dx / _alpha(mat), nsteps=3)
    elif name == 'empty_mask':  # nothing is solid: the step must return T unchanged
        s = (5, 6, 7)
        c.update(shape=s, dx=1e-3, mask=np.zeros(s, bool), T0=_t0_formula(s), robin_h=500.0,
                 dt=0.05, nsteps=2)
    elif name == 'dirichlet_only_gamma07':  # Dirichlet planes both ends of z, no Robin (robin_h=None)
        s = (9, 8, 21)
        dm = np.zeros(s, bool); dm[:, :, 0] = True; dm[:, :, -1] = True
        dv = np.zeros(s); dv[:, :, 0] = 1000.0; dv[:, :, -1] = 20.0
        dx = 1e-3
        c.update(shape=s, dx=dx, mask=np.ones(s, bool), T0=np.full(s, 20.0), dir_mask=dm, dir_value=dv,
                 dt=0.7 * dx * dx / _alpha(mat), nsteps=5)
    elif name == 'long_line_70':  # lines longer than one 64-lane wave tile in every direction is costly;
        s = (70, 5, 66)           # two long axes exercise multi-segment partitions with ragged tails
        rng = np.random.default_rng(21)
        mask = rng.random(s) > 0.1
        dx = 1e-3
        c.update(shape=s, dx=dx, mask=mask, T0=rng.uniform(20.0, 1500.0, s), robin_h=300.0,
                 neumann={'x-': 5e5}, dt=50.0 * dx * dx / _alpha(mat), nsteps=2)
    elif name == 'birth_sequence':  # layer birth along z: mask grows, packs rebuilt (waam_from_stl_v7_mm.py:487-550)
        s = (12, 12, 16)
        dx = 1e-3
        X = (np.arange(12) + 0.5 - 6)
        disk = (X[:, None] ** 2 + X[None, :] ** 2) <= 5.2 ** 2
        full = np.repeat(disk[:, :, None], 16, axis=2)
        base = full.copy(); base[:, :, 4:] = False
        births = [(4, 8), (8, 12), (12, 16)]  # z-ranges activated in turn
        c.update(shape=s, dx=dx, mask=base, T0=np.where(base, 200.0, 20.0), robin_h=40.0, full_mask=full,
                 births=births, Ts=1000.0, dt=20.0 * dx * dx / _alpha(mat), nsteps=2)  # nsteps per layer
    elif name == 'slab_chunks':  # ny >= 32: the distributed axis-0 sweep is pipelined over 4 chunks of lines
        s = (9, 34, 6)
        rng = np.random.default_rng(31)
        mask = rng.random(s) > 0.15
        dx = 1e-3
        c.update(shape=s, dx=dx, mask=mask, T0=rng.uniform(20.0, 900.0, s), robin_h={'x-': 200.0, 'x+': 350.0, 'z+': 80.0},
                 neumann={'y-': 2e5}, dt=120.0 * dx * dx / _alpha(mat), nsteps=2)
    elif name == 'config1_64':  # BASELINE.json configs[0] / SURVEY.md 8(d) config 1
        s = (64, 64, 64)
        dm = np.zeros(s, bool); dm[:, :, 0] = True; dm[:, :, -1] = True
        dv = np.zeros(s); dv[:, :, 0] = 1000.0; dv[:, :, -1] = 20.0
        dx = 1e-3
        c.update(shape=s, dx=dx, mask=np.ones(s, bool), T0=np.full(s, 20.0), dir_mask=dm, dir_value=dv,
                 robin_h={'x-': 500.0, 'x+': 500.0, 'y-': 500.0, 'y+': 500.0},
                 dt=2.0 * dx * dx / _alpha(mat), nsteps=100)
    else:
        raise KeyError(name)
    return c


CART_CASES = ['kat1', 'kat2', 'holes_mixed', 'robin_array_stiff', 'edge_shapes', 'empty_mask',
              'dirichlet_only_gamma07', 'long_line_70', 'birth_sequence']
CART_STAGE_CASE = 'holes_mixed'   # per-stage dumps (Lx, Ly, Lz, R0, U, V, W) of the first step
CART_SLOW_CASES = ['config1_64']  # generated once (minutes of CPython), stored as planes + checksums


# ---------------------------------------------------------------- cylindrical
def cyl_case(name):
    """Return dict(shape(nr,nphi,nz), dr, dphi, dz, R, mat, T0, robin_r(h,T_inf), zbc(dict), dt, nsteps,
    S or None, active or None)."""
    mat = dict(STEEL)
    c = dict(name=name, mat=mat, S=None, active=None)

    def geom(nr, nphi, nz, dr, dz):
        return dict(shape=(nr, nphi, nz), dr=dr, dz=dz, dphi=2.0 * np.pi / nphi, R=nr * dr)

    if name == 'kat3':
        c.update(geom(8, 16, 12, 1e-3, 1e-3))
        c.update(T0=_t0_formula((8, 16, 12)), robin_r=(400.0, 20.0),
                 zbc=dict(kind_bot='neumann0', kind_top='robin', h_top=500.0, T_inf_top=20.0),
                 dt=0.05, nsteps=3)
    elif name.startswith('zbc_'):  # zbc_<bot>_<top>
        _, kb, kt = name.split('_')
        c.update(geom(6, 8, 10, 8e-4, 1.1e-3))
        rng = np.random.default_rng(_KINDS.index(kb) * 3 + _KINDS.index(kt) + 100)
        c.update(T0=rng.uniform(20.0, 900.0, (6, 8, 10)), robin_r=(250.0, 30.0),
                 zbc=dict(kind_bot=kb, kind_top=kt, h_bot=120.0, h_top=700.0, T_inf_bot=25.0,
                          T_inf_top=15.0, T_bot=300.0, T_top=80.0),
                 dt=0.3, nsteps=2)
    elif name == 'nphi1_source':  # axisymmetric (nphi == 1 -> phi solve is a copy) with a source term
        c.update(geom(7, 1, 9, 1e-3, 1e-3))
        rng = np.random.default_rng(5)
        c.update(T0=rng.uniform(20.0, 400.0, (7, 1, 9)), robin_r=(0.0, 20.0),  # h == 0 branch
                 zbc=dict(kind_bot='robin', kind_top='robin', h_bot=90.0, h_top=60.0),
                 S=rng.uniform(0.0, 5e8, (7, 1, 9)), dt=0.1, nsteps=3)
    elif name == 'nphi2_source':  # two phi cells: both neighbours of a cell are the SAME cell (the folded branch of the
        c.update(geom(6, 2, 9, 7e-4, 9e-4))                # periodic solve; numpy.fft handles n = 2 like any other n)
        rng = np.random.default_rng(17)
        c.update(T0=rng.uniform(20.0, 800.0, (6, 2, 9)), robin_r=(300.0, 25.0),
                 zbc=dict(kind_bot='dirichlet', kind_top='robin', h_top=200.0, T_inf_top=18.0, T_bot=120.0),
                 S=rng.uniform(0.0, 2e8, (6, 2, 9)), dt=0.15, nsteps=3)
    elif name == 'nphi36_masked':  # non-power-of-two nphi, the shape of tests/test_spiral_vs_analytic.py (6x36x7)
        c.update(geom(6, 36, 7, 5e-4, 1e-3))
        rng = np.random.default_rng(9)
        active = rng.random((6, 36, 7)) > 0.35
        c.update(T0=rng.uniform(20.0, 1200.0, (6, 36, 7)), robin_r=(35.0, 20.0),
                 zbc=dict(kind_bot='neumann0', kind_top='robin', h_top=35.0, T_inf_top=20.0),
                 active=active, robin_inner=(10.0, 45.0), robin_void=(5.0, 27.0), dt=0.2, nsteps=3)
    elif name == 'long_70x12x66':
        c.update(geom(70, 12, 66, 3e-4, 4e-4))
        rng = np.random.default_rng(13)
        c.update(T0=rng.uniform(20.0, 1000.0, (70, 12, 66)), robin_r=(400.0, 20.0),
                 zbc=dict(kind_bot='dirichlet', kind_top='robin', h_top=500.0, T_bot=150.0),
                 dt=0.05, nsteps=2)
    else:
        raise KeyError(name)
    return c


_KINDS = ('neumann0', 'dirichlet', 'robin')
CYL_CASES = (['kat3'] + ['zbc_%s_%s' % (b, t) for b in _KINDS for t in _KINDS]
             + ['nphi1_source', 'nphi2_source', 'nphi36_masked', 'long_70x12x66'])
