"""cylindrical 128 x 256 x 512 BE step: the three sweeps ping-ponging between buffers (shipped) against sweeps run IN PLACE
(every thread of the FAST kernels reads only the rows it later writes), which halves the working set to 134 MB -- inside the
256 MB Infinity Cache.  (adi_cyl_sweep accepts d_out == d_in since ABI v13: this experiment is what decided it.)
    python scripts/cyl_inplace_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_cyl as cyl
from adi_thermal_fields_amd._lib import lib, check

nr, nphi, nz = 128, 256, 512
g = cyl.GridCyl(nr, nphi, nz, 2.5e-4, 2 * np.pi / nphi, 2.5e-4, 0.032)
mat = cyl.Material(7800.0, 490.0, 54.0)
prm = cyl.Params(0.05, 1.0, "be")
st = cyl.StagedCylStepper(g, mat, prm, cyl.RobinR(400.0, 20.0), cyl.ZBC('neumann0', 'robin', h_top=500.0, T_inf_top=20.0))
T0 = np.full((nr, nphi, nz), 20.0); T0[:, :, -16:] = 1000.0
X = cyl.to_device(T0).t
A, B = g.layout.empty(), g.layout.empty()
p = lambda t: cyl._p(t)
h = st.plan.handle
s_ = cyl._stream


def sweep(ax, a, b):
    check(lib.adi_cyl_sweep(h, ax, p(a), p(b), None, None, 0.0, 0.0, s_()))


def pingpong(x):
    sweep(0, x, A); sweep(1, A, B); sweep(2, B, x)


def inplace(x):
    sweep(0, x, x); sweep(1, x, x); sweep(2, x, x)


def r_out_then_inplace(x):
    sweep(0, x, A); sweep(1, A, A); sweep(2, A, A); sweep(0, A, x); sweep(1, x, x); sweep(2, x, x)


def dup(t):
    """a copy in the SAME padded-plane layout (t.clone() would be dense: the kernels address planes sx apart)"""
    c = g.layout.empty()
    c.copy_(t)
    return c


assert g.layout.is_native(X) and g.layout.is_native(A) and g.layout.is_native(B)
ref = dup(X)
for _ in range(4):
    pingpong(ref)
chk = dup(X)
for _ in range(4):
    inplace(chk)
print('in place == ping-pong after 4 steps:', bool(torch.equal(ref, chk)), flush=True)
for name, fn, steps in (('ping-pong (shipped)', pingpong, 1), ('all three in place', inplace, 1), ('r out of place, phi / z in place', r_out_then_inplace, 2)):
    Y = dup(X)
    for _ in range(5):
        fn(Y)
    K = 100
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K):
        fn(Y)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / (K * steps)
    print('%-36s %.4f ms/step  %.0f steps/s' % (name, ms, 1e3 / ms), flush=True)
# per-sweep times in place
for ax, nm in ((0, 'r'), (1, 'phi'), (2, 'z')):
    for mode in ('out of place', 'in place'):
        Y = dup(X)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            sweep(ax, Y, A if mode == 'out of place' else Y)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            sweep(ax, Y, A if mode == 'out of place' else Y)
        e1.record(); e1.synchronize()
        print('  sweep %-3s %-12s %.1f us' % (nm, mode, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
