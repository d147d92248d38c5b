"""One-off soak with the generators of tests/test_hip_fuzz.py: N more Cartesian and M more cylindrical seeded cases, HIP
against the CPU oracle; prints the worst relative L-inf and every case above 1e-10.
    python scripts/fuzz_soak.py [N=600] [M=300] [first_seed=1000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_hip_fuzz as F
from helpers import rel_linf, run_cart_case
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
import adi_thermal_fields_amd.adi3d_hip_cyl as hipcyl
from oracle import adi_oracle as orc
from oracle import cyl_oracle as cyl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
S0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
t0 = time.time()
worst, bad, kinds = 0.0, [], {}
for seed in range(S0, S0 + N):
    c, tag = F._case(seed)
    got = run_cart_case(hip, c); want = run_cart_case(orc, c)
    err = max(rel_linf(got[k], want[k]) for k in ('T_step1', 'T_final'))
    ok_off = np.array_equal(got['T_final'][~c['mask']], c['T0'][~c['mask']])
    worst = max(worst, err)
    k2 = (str(tag[0]), str(tag[1])); kinds[k2] = kinds.get(k2, 0) + 1
    if not (err <= 1e-10) or not ok_off:
        bad.append((seed, c['shape'], tag, err, ok_off))
    if (seed - S0) % 100 == 99:
        print('cartesian %d / %d, worst so far %.3e, %.0f s' % (seed - S0 + 1, N, worst, time.time() - t0), flush=True)
print('cartesian: %d cases (seeds %d..%d), worst rel L-inf %.3e, failures %s' % (N, S0, S0 + N - 1, worst, bad))
print('  mask kind x bc kind counts:', dict(sorted(kinds.items())))
worst_c, bad_c = 0.0, []
for seed in range(S0, S0 + M):
    c, mode = F._cyl_case(seed)
    got = F._run_cyl(hipcyl, c); want = F._run_cyl(cyl, c)
    err = rel_linf(got, want)
    worst_c = max(worst_c, err)
    if not err <= 1e-10:
        bad_c.append((seed, c['shape'], mode, err))
print('cylindrical: %d cases, worst rel L-inf %.3e, failures %s' % (M, worst_c, bad_c))
print('total %.0f s' % (time.time() - t0))
sys.exit(1 if (bad or bad_c) else 0)
