#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, as MI355X_MICROARCH.md prescribes) of
`bench.py` and `bench.py --config cyl` into profiles/pmc_traffic.json: HBM bytes per launch of each stage kernel, keyed by
bench.py's stage names, stamped with the hash of the library that ran (bench.py reports `roofline.traffic` only when the
stamp matches the library it is running).
gfx950 correction from the guide: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> doubled; WRITE_SIZE is
exact.  Both counters are in KiB.

    python scripts/pmc_summary.py CART_FETCH_DIR CART_WRITE_DIR CYL_FETCH_DIR CYL_WRITE_DIR OUT.json"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(root, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'adi::' in r['Kernel_Name']:
                agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def kernels(fetch_dir, write_dir):
    fetch, write = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
    return {k: dict(fetch_KiB_raw=fetch.get(k), write_KiB=write.get(k),
                    hbm_bytes=(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0) for k in sorted(set(fetch) | set(write))}


def targs(k):
    return [a.strip() for a in k[k.index('<') + 1:k.rindex('>')].split(',')] if '<' in k else []


def so_stamp():
    """adi_build_stamp() of the library in the tree: a hash of its sources and flags, independent of the build directory"""
    sys.path.insert(0, ROOT)
    from adi_thermal_fields_amd import _lib
    return _lib.lib.adi_build_stamp().decode()


cart = kernels(sys.argv[1], sys.argv[2])
cyl = kernels(sys.argv[3], sys.argv[4])


def pick(kern, sub, pred=lambda a: True):
    tot = [v['hbm_bytes'] for k, v in kern.items() if sub in k and pred(targs(k))]
    return sum(tot) if tot else None


# Cartesian stage kernels of the lean step: strided kernels <M, HAS_DIR, HAS_Q, FUSE[, MIXED]>, contiguous <M, MODE, HAS_DIR, HAS_Q>;
# the dense general-pack instantiations (<.., true, true, ..>) are bench.py's separate 42 B/cell measurements, not stages
lean = lambda a: len(a) >= 4 and not (a[1] == 'true' and a[2] == 'true')
out = dict(
    note='HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide reads); a sweep = its FAST '
         'kernel + the GENERAL kernel on the queued units; keys are bench.py stage names',
    library_stamp=so_stamp(),
    cart={'explicit+sweep_axis0': pick(cart, 'k_sweep_strided', lambda a: lean(a) and a[3] == 'true'),
          'sweep_axis1': pick(cart, 'k_sweep_strided', lambda a: lean(a) and a[3] == 'false'),
          'sweep_axis2_contig': pick(cart, 'k_sweep_contig', lambda a: len(a) >= 4 and not (a[2] == 'true' and a[3] == 'true'))},
    cyl={'sweep_r': pick(cyl, 'k_cyl_r_fast') or pick(cyl, 'k_cyl_strided', lambda a: a[1:] == ['0']),
         'sweep_phi': pick(cyl, 'k_cyl_phi_fast') or pick(cyl, 'k_cyl_strided', lambda a: a[1:] == ['1']),
         'sweep_z_contig': pick(cyl, 'k_cyl_z_fast') or pick(cyl, 'k_cyl_contig')},
    kernels=dict(cart=cart, cyl=cyl))
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}, indent=1))
