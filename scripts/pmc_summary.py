#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/pmc_traffic.json: HBM bytes per launch for each stage kernel of bench.py.
gfx950 correction from the guide: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> doubled;
WRITE_SIZE is exact.  Both counters are in KiB."""
import collections, csv, glob, json, sys

def load(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'adi::' in r['Kernel_Name']:
                agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}

fetch = load(sys.argv[1] + '/**/*counter_collection.csv', 'FETCH_SIZE') if True else {}
import os
fetch = {}
for root in (sys.argv[1],):
    fetch = load(os.path.join(root, '*', '*counter_collection.csv'), 'FETCH_SIZE')
write = load(os.path.join(sys.argv[2], '*', '*counter_collection.csv'), 'WRITE_SIZE')
kern = {}
for k in sorted(set(fetch) | set(write)):
    kern[k] = dict(fetch_KiB_raw=fetch.get(k), write_KiB=write.get(k),
                   hbm_bytes=(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0)
def pick(sub):
    # the dense general-pack instantiations (<.., true, true>) are bench.py's separate 42 B/cell measurements
    tot = [v['hbm_bytes'] for k, v in kern.items() if sub in k and 'true, true>' not in k]
    return sum(tot) if tot else None
def pick_fused(sub, fused):
    # template arguments of the strided kernels: <M, HAS_DIR, HAS_Q, FUSE[, MIXED]>
    tot = []
    for k, v in kern.items():
        if sub not in k or '<' not in k:
            continue
        args = [a.strip() for a in k[k.index('<') + 1:k.rindex('>')].split(',')]
        if len(args) < 4 or (args[1] == 'true' and args[2] == 'true'):     # dense general-pack measurements: not a stage
            continue
        if (args[3] == 'true') == fused:
            tot.append(v['hbm_bytes'])
    return sum(tot) if tot else None
out = dict(note='HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide reads); '
                'sweeps = FAST kernel + GENERAL kernel on the queued units; keys are bench.py stage names',
           explicit=pick('k_explicit'), sweep_axis0=pick_fused('k_sweep_strided', False),
           sweep_axis1=pick_fused('k_sweep_strided', False), sweep_axis2_contig=pick('k_sweep_contig'),
           kernels=kern)
out['explicit+sweep_axis0'] = pick_fused('k_sweep_strided', True)
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}, indent=1))
