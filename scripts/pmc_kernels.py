#!/usr/bin/env python3
"""Mean per-launch value of every PMC counter per kernel, from one or more rocprofv3 --pmc output directories.
    python scripts/pmc_kernels.py DIR [DIR ...] [--match SUBSTR]"""
import collections, csv, glob, os, sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
match = None
if '--match' in sys.argv:
    match = sys.argv[sys.argv.index('--match') + 1]
    args.remove(match)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'adi::' not in k or (match and match not in k):
                continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-28s %16.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
