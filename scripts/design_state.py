"""The "state of HEAD" kernel table of DESIGN.md section 0, generated -- not typed -- from
  * the code objects of the build in the tree (scripts/kernel_meta.py: VGPRs, spilled VGPRs, scratch bytes),
  * the rocprofv3 --kernel-trace --stats summaries kept under profiles/ (average duration per kernel),
  * the bench line kept beside them (stage times by HIP events; `also`).
    python scripts/design_state.py [TAG]           (default r04_z)  -> markdown on stdout
tests/test_design_state.py checks that the table in DESIGN.md carries the register / scratch figures of the build in the tree."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import kernel_meta  # noqa: E402

PEAK = 8000.0
N512 = 512 ** 3
NCYL = 128 * 256 * 512
# (kernel as rocprof / the metadata names it, what it is, bytes per cell, cells per launch, stats file)
ROWS = [
    ('adi::k_sweep_strided_fast<16, false, false, true, false, true>', 'explicit + axis-0 sweep, fused FAST (headline step; dominant)', 17, N512, 'cart'),
    ('adi::k_sweep_strided_fast<32, false, false, false, true, true>', 'axis-1 sweep, FAST, 32 rows per thread', 17, N512, 'cart'),
    ('adi::k_sweep_contig_fast<16, 2, false, false>', 'axis-2 (contiguous, "x") sweep, FAST', 17, N512, 'cart'),
    ('adi::k_sweep_contig<8, 2, true, true>', 'contiguous sweep, general pack 42 B/cell: the north-star kernel', 42, N512, 'cart'),
    ('adi::k_sweep_strided<8, true, true, false, true, 2, false, false>', 'strided sweeps (axes 0 and 1 mixed), general pack 42 B/cell', 42, N512, 'cart'),
    ('adi::k_cyl_r_fast<8>', 'cylindrical r sweep (config 4)', 16, NCYL, 'cyl'),
    ('adi::k_cyl_phi_fast<16>', 'cylindrical phi sweep', 16, NCYL, 'cyl'),
    ('adi::k_cyl_z_fast<16>', 'cylindrical z sweep', 16, NCYL, 'cyl'),
]
# further kernels listed with their footprint only (no per-kernel time in the kept traces)
FOOT = [
    ('adi::k_sweep_strided_fast<16, false, false, true, true, true>', 'fused FAST build with the surface-segment lanes (curved solids), coefficients from the flags'),
    ('adi::k_sweep_strided_fast<16, false, false, true, true, false>', 'the same, coefficients loaded (per-voxel h)'),
    ('adi::k_sweep_strided<8, true, true, true, false, 2, false, false>', 'fused GENERAL kernel, direct launch (dense / hand-built packs)'),
    ('adi::k_sweep_strided<8, true, true, true, false, 2, false, true>', 'fused GENERAL kernel draining the FAST kernel\'s queue'),
    ('adi::k_sweep_strided<8, true, true, false, true, 0, true, false>', 'axis-1 GENERAL kernel with the deferred slab correction'),
    ('adi::k_explicit_v5<2, false>', 'explicit stage as its own kernel'),
]


def stats(tag, which):
    out = {}
    with open(os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.csv' % (tag, which))) as f:
        for r in csv.DictReader(f):
            out[r['Name'].split('(')[0].replace('void ', '')] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04_z'
    meta = {k['short']: k for k in kernel_meta.all_kernels()}
    st = {w: stats(tag, w) for w in ('cart', 'cyl')}
    print('| kernel (name in the rocprof trace) | what | B/cell | VGPRs | spilled | scratch B | rocprof avg us | TB/s | of 8 TB/s |')
    print('|---|---|---|---|---|---|---|---|---|')
    for name, what, bpc, cells, which in ROWS:
        m = meta[name]
        calls, us = st[which][name]
        tbs = bpc * cells / (us * 1e-6) / 1e12
        print('| `%s` | %s | %d | %d | %d | %d | %.1f | %.2f | %.3f |' % (name.replace('adi::', ''), what, bpc, m['vgpr_count'], m.get('vgpr_spill_count', 0),
                                                                          m['scratch'], us, tbs, tbs * 1e3 / PEAK))
    for name, what in FOOT:
        m = meta.get(name)
        if m is not None:
            print('| `%s` | %s | | %d | %d | %d | | | |' % (name.replace('adi::', ''), what, m['vgpr_count'], m.get('vgpr_spill_count', 0), m['scratch']))
    d = None
    for ln in open(os.path.join(ROOT, 'profiles', '%s_bench_line.json' % tag)):
        if ln.startswith('{'):
            d = json.loads(ln)
    print()
    print('Bench line `profiles/%s_bench_line.json` (HIP events, 50 steps): **%.1f steps/s, %.4f ms/step**; stages %s; 42 B/cell sweeps %s; '
          'parity %.1e; CPU %.4f steps/s on 1 core, %.3f on %d cores.' %
          (tag, d['value'], d['ms_per_step'], ', '.join('%s %.3f ms (%.3f)' % (k, v['ms'], v['frac']) for k, v in d['kernels'].items()),
           ', '.join('%s %.3f ms (%.3f)' % (k, v['ms'], v['frac']) for k, v in d['general_pack_sweeps_42B'].items()),
           d['parity_rel_linf'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], d['cpu_baseline_all_cores']['cores']))
    print('`also`: ' + '; '.join('%s %.4f ms/step' % (k, v['ms_per_step']) for k, v in d['also'].items()) + '.')
    ks = list(meta.values())
    print('%d kernels in the library, %d of them with scratch (all queue-draining or slab-only builds, <= %d B).' %
          (len(ks), sum(1 for k in ks if k['scratch']), max(k['scratch'] for k in ks)))


if __name__ == '__main__':
    main()
