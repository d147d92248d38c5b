"""CPU: DESIGN.md's "state of HEAD" kernel table (section 0.2) describes the build in the tree -- the VGPR, spill and scratch columns
are compared with the gfx950 code objects under csrc/ (scripts/kernel_meta.py).  Regenerate the table with
`python scripts/design_state.py r04_z` after a kernel change."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import kernel_meta  # noqa: E402


def test_state_table_matches_the_code_objects():
    if not os.path.isdir(kernel_meta.LLVM):
        pytest.skip('no ROCm LLVM tools')
    txt = open(os.path.join(ROOT, 'DESIGN.md')).read()
    a, b = txt.index('<!-- BEGIN STATE TABLE'), txt.index('<!-- END STATE TABLE -->')
    rows = [ln for ln in txt[a:b].splitlines() if ln.startswith('| `')]
    assert len(rows) >= 12, len(rows)
    meta = {k['short']: k for k in kernel_meta.all_kernels()}
    for ln in rows:
        cells = [c.strip() for c in ln.strip('|').split('|')]
        name = 'adi::' + cells[0].strip('`')
        assert name in meta, 'DESIGN.md lists %s, the library has no such kernel' % name
        m = meta[name]
        assert (int(cells[3]), int(cells[4]), int(cells[5])) == (m['vgpr_count'], m.get('vgpr_spill_count', 0), m['scratch']), \
            (name, cells[3:6], m['vgpr_count'], m.get('vgpr_spill_count', 0), m['scratch'])


def test_front_section_names_the_evidence_that_exists():
    txt = open(os.path.join(ROOT, 'DESIGN.md')).read()
    part1 = txt[:txt.index('# Part II')]
    for f in set(re.findall(r'`(profiles/[A-Za-z0-9_./*-]+)`', part1)):
        if '*' in f:
            import glob
            assert glob.glob(os.path.join(ROOT, f)), f
        else:
            assert os.path.exists(os.path.join(ROOT, f)), f
