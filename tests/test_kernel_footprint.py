"""CPU: register / scratch footprint of the gfx950 kernels, read from the code objects the build left under csrc/.

Round 3 shipped the 42 B/cell strided GENERAL sweep with 56 B of scratch per lane (two features added behind run-time
branches) and lost 5 - 10 % on it without any test noticing.  scripts/kernel_meta.py unbundles the device code of every object
file (llvm-objcopy / clang-offload-bundler / llvm-readelf, all under /opt/rocm/llvm/bin) and this test pins what the kernel
metadata says: no scratch in any kernel that is launched directly on a whole sweep, a ceiling for the queue-draining forms.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import kernel_meta  # noqa: E402

# kernels behind a FAST kernel (grid-stride loop over the unit queue: surface / Dirichlet tiles only), pass A of the two-pass
# slab forms and the deferred-correction builds may keep a few spilled registers; nothing else may
SCRATCH_CEILING = 80


@pytest.fixture(scope='module')
def kernels():
    if not os.path.isdir(kernel_meta.LLVM):
        pytest.skip('no ROCm LLVM tools at %s' % kernel_meta.LLVM)
    objs = [f for f in os.listdir(kernel_meta.CSRC) if f.endswith('.o')]
    assert objs, 'no object files under csrc/: run `python -m adi_thermal_fields_amd.build` first'
    ks = kernel_meta.all_kernels()
    assert len(ks) > 300, len(ks)
    return ks


def _targs(short):
    m = re.match(r'adi::(\w+)<(.*)>$', short)
    return (m.group(1), [a.strip() for a in m.group(2).split(',')]) if m else (short.replace('adi::', ''), [])


def _is_hot(k):
    """launched on every tile of a sweep / a step stage (not a queue drain, not a slab-only helper)"""
    name, a = _targs(k['short'])
    if name == 'k_sweep_strided':
        # <M, HAS_DIR, HAS_Q, FUSE, WHOLE, FCM, CORR, QUEUED>: the direct launches of the ordinary sweeps
        return a[7] == 'false' and a[6] == 'false'
    if name in ('k_sweep_strided_fast', 'k_sweep_contig', 'k_sweep_contig_fast', 'k_explicit_v5', 'k_build_coeffs',
                'k_build_flags', 'k_cyl_r_fast', 'k_cyl_phi_fast', 'k_cyl_z_fast', 'k_cyl_strided', 'k_cyl_contig'):
        return True
    return False


def test_no_scratch_in_the_kernels_of_a_step(kernels):
    hot = [k for k in kernels if _is_hot(k)]
    assert len(hot) > 150, len(hot)
    bad = ['%s: %d B scratch, %d spilled VGPRs' % (k['short'], k['scratch'], k.get('vgpr_spill_count', 0))
           for k in hot if k['scratch'] != 0 or k.get('vgpr_spill_count', 0) != 0]
    assert not bad, '\n'.join(bad)


def test_the_42_byte_general_sweep_fits_its_register_budget(kernels):
    """k_sweep_strided<8, dir, q, unfused, WHOLE, arrays, no correction, direct>: the north-star data model on the strided
    axes.  Two 512-thread workgroups per CU need <= 128 VGPRs; round 2: 128 + 3 spilled, round 3: 128 + 14 spilled."""
    ks = [k for k in kernels if k['short'] == 'adi::k_sweep_strided<8, true, true, false, true, 2, false, false>']
    assert len(ks) == 1, [k['short'] for k in kernels if 'k_sweep_strided<8, true, true, false, true' in k['short']]
    k = ks[0]
    assert k['scratch'] == 0 and k.get('vgpr_spill_count', 0) == 0
    assert k['vgpr_count'] <= 104, k['vgpr_count']


def test_scratch_ceiling_everywhere_else(kernels):
    bad = ['%s (%s): %d B' % (k['short'], k['obj'], k['scratch']) for k in kernels if k['scratch'] > SCRATCH_CEILING]
    assert not bad, '\n'.join(bad)


def test_deferred_correction_only_in_its_own_translation_unit(kernels):
    """CORR = true builds of 8 / 16 rows per thread live in adi_sweep_strided_gk.o and nowhere else"""
    for k in kernels:
        name, a = _targs(k['short'])
        if name == 'k_sweep_strided' and a[0] in ('8', '16'):
            assert (a[6] == 'true') == (k['obj'] == 'adi_sweep_strided_gk.o'), (k['short'], k['obj'])
