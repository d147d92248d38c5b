"""Register / scratch footprint of every gfx950 kernel of libadi_hip.so, read from the code objects.

    python scripts/kernel_meta.py [--scratch-only] [--md]

For each object file under adi_thermal_fields_amd/csrc the gfx950 code object is unbundled
(llvm-objcopy --dump-section=.hip_fatbin, clang-offload-bundler --unbundle) and the AMDGPU metadata note is read
(llvm-readelf --notes): VGPRs, spilled VGPRs / SGPRs, bytes of scratch (.private_segment_fixed_size), LDS.
No GPU needed.  tests/test_kernel_footprint.py asserts the scratch figures; DESIGN.md's kernel table is --md.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'adi_thermal_fields_amd', 'csrc')
LLVM = os.environ.get('ADI_LLVM_BIN', '/opt/rocm/llvm/bin')
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
_KEYS = ('name', 'private_segment_fixed_size', 'group_segment_fixed_size', 'sgpr_count', 'sgpr_spill_count',
         'vgpr_count', 'agpr_count', 'vgpr_spill_count', 'max_flat_workgroup_size')


def _demangle(names):
    if not names:
        return []
    out = subprocess.run(['c++filt'], input='\n'.join(names) + '\n', capture_output=True, text=True, check=True).stdout.splitlines()
    short = []
    for d in out:
        d = re.sub(r'^void ', '', d)
        m = re.match(r'(adi::[A-Za-z0-9_]+(<[^(]*>)?)\(', d)
        short.append(m.group(1) if m else d.split('(')[0])
    return short


def object_kernels(obj):
    """[{name, short, scratch, lds, vgpr, agpr, vgpr_spill, sgpr_spill, wg}] of one .o (empty when it holds no device code)"""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section=.hip_fatbin=' + fat, obj, os.path.join(tmp, 'x.o')],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                        '--targets=' + TARGET, '--output=' + co], check=True, capture_output=True)
        if os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    # amdhsa.kernels is a YAML list: an entry starts with "  - .key:" and its own keys sit at four spaces of indentation
    # (argument entries are nested deeper and are skipped by the indentation test)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r'^(  - |    )\.([a-z_]+):\s*(\S*)\s*$', line)
        if not m:
            continue
        if m.group(1) == '  - ':
            cur = {}
            kernels.append(cur)
        if cur is not None and m.group(2) in _KEYS:
            cur[m.group(2)] = m.group(3) if m.group(2) == 'name' else int(m.group(3))
    kernels = [k for k in kernels if 'vgpr_count' in k and 'name' in k]
    for k, s in zip(kernels, _demangle([k['name'] for k in kernels])):
        k['short'] = s
        k['scratch'] = k.get('private_segment_fixed_size', 0)
        k['lds'] = k.get('group_segment_fixed_size', 0)
        k['obj'] = os.path.basename(obj)
    return kernels


def all_kernels(csrc=CSRC):
    out = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith('.o'):
            out.extend(object_kernels(os.path.join(csrc, f)))
    return out


def main():
    ks = all_kernels()
    if '--scratch-only' in sys.argv:
        ks = [k for k in ks if k['scratch']]
    md = '--md' in sys.argv
    if md:
        print('| kernel | object | VGPR | spilled VGPR | spilled SGPR | scratch B | workgroup |')
        print('|---|---|---|---|---|---|---|')
    for k in sorted(ks, key=lambda k: (k['obj'], k['short'])):
        row = (k['short'], k['obj'], k['vgpr_count'], k.get('vgpr_spill_count', 0), k.get('sgpr_spill_count', 0), k['scratch'],
               k.get('max_flat_workgroup_size', 0))
        print(('| `%s` | %s | %d | %d | %d | %d | %d |' if md else '%-70s %-26s vgpr %3d  vspill %3d  sspill %3d  scratch %4d  wg %4d') % row)
    print('%d kernels, %d with scratch' % (len(ks), sum(1 for k in ks if k['scratch'])), file=sys.stderr)


if __name__ == '__main__':
    main()
