"""Build libadi_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m adi_thermal_fields_amd.build [--force] [--verbose]

The .so lands next to the sources (adi_thermal_fields_amd/csrc/libadi_hip.so); it is git-ignored
but travels to the GPU box with the gpurun snapshot.  A source or header named below that is missing
is an error here, not a library without kernels later.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libadi_hip.so')
STAMP_SRC = 'adi_stamp.hip'
SOURCES = ['adi_cart_api.hip', 'adi_explicit.hip', 'adi_sweep_contig.hip', 'adi_sweep_contig_x.hip', 'adi_sweep_strided.hip', 'adi_sweep_strided_x.hip', 'adi_sweep_strided_y.hip', 'adi_sweep_strided_fc.hip', 'adi_sweep_strided_fx.hip', 'adi_sweep_strided_fy.hip', 'adi_sweep_strided_gc.hip', 'adi_sweep_strided_gk.hip', 'adi_condense.hip', 'adi_cyl.hip', 'adi_ctx.hip', 'adi_morph.hip', STAMP_SRC]
HEADERS = ['adi_core.hpp', 'adi_common.hpp', 'adi_cart_dev.hpp', 'adi_cart_host.hpp', 'adi_strided_dev.hpp', 'adi_strided_fast.hpp', 'adi_strided_general.hpp', 'adi_contig_dev.hpp', os.path.join('..', '..', 'include', 'adi_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-Wall',
         '-Wno-unused-function']
# __FILE__ (error strings) relative to the package: the same sources give the same library wherever they are checked out
PREFIX_MAP = ['-ffile-prefix-map=%s=csrc' % CSRC, '-ffile-prefix-map=%s=include' % os.path.normpath(os.path.join(CSRC, '..', '..', 'include'))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_stamp():
    """first 16 hex digits of the SHA-256 over the names and contents of every source and header plus the compile flags"""
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(name).encode() + b'\0')
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    missing = [p for p in srcs + hdrs if not os.path.exists(p)]
    if missing:
        raise FileNotFoundError('adi_thermal_fields_amd.build: missing source(s): %s' % ', '.join(missing))
    stamp = source_stamp()
    stamp_file = os.path.join(CSRC, 'adi_stamp.txt')
    old_stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if os.path.basename(s) == STAMP_SRC:
            if force or old_stamp != stamp or not os.path.exists(o):
                jobs.append([HIPCC] + FLAGS + PREFIX_MAP + ['-DADI_SOURCE_STAMP="%s"' % stamp, '-c', s, '-o', o])
        elif force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + PREFIX_MAP + ['-c', s, '-o', o])

    def run(cmd):
        import time
        t0 = time.time()
        subprocess.check_call(cmd)
        if verbose:
            print('%6.1f s  %s' % (time.time() - t0, ' '.join(cmd[-4:])), flush=True)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs)
    with open(stamp_file, 'w') as f:
        f.write(stamp + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or True))
