"""Build libadi_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m adi_thermal_fields_amd.build [--force] [--verbose]

The .so lands next to the sources (adi_thermal_fields_amd/csrc/libadi_hip.so); it is git-ignored
but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libadi_hip.so')
SOURCES = ['adi_cart_api.hip', 'adi_explicit.hip', 'adi_sweep_contig.hip', 'adi_sweep_contig_x.hip', 'adi_sweep_strided.hip', 'adi_sweep_strided_x.hip', 'adi_condense.hip', 'adi_cyl.hip', 'adi_ctx.hip', 'adi_morph.hip']
HEADERS = ['adi_core.hpp', 'adi_common.hpp', 'adi_cart_dev.hpp', 'adi_cart_host.hpp', 'adi_strided_dev.hpp', 'adi_strided_fast.hpp', 'adi_contig_dev.hpp', os.path.join('..', '..', 'include', 'adi_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-Wall',
         '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or True))
