#!/bin/bash
# A/B build of the kernels: scripts/ab_build.sh NAME "<extra hipcc flags>" file1.hip [file2.hip ...]
# compiles the named translation units of csrc/ with the extra flags and links them with the standard objects of the
# other units into scripts/_build/libadi_NAME.so (git-ignored; select it with ADI_HIP_LIB=scripts/_build/libadi_NAME.so).
set -e
NAME=$1; EXTRA=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adi_thermal_fields_amd/csrc
B=$R/scripts/_build/$NAME
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function"
objs=""
for o in $C/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$(basename $f .hip)" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
pids=""
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $C/$b.hip -o $B/$b.o &
  pids="$pids $!"
  objs="$objs $B/$b.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/_build/libadi_$NAME.so $objs
echo built scripts/_build/libadi_$NAME.so
