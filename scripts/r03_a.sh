#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite, the default bench line (with parity_rel_linf), the rehearsals of a
# middle rank of 8 (weak and strong, loopback and RCCL self-loop), the ellipsoid line, and a kernel trace of the weak rehearsal
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out
cd $R
exec scripts/gpu_seq.sh \
  "900 r03a_pytest python -m pytest tests -m gpu -q" \
  "600 r03a_bench python bench.py" \
  "300 r03a_reh8_weak python bench.py --rehearse-world 8 --no-cpu" \
  "300 r03a_reh8_weak_rccl python bench.py --rehearse-world 8 --no-cpu --force-dist" \
  "300 r03a_reh8_strong python bench.py --rehearse-world 8 --scaling strong --no-cpu" \
  "300 r03a_reh8_strong_rccl python bench.py --rehearse-world 8 --scaling strong --no-cpu --force-dist" \
  "300 r03a_ellipsoid python bench.py --mask ellipsoid --no-cpu" \
  "300 r03a_reh8_prof cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r03a_reh8_prof -- python3 $R/bench.py --rehearse-world 8 --no-cpu && cp \$(ls $R/$O/r03a_reh8_prof/*/*kernel_stats.csv | head -1) $R/$O/r03a_reh8_kernel_stats.csv"
