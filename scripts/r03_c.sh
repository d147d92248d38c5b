#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out
cd $R
exec scripts/gpu_seq.sh \
  "600 r03c_pytest python -m pytest tests/test_dist_slab_gpu.py -m gpu -q -x -k deferred" \
  "300 r03c_reh8_weak python bench.py --rehearse-world 8 --no-cpu" \
  "300 r03c_reh8_weak_rccl python bench.py --rehearse-world 8 --no-cpu --force-dist" \
  "300 r03c_reh8_prof cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r03c_reh8_prof -- python3 $R/bench.py --rehearse-world 8 --no-cpu && cp \$(ls $R/$O/r03c_reh8_prof/*/*kernel_stats.csv | head -1) $R/$O/r03c_reh8_kernel_stats.csv"
