#!/bin/bash
# round 3, second GPU call: the deferred form of the sharded-axis sweep -- slab tests, rehearsals, kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out
cd $R
exec scripts/gpu_seq.sh \
  "900 r03b_pytest python -m pytest tests/test_dist_slab_gpu.py tests/test_waam_harness.py tests/test_hip_ctx_api.py -m gpu -q -x" \
  "300 r03b_reh8_weak python bench.py --rehearse-world 8 --no-cpu" \
  "300 r03b_reh8_weak_rccl python bench.py --rehearse-world 8 --no-cpu --force-dist" \
  "300 r03b_reh8_prof cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r03b_reh8_prof -- python3 $R/bench.py --rehearse-world 8 --no-cpu && cp \$(ls $R/$O/r03b_reh8_prof/*/*kernel_stats.csv | head -1) $R/$O/r03b_reh8_kernel_stats.csv"
