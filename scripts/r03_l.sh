#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03l_pytest python -m pytest tests/test_dist_slab_gpu.py tests/test_waam_harness.py -m gpu -q" \
  "300 r03l_reh8_strong python bench.py --rehearse-world 8 --scaling strong --no-cpu" \
  "300 r03l_reh8_strong_rccl python bench.py --rehearse-world 8 --scaling strong --no-cpu --force-dist" \
  "300 r03l_reh8_weak python bench.py --rehearse-world 8 --no-cpu" \
  "300 r03l_reh4_strong python bench.py --rehearse-world 4 --scaling strong --no-cpu"
