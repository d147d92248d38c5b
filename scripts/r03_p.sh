#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03p_pytest python -m pytest tests -m gpu -q" \
  "300 r03p_reh8_strong python bench.py --rehearse-world 8 --scaling strong --no-cpu" \
  "300 r03p_reh8_strong_rccl python bench.py --rehearse-world 8 --scaling strong --no-cpu --force-dist" \
  "300 r03p_reh8_weak python bench.py --rehearse-world 8 --no-cpu" \
  "300 r03p_map python scripts/perf_map.py"
