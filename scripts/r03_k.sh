#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03k_pytest python -m pytest tests -m gpu -q" \
  "300 r03k_mask python scripts/mask_probe.py" \
  "300 r03k_head python scripts/head_probe.py" \
  "300 r03k_bench python bench.py --no-cpu" \
  "300 r03k_ellipsoid python bench.py --mask ellipsoid --no-cpu" \
  "300 r03k_waam python scripts/waam_probe.py"
