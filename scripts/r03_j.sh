#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03j_pytest python -m pytest tests -m gpu -q" \
  "300 r03j_mask python scripts/mask_probe.py" \
  "300 r03j_mask_nocoef ADI_HIP_LIB=$R/scripts/_build/libadi_nocoef.so python scripts/mask_probe.py" \
  "300 r03j_bench python bench.py --no-cpu" \
  "300 r03j_256 python bench.py --n 256 --no-cpu"
