#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03e_pytest python -m pytest tests -m gpu -q" \
  "300 r03e_bench python bench.py --no-cpu" \
  "300 r03e_bench_dl16 ADI_HIP_LIB=$R/scripts/_build/libadi_dl16.so python bench.py --no-cpu" \
  "300 r03e_ellipsoid python bench.py --mask ellipsoid --no-cpu" \
  "300 r03e_head python scripts/head_probe.py"
