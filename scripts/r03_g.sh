#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "600 r03g_pytest_cyl python -m pytest tests -m gpu -q -k 'cyl or config4 or spiral or analytic'" \
  "300 r03g_cyl_probe python scripts/cyl_probe.py" \
  "300 r03g_cyl_bench python bench.py --config cyl --no-cpu" \
  "300 r03g_cart_inplace ADI_HIP_LIB=$R/scripts/_build/libadi_cartip.so python scripts/inplace_probe.py"
