#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "900 r03i_pytest python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_dist_slab_gpu.py tests/test_waam_harness.py -m gpu -q" \
  "300 r03i_mask python scripts/mask_probe.py" \
  "300 r03i_head python scripts/head_probe.py" \
  "300 r03i_map python scripts/perf_map_small.py" \
  "300 r03i_map_m16 ADI_HIP_LIB=$R/scripts/_build/libadi_m16.so python scripts/perf_map_small.py"
