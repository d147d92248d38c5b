#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "300 r03o_strong python bench.py --rehearse-world 8 --scaling strong --no-cpu" \
  "300 r03o_strong_wg128 ADI_HIP_LIB=$R/scripts/_build/libadi_wg128.so python bench.py --rehearse-world 8 --scaling strong --no-cpu" \
  "600 r03o_pytest_wg128 ADI_HIP_LIB=$R/scripts/_build/libadi_wg128.so python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_dist_slab_gpu.py -m gpu -q -x"
