#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exec scripts/gpu_seq.sh \
  "300 r03d_brick scripts/_build/brick_probe" \
  "300 r03d_mask_intree python scripts/mask_probe.py" \
  "300 r03d_mask_d8 ADI_HIP_LIB=$R/scripts/_build/libadi_d8.so python scripts/mask_probe.py" \
  "300 r03d_mask_d6 ADI_HIP_LIB=$R/scripts/_build/libadi_d6.so python scripts/mask_probe.py" \
  "300 r03d_pcie python scripts/pcie_probe.py" \
  "600 r03d_pytest python -m pytest tests/test_hip_parity.py -m gpu -q -x -k 'mask or packs or stale or fresh'"
