#!/bin/bash
# The evidence kept under profiles/ for one build (run on the GPU box, from the repo root):
#   scripts/profile_round.sh r02_a
# rocprofv3 --kernel-trace --stats of `bench.py` and `bench.py --config cyl`, the JSON lines they printed under the
# profiler, and the FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only) -> gpurun_out/<tag>_*; the summaries are
# then copied into profiles/ by hand (gpurun_out/ is scratch).
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cart_prof -- python3 $R/bench.py --no-cpu > $O/${TAG}_cart_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cyl_prof -- python3 $R/bench.py --config cyl --no-cpu > $O/${TAG}_cyl_line.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cart_fetch -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cart_write -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cyl_fetch -- python3 $R/bench.py --config cyl --no-cpu --steps 3 --warmup 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cyl_write -- python3 $R/bench.py --config cyl --no-cpu --steps 3 --warmup 1 > /dev/null
python3 $R/scripts/pmc_summary.py $O/${TAG}_cart_fetch $O/${TAG}_cart_write $O/${TAG}_cyl_fetch $O/${TAG}_cyl_write $O/${TAG}_pmc_traffic.json
cp $(ls $O/${TAG}_cart_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_cart_kernel_stats.csv
cp $(ls $O/${TAG}_cyl_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_cyl_kernel_stats.csv
echo profiled $TAG
