#!/bin/bash
# The evidence kept under profiles/ for one build (run on the GPU box, from the repo root):
#   scripts/profile_round.sh r03_z
# rocprofv3 --kernel-trace --stats of `bench.py --no-also` (the timed region and the 42 B/cell sweeps only: the `also` object would
# mix launches of the same kernels on other grids into the averages) and `bench.py --config cyl`, the JSON lines they printed under the
# profiler, the FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only), the plain bench line with the CPU baseline and
# parity, the ellipsoid line, and the rehearsals of a middle rank of 8 (weak / strong, loopback / RCCL self-loop) with a kernel
# trace of the weak one -> gpurun_out/<tag>_*; the summaries are then copied into profiles/ (gpurun_out/ is scratch).
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cart_prof -- python3 $R/bench.py --no-cpu --no-also > $O/${TAG}_cart_line.json
echo "cart trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cyl_prof -- python3 $R/bench.py --config cyl --no-cpu > $O/${TAG}_cyl_line.json
echo "cyl trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cart_fetch -- python3 $R/bench.py --no-cpu --no-also --steps 3 --warmup 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cart_write -- python3 $R/bench.py --no-cpu --no-also --steps 3 --warmup 1 > /dev/null
echo "cart pmc done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cyl_fetch -- python3 $R/bench.py --config cyl --no-cpu --steps 3 --warmup 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cyl_write -- python3 $R/bench.py --config cyl --no-cpu --steps 3 --warmup 1 > /dev/null
echo "cyl pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_reh8_prof -- python3 $R/bench.py --rehearse-world 8 --no-cpu > $O/${TAG}_reh8_weak_prof_line.json
echo "rehearsal trace done"
cd $R
python3 scripts/pmc_summary.py $O/${TAG}_cart_fetch $O/${TAG}_cart_write $O/${TAG}_cyl_fetch $O/${TAG}_cyl_write $O/${TAG}_pmc_traffic.json > /dev/null
cp $(ls $O/${TAG}_cart_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_cart_kernel_stats.csv
cp $(ls $O/${TAG}_cyl_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_cyl_kernel_stats.csv
cp $(ls $O/${TAG}_reh8_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_reh8_weak_kernel_stats.csv
python3 bench.py --rehearse-world 8 --no-cpu > $O/${TAG}_reh8_weak_line.json
python3 bench.py --rehearse-world 8 --no-cpu --force-dist 2> /dev/null > $O/${TAG}_reh8_weak_rccl_line.json
python3 bench.py --rehearse-world 8 --scaling strong --no-cpu > $O/${TAG}_reh8_strong_line.json
python3 bench.py --rehearse-world 8 --scaling strong --no-cpu --force-dist 2> /dev/null > $O/${TAG}_reh8_strong_rccl_line.json
python3 bench.py --mask ellipsoid --no-cpu --no-also > $O/${TAG}_ellipsoid_line.json
echo "rehearsal lines done"
# two REAL ranks on this one GPU over the gloo-staged test transport: the N > 1 line with parity_vs_one_domain (not a 2-GPU measurement)
python3 bench.py --gpus 2 --n 256 --no-cpu --transport gloo-staged > $O/${TAG}_staged2_weak_line.json 2> /dev/null
python3 bench.py --gpus 2 --n 256 --no-cpu --transport gloo-staged --scaling strong > $O/${TAG}_staged2_strong_line.json 2> /dev/null
echo "staged two-rank lines done"
python3 scripts/deferred_lines_probe.py 512x512x512 cfl=1 cfl=10 cfl=40 cfl=200 > $O/${TAG}_deferred_lines_probe.txt 2> /dev/null
python3 scripts/waam_slab_probe.py > $O/${TAG}_waam_slab_probe.txt 2> /dev/null
python3 scripts/perf_map.py 256x256x256 320x256x256 384x256x256 448x256x256 512x256x256 512x128x256 512x512x256 512x512x512 > $O/${TAG}_perf_map_elongated.txt 2> /dev/null
python3 scripts/slab_pad_probe.py 256x500x500 > $O/${TAG}_slab_pad_probe.txt 2> /dev/null
echo "probes done"
python3 bench.py > $O/${TAG}_bench_line.json
echo profiled $TAG
