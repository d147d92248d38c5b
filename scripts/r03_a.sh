#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite, the default bench line (with parity_rel_linf), the rehearsals of a
# middle rank of 8 (weak and strong, loopback and RCCL self-loop), the ellipsoid line, and a kernel trace of the weak rehearsal
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03a_pytest.log 2>&1 || { tail -30 $O/r03a_pytest.log; exit 1; }
tail -3 $O/r03a_pytest.log
timeout -k 10 600 python bench.py > $O/r03a_bench.json 2> $O/r03a_bench.err
timeout -k 10 300 python bench.py --rehearse-world 8 --no-cpu > $O/r03a_reh8_weak.json 2> $O/r03a_reh8_weak.err
timeout -k 10 300 python bench.py --rehearse-world 8 --no-cpu --force-dist > $O/r03a_reh8_weak_rccl.json 2> $O/r03a_reh8_weak_rccl.err
timeout -k 10 300 python bench.py --rehearse-world 8 --scaling strong --no-cpu > $O/r03a_reh8_strong.json 2> $O/r03a_reh8_strong.err
timeout -k 10 300 python bench.py --rehearse-world 8 --scaling strong --no-cpu --force-dist > $O/r03a_reh8_strong_rccl.json 2> $O/r03a_reh8_strong_rccl.err
timeout -k 10 300 python bench.py --mask ellipsoid --no-cpu > $O/r03a_ellipsoid.json 2> $O/r03a_ellipsoid.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03a_reh8_prof -- python3 $R/bench.py --rehearse-world 8 --no-cpu > $O/r03a_reh8_prof_line.json 2> $O/r03a_reh8_prof.err
cp $(ls $O/r03a_reh8_prof/*/*kernel_stats.csv | head -1) $O/r03a_reh8_kernel_stats.csv
echo r03a done
