#!/bin/bash
# Run GPU steps one after another on the gpurun box: each under its own timeout; an ordinary failure (non-zero exit)
# is logged and the next step runs, a timeout / kill (124, 137) stops the sequence -- no GPU step is started after one
# that had to be killed.   usage: scripts/gpu_seq.sh "<seconds> <logname> <command...>" ...
mkdir -p gpurun_out
rc_all=0
for spec in "$@"; do
    secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
    echo "=== [$name] $cmd" | tee -a gpurun_out/seq.log
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] exit $rc" | tee -a gpurun_out/seq.log
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== stopping: $name was killed at its limit" | tee -a gpurun_out/seq.log; exit $rc; fi
    [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
