#!/bin/bash
# run one GPU step under a timeout; exit 0 unless it had to be killed (then stop the whole gpurun call)
#   scripts/gpu_step.sh <seconds> <logfile> <command ...>
T=$1; LOG=$2; shift 2
timeout -k 10 "$T" "$@" > "$LOG" 2>&1
rc=$?
echo "[gpu_step] rc=$rc  $*" | tee -a "$LOG"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
exit 0
