#!/bin/bash
# Two data-parallel ranks of bench.py sharing device 0 (backend gloo), rank 0 under rocprofv3 --kernel-trace.
#   scripts/dp_rehearsal.sh <tag> [bench.py args ...]
# Output: gpurun_out/dp_<tag>/{rank0.log,rank1.log,prof/...}; summarise with scripts/dp_trace_report.py.
set -e
TAG=$1; shift 1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/dp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + RANDOM % 20000)) WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$ROOT"
RANK=1 LOCAL_RANK=1 python3 bench.py --gpus 2 --backend gloo --device 0 --no-cpu-baseline --no-other-precisions --no-kernel-events "$@" > "$OUT/rank1.log" 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o r0 -- python3 bench.py --gpus 2 --backend gloo --device 0 --no-cpu-baseline --no-other-precisions --no-kernel-events "$@" > "$OUT/rank0.log" 2>&1
wait $P1
grep -h '^{' "$OUT/rank0.log" | tail -1
