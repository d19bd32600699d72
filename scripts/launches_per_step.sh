#!/bin/bash
# gpurun_out/<tag>/launches_per_step.txt: steady-state launches per step of the headline bench (see launches_per_step.py)
set -u
TAG=${1:-launches}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--warmup 1 --no-cpu-baseline --no-other-precisions --no-other-configs --no-hipgraph --no-kernel-events"
for n in 1 6; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace$n" -o step -- python3 bench.py --steps $n $ARGS > "$OUT/bench$n.json" 2> "$OUT/trace$n.err" || exit 1
  python3 scripts/kernel_stats.py "$(find "$OUT/trace$n" -name '*.db' | head -1)" > "$OUT/kernel_stats_$n.csv"
  rm -rf "$OUT/trace$n"
done
python3 scripts/launches_per_step.py "$OUT/kernel_stats_1.csv" "$OUT/kernel_stats_6.csv" 5 > "$OUT/launches_per_step.txt"
