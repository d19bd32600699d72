#!/bin/bash
# Everything `profiles/` needs for one build of the kernels, on the GPU box:
#   scripts/measure_round.sh <tag> [extra bench.py arguments, e.g. --storage bf16]
#       ->  gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_traffic.txt, pmc_hbm_by_kernel.txt, pmc_mfma_busy.txt, ...}
# (1) bench.py  (2) rocprofv3 --kernel-trace --stats of the same step  (3) two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, as
# MI355X_MICROARCH.md prescribes) -> HBM bytes per conv family (profiles/traffic_latest.json, headline configuration only) and per
# kernel  (4) SQ / GRBM counters of the conv kernels.
# With extra arguments the run is one of the OTHER configurations (bf16 storage, PPM head, R101 1024^2): the plain bench.py line of
# step (1) is skipped (the headline run's `other_configs` holds it) and traffic_latest.json is left alone.
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
set -u
TAG=${1:-measure}
shift || true
EXTRA="$*"
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS=3
ARGS="--steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs --no-hipgraph $EXTRA"
if [ -z "$EXTRA" ]; then
  echo "[measure] bench"; timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
fi
# Kernel trace with the weight gradients on the MAIN stream (UEM_SIDE_WGRAD=0 / UEM_BF16_SIDE_WGRAD=0): one kernel at a time, so the
# per-kernel durations add up to the step and agree with the bench line's roofline (whose one profiled step is serial too).  A second
# trace with the shipped default (weight gradients on a side stream) follows: its durations OVERLAP -- their sum exceeds the step.
# (round 6: the step's two graphs run on two streams by default -- UEM_TWO_STREAM_FWD=0 keeps this trace and the counter passes serial)
export UEM_SIDE_WGRAD=0 UEM_BF16_SIDE_WGRAD=0 UEM_TWO_STREAM_FWD=0
echo "[measure] kernel trace (serial)"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o step -- python3 bench.py $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" || exit 1
python3 scripts/kernel_stats.py "$(find "$OUT/trace" -name '*.db' | head -1)" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
unset UEM_SIDE_WGRAD UEM_BF16_SIDE_WGRAD UEM_TWO_STREAM_FWD
echo "[measure] kernel trace (side stream + two graph streams, overlapped)"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o step -- python3 bench.py $ARGS > "$OUT/bench_under_rocprof_overlapped.json" 2> "$OUT/trace2.err" || exit 1
python3 scripts/kernel_stats.py "$(find "$OUT/trace" -name '*.db' | head -1)" > "$OUT/kernel_stats_overlapped.csv"
export UEM_TWO_STREAM_FWD=0
echo "[measure] pmc FETCH_SIZE"; timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS --no-kernel-events > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 1
echo "[measure] pmc WRITE_SIZE"; timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS --no-kernel-events > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 1
if [ -z "$EXTRA" ]; then
  python3 scripts/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" $((STEPS + 2)) > "$OUT/pmc_traffic.txt" 2>&1
  cp profiles/traffic_latest.json "$OUT/traffic_latest.json"
fi
python3 scripts/pmc_hbm_by_kernel.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/kernel_stats.csv" 40 > "$OUT/pmc_hbm_by_kernel.txt" 2>&1
echo "[measure] pmc SQ"; timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS --no-kernel-events > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err" || exit 1
python3 scripts/pmc_summary.py "$OUT/pmc_sq" > "$OUT/pmc_mfma_busy.txt" 2>&1
if [ -z "$EXTRA" ]; then
  echo "[measure] pmc GRBM"; timeout -k 10 900 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/pmc_grbm" -- python3 bench.py $ARGS --no-kernel-events > "$OUT/pmc_grbm.json" 2> "$OUT/pmc_grbm.err" || exit 1
  python3 scripts/pmc_summary.py "$OUT/pmc_grbm" >> "$OUT/pmc_mfma_busy.txt" 2>&1
fi
if [ -z "$EXTRA" ]; then
  # HBM bytes per conv launch BY SHAPE: the same two counters over ONE timed step whose conv ops are preceded by marker launches
  echo "[measure] pmc per op"
  export UEM_PROF_MARK=1
  PARGS="--steps 1 --warmup 2 --no-cpu-baseline --no-other-configs --no-hipgraph"
  timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_op_fetch" -- python3 bench.py $PARGS --dump-conv-events "$OUT/ev_fetch.json" > "$OUT/pmc_op_fetch.json" 2> "$OUT/pmc_op_fetch.err" || exit 1
  timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_op_write" -- python3 bench.py $PARGS --dump-conv-events "$OUT/ev_write.json" > "$OUT/pmc_op_write.json" 2> "$OUT/pmc_op_write.err" || exit 1
  unset UEM_PROF_MARK
  python3 scripts/pmc_per_op.py "$OUT/pmc_op_fetch" "$OUT/pmc_op_write" "$OUT/ev_fetch.json" > "$OUT/pmc_traffic_per_shape.txt" 2>&1
  rm -rf "$OUT/pmc_op_fetch" "$OUT/pmc_op_write"
fi
# the raw counter dumps are large: keep the summaries only
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_grbm" "$OUT/trace"
echo "[measure] done"
