cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
$S 900 gpurun_out/r2i_tests.log python -m pytest tests -m gpu -x -q &&
$S 400 gpurun_out/r2i_bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions &&
$S 400 gpurun_out/r2i_bench_s0.log env UEM_WGRAD_STREAM=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions
tail -3 gpurun_out/r2i_tests.log; grep -h '^{' gpurun_out/r2i_bench.log gpurun_out/r2i_bench_s0.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['wgrad_side_stream'], json.dumps(d['roofline']['families']), json.dumps(d['phases_ms']))"
