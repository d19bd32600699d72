cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
$S 900 gpurun_out/r2e_tests.log python -m pytest tests -m gpu -x -q &&
$S 400 gpurun_out/r2e_shapes.log python scripts/bench_conv_shapes.py --reps 5 &&
$S 400 gpurun_out/r2e_bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
tail -4 gpurun_out/r2e_tests.log; grep totals gpurun_out/r2e_shapes.log; grep -h '^{' gpurun_out/r2e_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']), json.dumps(d['phases_ms']))"
