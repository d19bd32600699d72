cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
$S 900 gpurun_out/r2w_tests.log python -m pytest tests -m gpu -q -x &&
$S 400 gpurun_out/r2w_bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions
tail -2 gpurun_out/r2w_tests.log; grep -h '^{' gpurun_out/r2w_bench.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['families']), d['steps_ms'])"
