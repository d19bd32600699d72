cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
scripts/gpu_step.sh 400 gpurun_out/r2q.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions
grep -h '^{' gpurun_out/r2q.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['steps_ms'], d['host_enqueued_at_ms'], d['roofline']['families'])"
