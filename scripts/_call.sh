cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
$S 900 gpurun_out/r2t_tests.log python -m pytest tests -m gpu -q -s
grep -v amdgpu gpurun_out/r2t_tests.log | grep -E "update error|operands, R101|passed|failed|^E  |Error" | cut -c1-400
