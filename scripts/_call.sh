cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
$S 600 gpurun_out/r2c_tests.log python -m pytest tests/test_gpu_model.py -x -q -k "wgrad or conv_forward or prologue" &&
$S 400 gpurun_out/r2c_shapes_dma.log python scripts/bench_conv_shapes.py --reps 5 &&
$S 400 gpurun_out/r2c_shapes_old.log env UEM_WGRAD_DMA=0 python scripts/bench_conv_shapes.py --reps 5
tail -5 gpurun_out/r2c_tests.log; paste <(cut -c1-34,90- gpurun_out/r2c_shapes_dma.log) <(cut -c90- gpurun_out/r2c_shapes_old.log)
