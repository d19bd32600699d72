cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=scripts/gpu_step.sh
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-precisions"
$S 600 gpurun_out/r2u_tests.log python -m pytest tests/test_gpu_model.py -q -x &&
$S 400 gpurun_out/r2u_shapes.log python scripts/bench_conv_shapes.py --reps 5 &&
$S 600 gpurun_out/r2u_bench.log python bench.py &&
$S 400 gpurun_out/r2u_prof.log rocprofv3 --kernel-trace --stats -d gpurun_out/r2u_prof -o run -- python3 $B &&
$S 400 gpurun_out/r2u_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2u_fetch -- python3 $B --no-kernel-events &&
$S 400 gpurun_out/r2u_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2u_write -- python3 $B --no-kernel-events &&
$S 400 gpurun_out/r2u_sq.log rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/r2u_sq -- python3 $B --no-kernel-events &&
$S 400 gpurun_out/r2u_grbm.log rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/r2u_grbm -- python3 $B --no-kernel-events
python scripts/kernel_stats.py gpurun_out/r2u_prof/run_results.db > gpurun_out/r2u_kernel_stats.csv
python scripts/pmc_traffic.py gpurun_out/r2u_fetch gpurun_out/r2u_write "resnet50-aspp ssl B=32 size=512 prec=fp32" > gpurun_out/r2u_pmc_traffic.txt 2>&1
cp profiles/traffic_latest.json gpurun_out/r2u_traffic_latest.json
python scripts/pmc_summary.py gpurun_out/r2u_sq _kernel > gpurun_out/r2u_pmc_sq.txt 2>&1
python scripts/pmc_summary.py gpurun_out/r2u_grbm _kernel > gpurun_out/r2u_pmc_grbm.txt 2>&1
rm -rf gpurun_out/r2u_prof gpurun_out/r2u_fetch gpurun_out/r2u_write gpurun_out/r2u_sq gpurun_out/r2u_grbm
tail -2 gpurun_out/r2u_tests.log; grep totals gpurun_out/r2u_shapes.log; cat gpurun_out/r2u_pmc_traffic.txt; grep -h '^{' gpurun_out/r2u_bench.log | cut -c1-1500
