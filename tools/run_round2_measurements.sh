# The round-2 measurement pass, run on the GPU box from the repo root (gpurun): GPU tests, the bench lines committed
# under profiles/r02_bench_*.json, secondary configs, the calibration bench and the rocprofv3 passes.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -rs --durations=5 2>&1 | tail -15 > gpurun_out/pytest_gpu_r2_final.log; tail -15 gpurun_out/pytest_gpu_r2_final.log
timeout 900 python bench.py > gpurun_out/bench_r2_final.json 2> gpurun_out/bench_r2_final.err; tail -c 600 gpurun_out/bench_r2_final.json; echo
timeout 900 python bench.py --config c4 > gpurun_out/bench_r2_final_c4.json 2> gpurun_out/bench_r2_final_c4.err; tail -c 300 gpurun_out/bench_r2_final_c4.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_r2_final_2rank_gloo.json 2> gpurun_out/bench_r2_final_2rank_gloo.err; tail -c 300 gpurun_out/bench_r2_final_2rank_gloo.json; echo
SVMC_DIST_COMM=rccl SVMC_DIST_SINGLE_RANK_GROUP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 900 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r2_final_c4_rcclcomm.json 2> gpurun_out/bench_r2_final_c4_rcclcomm.err; tail -c 300 gpurun_out/bench_r2_final_c4_rcclcomm.json; echo
timeout 900 python tools/bench_configs.py > gpurun_out/configs_r2.jsonl 2> gpurun_out/configs_r2.err; cat gpurun_out/configs_r2.jsonl
timeout 600 python tools/bench_calibration.py 100000 > gpurun_out/calib_r2.log 2>&1; tail -8 gpurun_out/calib_r2.log
bash tools/collect_profiles.sh
# single kernels of the final build on this box, the counter passes of the C3 kernels and of the payoff pass, the rough kernels
timeout 300 python tools/ubench/ab_kernels.py stochvolmodels_amd/libsvmc.so final 2>/dev/null | tail -1 > gpurun_out/ab_final.jsonl; cat gpurun_out/ab_final.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/c3pmc -o c3 -- python $GRAFT_REPO_ROOT/tools/ubench/c3_probe.py > $GRAFT_REPO_ROOT/gpurun_out/c3pmc.log 2>&1; echo c3pmc rc=$?)
bash tools/ubench/payoff_pmc.sh 2>&1 | tail -3
timeout 600 python tools/bench_rough.py > gpurun_out/rough_r2.jsonl 2> gpurun_out/rough_r2.err; tail -5 gpurun_out/rough_r2.jsonl
