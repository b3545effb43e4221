# The round-3 measurement pass, run on the GPU box from the repo root (gpurun): GPU tests, the bench lines committed
# under profiles/r03_bench_*.json, secondary configs, the calibration bench, the rocprofv3 passes and the single-kernel table.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -rs 2>&1 | grep -E "FULLSIZE|C5 |C drivers|analytic vs MC|passed|failed|FAILED|SKIPPED" > gpurun_out/pytest_gpu_r3_final.log; tail -4 gpurun_out/pytest_gpu_r3_final.log
# the counter passes first: bench.py quotes their instruction counts only when profiles/r03_pmc.json names the library it loaded
bash tools/collect_profiles.sh
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r03_pmc.json
timeout 900 python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 300 gpurun_out/bench_r3_final.json; echo
timeout 900 python bench.py --config c4 > gpurun_out/bench_r3_final_c4.json 2> gpurun_out/bench_r3_final_c4.err; tail -c 300 gpurun_out/bench_r3_final_c4.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_r3_final_2rank_gloo.json 2> gpurun_out/bench_r3_final_2rank_gloo.err; tail -c 300 gpurun_out/bench_r3_final_2rank_gloo.json; echo
SVMC_DIST_COMM=rccl SVMC_DIST_SINGLE_RANK_GROUP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 900 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r3_final_c4_rcclcomm.json 2> gpurun_out/bench_r3_final_c4_rcclcomm.err; tail -c 300 gpurun_out/bench_r3_final_c4_rcclcomm.json; echo
SVMC_DIST_SINGLE_RANK_GROUP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 timeout 900 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r3_final_c4_torch_rccl_single.json 2> gpurun_out/bench_r3_final_c4_torch_rccl_single.err; tail -c 300 gpurun_out/bench_r3_final_c4_torch_rccl_single.json; echo
timeout 900 python bench.py --steps 400 --no-cpu-baseline --no-extra-legs --no-streamed > gpurun_out/bench_r3_400steps.json 2>/dev/null
timeout 900 python tools/bench_configs.py > gpurun_out/configs_r3.jsonl 2> gpurun_out/configs_r3.err; cat gpurun_out/configs_r3.jsonl
timeout 600 python tools/bench_calibration.py 100000 > gpurun_out/calib_r3.log 2>&1; tail -4 gpurun_out/calib_r3.log
timeout 300 python tools/ubench/ab_kernels.py stochvolmodels_amd/libsvmc.so final 2>/dev/null | tail -1 > gpurun_out/ab_final_r3.jsonl; cat gpurun_out/ab_final_r3.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/c3pmc -o c3 -- python $GRAFT_REPO_ROOT/tools/ubench/c3_probe.py > $GRAFT_REPO_ROOT/gpurun_out/c3pmc.log 2>&1; echo c3pmc rc=$?)
timeout 600 python tools/bench_rough.py > gpurun_out/rough_r3.jsonl 2> gpurun_out/rough_r3.err; tail -3 gpurun_out/rough_r3.jsonl
timeout 300 python tools/ubench/sync_latency.py > gpurun_out/sync_latency_r3.json 2>/dev/null; cat gpurun_out/sync_latency_r3.json
timeout 300 python tools/ubench/fused_driver_overhead.py 2>/dev/null | tail -1 > gpurun_out/fused_driver_r3.json; cat gpurun_out/fused_driver_r3.json
