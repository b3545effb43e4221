# The round-4 measurement pass, run on the GPU box from the repo root (gpurun): GPU tests, the bench lines committed under
# profiles/r04_bench_*.json, the rocprofv3 passes (kernel trace + the counter groups), secondary configs.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final
rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s -rs 2>&1 | grep -E "FULLSIZE|C5 |C drivers|analytic vs MC|8 ranks|passed|failed|FAILED|SKIPPED|Error" > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
# the counter passes first: bench.py quotes their instruction counters only when profiles/r04_pmc.json names the library it loaded
bash tools/collect_profiles.sh
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r04_pmc.json
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 200 $O/bench_c2.json; echo
timeout 900 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 200 $O/bench_c4.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 200 $O/bench_2rank_gloo.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --paths-per-gpu 262144 --steps 10 --warmup 2 --cpu-sample-paths 65536 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err; tail -c 200 $O/bench_8rank_gloo.json; echo
SVMC_DIST_SINGLE_RANK_GROUP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 timeout 900 python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_torch_rccl_single.json 2> $O/bench_c4_torch_rccl_single.err; tail -c 200 $O/bench_c4_torch_rccl_single.json; echo
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cat $O/configs.jsonl
timeout 600 python tools/bench_calibration.py 100000 2>/dev/null | head -1 > $O/calib.jsonl; cat $O/calib.jsonl
timeout 300 python tools/ubench/ab_kernels.py stochvolmodels_amd/libsvmc.so final 2>/dev/null | tail -1 > $O/single_kernels.jsonl; cat $O/single_kernels.jsonl
timeout 300 python tools/ubench/fused_driver_overhead.py 200 2>/dev/null | tail -1 > $O/fused_driver.json; cat $O/fused_driver.json
