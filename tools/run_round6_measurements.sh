# The round-6 measurement pass, run on the GPU box from the repo root (gpurun): GPU tests, the bench lines committed under
# profiles/r06_bench_*.json, the rocprofv3 passes (kernel trace + the counter groups), the round's new legs.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final
rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s -rs 2>&1 | grep -E "FULLSIZE|C5 |C4 8 x|C drivers|analytic vs MC|8 ranks|passed|failed|FAILED|SKIPPED|Error" > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
# the counter passes first: bench.py quotes their instruction counters only when profiles/r06_pmc.json names the library it loaded
bash tools/collect_profiles.sh
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r06_pmc.json
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2.json; echo
timeout 900 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 200 $O/bench_c4.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 200 $O/bench_2rank_gloo.json; echo
# the 8-rank rehearsal at C4's REAL per-rank size (2^21 paths per rank, 2^24 in all; the ranks share this GPU)
SVMC_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 10 --warmup 2 --cpu-sample-paths 65536 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err; tail -c 200 $O/bench_8rank_gloo.json; echo
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cat $O/configs.jsonl | cut -c1-200
timeout 600 python tools/r05/bench_frozen.py 100000 300 > $O/frozen.jsonl 2> $O/frozen.err; cat $O/frozen.jsonl
bash tools/r06/frozen_trace.sh; cp gpurun_out/r06_frozen_trace.txt $O/frozen_trace.txt
timeout 900 python tools/r06/mid_waves_sweep.py 100 --variants -1,0,1 --product 2>/dev/null | grep '^{' > $O/mid_waves_sweep.json; cat $O/mid_waves_sweep.json | cut -c1-400
timeout 600 python tools/r06/chain_call_breakdown.py 2>/dev/null | grep '^{' > $O/chain_call_breakdown.json; cat $O/chain_call_breakdown.json | cut -c1-300
timeout 300 python tools/r06/reducers_bw.py 10 2>/dev/null | grep '^{' > $O/reducers_bw.jsonl; cat $O/reducers_bw.jsonl
timeout 300 python tools/ubench/ab_kernels.py stochvolmodels_amd/libsvmc.so final 2>/dev/null | tail -1 > $O/single_kernels.jsonl; cat $O/single_kernels.jsonl | cut -c1-400
timeout 900 python tools/bench_calibration_mc.py 2>/dev/null | grep '^{' | tail -1 > $O/calibration_mc.json; cat $O/calibration_mc.json | cut -c1-600
