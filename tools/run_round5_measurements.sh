# The round-5 measurement pass, run on the GPU box from the repo root (gpurun): GPU tests, the bench lines committed under
# profiles/r05_bench_*.json, the rocprofv3 passes (kernel trace + the counter groups), the round's new legs.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final
rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s -rs 2>&1 | grep -E "FULLSIZE|C5 |C4 8 x|C drivers|analytic vs MC|8 ranks|passed|failed|FAILED|SKIPPED|Error" > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
# the counter passes first: bench.py quotes their instruction counters only when profiles/r05_pmc.json names the library it loaded
bash tools/collect_profiles.sh
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r05_pmc.json
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2.json; echo
timeout 900 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 200 $O/bench_c4.json; echo
SVMC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 200 $O/bench_2rank_gloo.json; echo
# the 8-rank rehearsal at C4's REAL per-rank size (2^21 paths per rank, 2^24 in all; the ranks share this GPU)
SVMC_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 10 --warmup 2 --cpu-sample-paths 65536 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err; tail -c 200 $O/bench_8rank_gloo.json; echo
# the whole fallback ladder, failing where this box makes it fail (two ranks on one device: the nccl and rccl probes fail on their own)
SVMC_BENCH_SHARE_DEVICES=1 SVMC_DIST_FORCE_PROBE=1 SVMC_BENCH_INIT_TIMEOUT=90 timeout 1500 python bench.py --gpus 2 --steps 10 --warmup 2 --cpu-sample-paths 65536 > $O/bench_2rank_ladder.json 2> $O/bench_2rank_ladder.err; tail -c 200 $O/bench_2rank_ladder.json; echo
# one process, eight shards of 2^21 paths (all on this device, host transport): the single-process route at C4's real size
timeout 900 python bench.py --gpus 8 --single-process --devices 0,0,0,0,0,0,0,0 --steps 10 --warmup 2 > $O/bench_single_process_8shards.json 2> $O/bench_single_process.err; tail -c 300 $O/bench_single_process_8shards.json; echo
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cat $O/configs.jsonl | cut -c1-200
timeout 600 python tools/r05/bench_frozen.py 100000 300 > $O/frozen.jsonl 2> $O/frozen.err; cat $O/frozen.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/frozen_prof -o fr -- python $GRAFT_REPO_ROOT/tools/r05/bench_frozen.py 100000 100 > $GRAFT_REPO_ROOT/$O/frozen_prof.log 2>&1)
find $O/frozen_prof -type f ! -name '*.db' -delete 2>/dev/null
timeout 300 python tools/r05/frozen_breakdown.py 100000 300 2>/dev/null | grep '^{' > $O/frozen_breakdown.jsonl; cat $O/frozen_breakdown.jsonl
timeout 300 python tools/r05/frozen_scaling.py 100 2>/dev/null | grep '^{' > $O/frozen_scaling.jsonl; cat $O/frozen_scaling.jsonl
timeout 600 python tools/r05/few_waves_sweep.py 100 2>/dev/null | grep '^{' > $O/few_waves_sweep.json; cat $O/few_waves_sweep.json | cut -c1-400
timeout 600 python tools/r05/bulk_outputs.py > $O/bulk_outputs.jsonl 2> $O/bulk.err; cat $O/bulk_outputs.jsonl | cut -c1-300
timeout 300 python tools/r05/moments_timing.py 2>/dev/null | grep '^{' > $O/moments_timing.jsonl; cat $O/moments_timing.jsonl
timeout 600 python tools/ubench/ab_vol_paths.py stochvolmodels_amd/libsvmc.so round5 2>$O/vol_paths.err | tail -1 > $O/vol_paths.json; cat $O/vol_paths.json | cut -c1-600
timeout 300 python tools/r04/power_probe.py 2>/dev/null | grep '^{' > $O/power.jsonl; cat $O/power.jsonl
timeout 300 python tools/ubench/ab_kernels.py stochvolmodels_amd/libsvmc.so final 2>/dev/null | tail -1 > $O/single_kernels.jsonl; cat $O/single_kernels.jsonl | cut -c1-400
timeout 900 python tools/bench_calibration_mc.py 2>/dev/null | grep '^{' | tail -1 > $O/calibration_mc.json; cat $O/calibration_mc.json | cut -c1-600
