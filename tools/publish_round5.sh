#!/bin/bash
# After `gpurun -- bash tools/run_round5_measurements.sh`: turn what came back under gpurun_out/ into the committed
# evidence under profiles/ (run here, in the build container, from the repo root).
set -e
R=r05
O=gpurun_out/r05_final
for CFG in c2 c4; do
  D=gpurun_out/prof_$CFG
  { echo "# round 5, build sha256 $(sha256sum stochvolmodels_amd/libsvmc.so | cut -c1-16): rocprofv3 --kernel-trace --stats of \`python bench.py --config $CFG --no-cpu-baseline --no-extra-legs\` (K = 50, W = 10 after 10 pre-warm calls); then the PMC passes of the same command at --steps 3 --warmup 1 (SQ group; FETCH_SIZE; WRITE_SIZE: separate runs, --kernel-trace only)"
    python tools/rocpd_summary.py $(find $D/kt -name '*.db') $(find $D/sq -name '*.db') $(find $D/fetch -name '*.db') $(find $D/write -name '*.db'); } > profiles/${R}_${CFG}_rocprofv3_summary.txt
done
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/${R}_pmc.json
for pair in bench_c2:bench_c2 bench_c4:bench_c4_one_gpu bench_2rank_gloo:bench_c4_2ranks_one_gpu_gloo bench_8rank_gloo:bench_c4_8ranks_one_gpu_gloo_full_size bench_2rank_ladder:bench_c4_2ranks_ladder_lands_on_gloo bench_single_process_8shards:bench_c4_single_process_8_shards_one_gpu; do
  src=${pair%%:*}; dst=${pair#*:}
  grep '^{"metric' $O/$src.json | tail -1 > profiles/${R}_$dst.json
done
cp $O/configs.jsonl profiles/${R}_configs.jsonl
cp $O/frozen.jsonl profiles/${R}_frozen_objective.jsonl
{ echo "# rocprofv3 --kernel-trace --stats of python tools/r05/bench_frozen.py 100000 100 (4 x 13 chain, 10^5 paths x 364 steps): the frozen route's kernels (logsv_chain_rng_sets_kernel<P>) and the round-4 route's (logsv_chain_w_*: 582 MB of resident randoms streamed per evaluation) side by side"; python tools/rocpd_summary.py $(find $O/frozen_prof -name '*.db') | cut -c1-160; } > profiles/${R}_calibration_objective.txt
cp $O/frozen_breakdown.jsonl profiles/${R}_frozen_breakdown.jsonl
cp $O/frozen_scaling.jsonl profiles/${R}_frozen_scaling.jsonl
cp $O/few_waves_sweep.json profiles/${R}_few_waves_sweep.json
cp $O/bulk_outputs.jsonl profiles/${R}_bulk_outputs.jsonl
cp $O/moments_timing.jsonl profiles/${R}_moments_timing.jsonl
cp $O/vol_paths.json profiles/${R}_vol_paths.json
cp $O/power.jsonl profiles/${R}_power.jsonl
cp $O/single_kernels.jsonl profiles/${R}_single_kernels.jsonl
cp $O/calibration_mc.json profiles/${R}_calibration_mc.json
{ echo "# python -m pytest tests -m gpu -q -s on one MI355X, round-5 build: what the full-size same-stream parity tests (rank 0's, rank 3's and rank 7's C4 share; C5 at 2^20 and 2^23), the verdict-parity tests, the sharded C-driver test, the 8-shard multi-session and the 8-rank bench rehearsal printed"; cat $O/pytest_gpu.log | sed 's/^[.sF]*//'; } > profiles/${R}_fullsize_parity.txt
python - <<'PY'
import json
for f in ['r05_bench_c2','r05_bench_c4_one_gpu','r05_bench_c4_2ranks_one_gpu_gloo','r05_bench_c4_8ranks_one_gpu_gloo_full_size','r05_bench_c4_2ranks_ladder_lands_on_gloo','r05_bench_c4_single_process_8_shards_one_gpu']:
    try:
        d=json.load(open(f'profiles/{f}.json'))
    except Exception as e:
        print(f, 'MISSING', e); continue
    r=d.get('roofline',{})
    print(f, '%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], 'N', d['n_gpus'], 'frac', round(r.get('frac',0),4), 'clock', r.get('clock_mhz_in_kernel'), 'stale', r.get('stale'), d.get('comm'), (d.get('comm_ladder') or {}).get('rung'), 'selfcheck', d.get('sharded_vs_one_gpu_max_rel_dev'), 'sp', (d.get('single_process_route') or {}).get('value'), d.get('reduce'))
PY
