#!/bin/bash
# After `gpurun -- bash tools/run_round6_measurements.sh`: turn what came back under gpurun_out/ into the committed
# evidence under profiles/ (run here, in the build container, from the repo root).
set -e
R=r06
O=gpurun_out/r06_final
for CFG in c2 c4; do
  D=gpurun_out/prof_$CFG
  { echo "# round 6, build sha256 $(sha256sum stochvolmodels_amd/libsvmc.so | cut -c1-16): rocprofv3 --kernel-trace --stats of \`python bench.py --config $CFG --no-cpu-baseline --no-extra-legs\` (K = 50, W = 10 after 10 pre-warm calls); then the PMC passes of the same command at --steps 3 --warmup 1 (SQ group; FETCH_SIZE; WRITE_SIZE: separate runs, --kernel-trace only)"
    python tools/rocpd_summary.py $(find $D/kt -name '*.db') $(find $D/sq -name '*.db') $(find $D/fetch -name '*.db') $(find $D/write -name '*.db'); } > profiles/${R}_${CFG}_rocprofv3_summary.txt
done
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/${R}_pmc.json
for pair in bench_c2:bench_c2 bench_c4:bench_c4_one_gpu bench_2rank_gloo:bench_c4_2ranks_one_gpu_gloo bench_8rank_gloo:bench_c4_8ranks_one_gpu_gloo_full_size; do
  src=${pair%%:*}; dst=${pair#*:}
  grep '^{"metric' $O/$src.json | tail -1 > profiles/${R}_$dst.json
done
cp $O/configs.jsonl profiles/${R}_configs.jsonl
cp $O/frozen.jsonl profiles/${R}_frozen_objective.jsonl
cp $O/frozen_trace.txt profiles/${R}_frozen_trace.txt
cp $O/mid_waves_sweep.json profiles/${R}_mid_waves_sweep.json
cp $O/chain_call_breakdown.json profiles/${R}_chain_call_breakdown.json
cp $O/reducers_bw.jsonl profiles/${R}_reducers_bw.jsonl
cp $O/single_kernels.jsonl profiles/${R}_single_kernels.jsonl
cp $O/calibration_mc.json profiles/${R}_calibration_mc.json
{ echo "# python -m pytest tests -m gpu -q -s on one MI355X, round-6 build: what the full-size same-stream parity tests (rank 0's, rank 3's and rank 7's C4 share; C5 at 2^20 and 2^23), the verdict-parity tests, the sharded C-driver test, the 8-shard multi-session and the 8-rank bench rehearsal printed"; cat $O/pytest_gpu.log | sed 's/^[.sF]*//'; } > profiles/${R}_fullsize_parity.txt
python - <<'PY'
import json
for f in ['r06_bench_c2','r06_bench_c4_one_gpu','r06_bench_c4_2ranks_one_gpu_gloo','r06_bench_c4_8ranks_one_gpu_gloo_full_size']:
    try:
        d=json.load(open(f'profiles/{f}.json'))
    except Exception as e:
        print(f, 'MISSING', e); continue
    r=d.get('roofline',{})
    print(f, '%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], 'N', d['n_gpus'], 'frac', round(r.get('frac',0),4), 'fp64', round(r.get('fp64_fma_frac') or 0,3), 'cyc', r.get('cycles_per_wave_step_measured'), 'lds', r.get('lds_busy_frac'), 'clock', r.get('clock_mhz_in_kernel'), 'stale', r.get('stale'), d.get('comm'), 'selfcheck', d.get('sharded_vs_one_gpu_max_rel_dev'))
PY
