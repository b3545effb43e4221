#!/bin/bash
# After `gpurun -- bash tools/run_round4_measurements.sh`: turn what came back under gpurun_out/ into the committed
# evidence under profiles/ (run here, in the build container, from the repo root).
set -e
R=r04
O=gpurun_out/r04_final
for CFG in c2 c4; do
  D=gpurun_out/prof_$CFG
  { echo "# round 4, build sha256 $(sha256sum stochvolmodels_amd/libsvmc.so | cut -c1-16): rocprofv3 --kernel-trace --stats of \`python bench.py --config $CFG --no-cpu-baseline --no-extra-legs\` (K = 50, W = 10 after 10 pre-warm calls); then the PMC passes of the same command at --steps 3 --warmup 1 (SQ group; FETCH_SIZE; WRITE_SIZE: separate runs, --kernel-trace only)"
    python tools/rocpd_summary.py $(find $D/kt -name '*.db') $(find $D/sq -name '*.db') $(find $D/fetch -name '*.db') $(find $D/write -name '*.db'); } > profiles/${R}_${CFG}_rocprofv3_summary.txt
done
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/${R}_pmc.json
for pair in bench_c2:bench_c2 bench_c4:bench_c4_one_gpu bench_2rank_gloo:bench_c4_2ranks_one_gpu_gloo bench_8rank_gloo:bench_c4_8ranks_one_gpu_gloo bench_c4_torch_rccl_single:bench_c4_torch_rccl_single_rank; do
  src=${pair%%:*}; dst=${pair#*:}
  grep '^{"metric' $O/$src.json | tail -1 > profiles/${R}_$dst.json
done
cp $O/configs.jsonl profiles/${R}_configs.jsonl
cp $O/calib.jsonl profiles/${R}_calibration_bench.jsonl
cp $O/single_kernels.jsonl profiles/${R}_single_kernels.jsonl
cp $O/fused_driver.json profiles/${R}_fused_driver_overhead.json
{ echo "# python -m pytest tests -m gpu -q -s on one MI355X, round-4 build: what the full-size same-stream parity tests, the C5 verdict-parity tests, the sharded C-driver test and the 8-rank bench rehearsal printed"; cat $O/pytest_gpu.log | sed 's/^[.sF]*//'; } > profiles/${R}_fullsize_parity.txt
python - <<'PY'
import json
for f in ['r04_bench_c2','r04_bench_c4_one_gpu','r04_bench_c4_2ranks_one_gpu_gloo','r04_bench_c4_8ranks_one_gpu_gloo','r04_bench_c4_torch_rccl_single_rank']:
    d=json.load(open(f'profiles/{f}.json'))
    r=d.get('roofline',{})
    print(f, '%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], 'N', d['n_gpus'], 'frac', round(r.get('frac',0),4), 'clock', r.get('clock_mhz_in_kernel'), 'frac@clock', r.get('frac_at_sustained_clock'), 'stale', r.get('stale'), 'counters', r.get('insts_per_wave_step_counters'), d.get('comm'), 'rccl seen', d.get('rccl_ranks_seen'), 'c_abi', (d.get('c_abi_route') or {}).get('ms_per_step'), 'selfcheck', d.get('sharded_vs_one_gpu_max_rel_dev'))
PY
