#!/bin/bash
# After `gpurun -- bash tools/run_round2_measurements.sh`: turn what came back under gpurun_out/ into the committed
# evidence under profiles/ (run here, in the build container, from the repo root).
set -e
for CFG in c2 c4; do
  D=gpurun_out/prof_$CFG
  { echo "# round 2, final build: rocprofv3 --kernel-trace --stats of \`python bench.py --config $CFG --no-cpu-baseline --no-extra-legs\` (K = 50, W = 10 after 200 pre-warm calls); then the PMC passes of the same command at --steps 3 --warmup 1 (SQ group; FETCH_SIZE; WRITE_SIZE: separate runs, --kernel-trace only)"
    python tools/rocpd_summary.py $(find $D/kt -name '*.db') $(find $D/sq -name '*.db') $(find $D/fetch -name '*.db') $(find $D/write -name '*.db'); } > profiles/r02_final_${CFG}_rocprofv3_summary.txt
done
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r02_pmc.json
grep '^{"metric' gpurun_out/bench_r2_final.json | tail -1 > profiles/r02_bench_c2.json
grep '^{"metric' gpurun_out/bench_r2_final_c4.json | tail -1 > profiles/r02_bench_c4_one_gpu.json
grep '^{"metric' gpurun_out/bench_r2_final_2rank_gloo.json | tail -1 > profiles/r02_bench_c4_2ranks_one_gpu_gloo.json
grep '^{"metric' gpurun_out/bench_r2_final_c4_rcclcomm.json | tail -1 > profiles/r02_bench_c4_rcclcomm_single_rank.json
cp gpurun_out/configs_r2.jsonl profiles/r02_configs.jsonl
cp gpurun_out/rough_r2.jsonl profiles/r02_rough_bench.jsonl
cp gpurun_out/ab_final.jsonl profiles/r02_final_single_kernels.jsonl
grep "^{" gpurun_out/calib_r2.log >> profiles/r02_calibration_bench.jsonl
python - <<'PY'
import json
for f in ['r02_bench_c2','r02_bench_c4_one_gpu','r02_bench_c4_2ranks_one_gpu_gloo','r02_bench_c4_rcclcomm_single_rank']:
    d=json.load(open(f'profiles/{f}.json'))
    r=d.get('roofline',{})
    print(f, '%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], 'N', d['n_gpus'], 'frac', round(r.get('frac',0),4), d.get('comm'), d.get('n1_share_value'), d.get('weak_scaling_ratio'))
PY
