#!/bin/bash
# After `gpurun -- bash tools/run_round3_measurements.sh`: turn what came back under gpurun_out/ into the committed
# evidence under profiles/ (run here, in the build container, from the repo root).
set -e
R=r03
for CFG in c2 c4; do
  D=gpurun_out/prof_$CFG
  { echo "# round 3, final build (sha256 $(sha256sum stochvolmodels_amd/libsvmc.so | cut -c1-16)): rocprofv3 --kernel-trace --stats of \`python bench.py --config $CFG --no-cpu-baseline --no-extra-legs\` (K = 50, W = 10 after 10 pre-warm calls); then the PMC passes of the same command at --steps 3 --warmup 1 (SQ group; FETCH_SIZE; WRITE_SIZE: separate runs, --kernel-trace only)"
    python tools/rocpd_summary.py $(find $D/kt -name '*.db') $(find $D/sq -name '*.db') $(find $D/fetch -name '*.db') $(find $D/write -name '*.db'); } > profiles/${R}_final_${CFG}_rocprofv3_summary.txt
done
python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/${R}_pmc.json
grep '^{"metric' gpurun_out/bench_r3_final.json | tail -1 > profiles/${R}_bench_c2.json
grep '^{"metric' gpurun_out/bench_r3_final_c4.json | tail -1 > profiles/${R}_bench_c4_one_gpu.json
grep '^{"metric' gpurun_out/bench_r3_final_2rank_gloo.json | tail -1 > profiles/${R}_bench_c4_2ranks_one_gpu_gloo.json
grep '^{"metric' gpurun_out/bench_r3_final_c4_rcclcomm.json | tail -1 > profiles/${R}_bench_c4_rcclcomm_single_rank.json
grep '^{"metric' gpurun_out/bench_r3_final_c4_torch_rccl_single.json | tail -1 > profiles/${R}_bench_c4_torch_rccl_single_rank.json
grep '^{"metric' gpurun_out/bench_r3_400steps.json | tail -1 > profiles/${R}_bench_c2_400_steps.json
cp gpurun_out/configs_r3.jsonl profiles/${R}_configs.jsonl
cp gpurun_out/rough_r3.jsonl profiles/${R}_rough_bench.jsonl
cp gpurun_out/ab_final_r3.jsonl profiles/${R}_final_single_kernels.jsonl
cp gpurun_out/sync_latency_r3.json profiles/${R}_sync_latency.json
cp gpurun_out/fused_driver_r3.json profiles/${R}_fused_driver_overhead.json
grep "^{" gpurun_out/calib_r3.log > profiles/${R}_calibration_bench.jsonl || true
{ echo "# tests/test_gpu_fullsize.py and the sharded C-driver test on one MI355X (python -m pytest tests -m gpu -q -s), round-3 final build: GPU vs the CPU oracle on the same stream"; cat gpurun_out/pytest_gpu_r3_final.log | sed 's/^[.sF]*//'; } > profiles/${R}_fullsize_parity.txt
{ echo "# rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE over tools/ubench/c3_probe.py (Heston kernels alone, 2^22 paths x 512 steps; dispatch order: base euler x3, base qe x3, btc euler x3, btc qe x3), round-3 final build"; python tools/rocpd_summary.py $(find gpurun_out/c3pmc -name '*.db'); } > profiles/${R}_c3_heston_pmc.txt
python - <<'PY'
import json
for f in ['r03_bench_c2','r03_bench_c4_one_gpu','r03_bench_c4_2ranks_one_gpu_gloo','r03_bench_c4_rcclcomm_single_rank','r03_bench_c4_torch_rccl_single_rank','r03_bench_c2_400_steps']:
    d=json.load(open(f'profiles/{f}.json'))
    r=d.get('roofline',{})
    print(f, '%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], 'N', d['n_gpus'], 'frac', round(r.get('frac',0),4), round(r.get('frac_in_stream_int32_cost',0),4), 'stale', r.get('stale'), 'counters', r.get('insts_per_wave_step_counters'), d.get('comm'), 'rccl seen', d.get('rccl_ranks_seen'), (d.get('rccl_route') or {}).get('ms_per_step'), d.get('weak_scaling_ratio'))
    if f.endswith('400_steps'): print('   ', d['ms_per_step_profile'])
PY
