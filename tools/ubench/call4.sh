set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab5.jsonl gpurun_out/ab5.err
for rep in 1 2; do for lib in r1 cur pairs; do timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$lib.so >> gpurun_out/ab5.jsonl 2>>gpurun_out/ab5.err; done; done
cat gpurun_out/ab5.jsonl; tail -5 gpurun_out/ab5.err
timeout 1800 python -m pytest tests -m gpu -q -rs --durations=5 2>&1 | tail -40 > gpurun_out/pytest_gpu_r2d.log; tail -40 gpurun_out/pytest_gpu_r2d.log
timeout 900 python tools/c5_bias.py 23 > gpurun_out/c5_bias.json 2> gpurun_out/c5_bias.err; tail -12 gpurun_out/c5_bias.err
for ev in 0 1; do SVMC_BENCH_NO_KERNEL_EVENTS=$ev timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r2d_c4_ev$ev.json 2> gpurun_out/bench_r2d_c4_ev$ev.err; python - <<PY
import json
j=json.load(open("gpurun_out/bench_r2d_c4_ev$ev.json")); r=j["roofline"]
print("c4 noevents=$ev value %.4g ms/step %.3f kernel ms %.3f n1_share %.4g ms %.3f ratio %.3f" % (j["value"], j["ms_per_step"], r["ms_per_launch"], j["n1_share_value"], j["n1_share_ms_per_step"], j["weak_scaling_ratio"]))
PY
done
