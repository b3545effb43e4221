set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab4.jsonl gpurun_out/ab4.err
for rep in 1 2; do for lib in r1 preqe cur w8; do timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$lib.so >> gpurun_out/ab4.jsonl 2>>gpurun_out/ab4.err; done; done
cat gpurun_out/ab4.jsonl; tail -5 gpurun_out/ab4.err
timeout 1800 python -m pytest tests -m gpu -q -rs --durations=5 2>&1 | tail -40 > gpurun_out/pytest_gpu_r2c.log; tail -40 gpurun_out/pytest_gpu_r2c.log
timeout 600 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; python - <<'PY'
import json
for f in ("bench_r2c.json",):
    j=json.load(open("gpurun_out/"+f)); r=j["roofline"]
    print(f, "value %.4g ms/step %.3f kernel ms %.3f frac %.3f clock %s frac@clk %s" % (j["value"], j["ms_per_step"], r["ms_per_launch"], r["frac"], r.get("clock_mhz_during_run"), r.get("frac_at_measured_clock")))
PY
timeout 600 python bench.py --config c4 > gpurun_out/bench_r2c_c4.json 2> gpurun_out/bench_r2c_c4.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/bench_r2c_c4.json")); r=j["roofline"]
print("c4 value %.4g ms/step %.3f kernel ms %.3f frac %.3f n1_share %.4g ratio %.3f full %s" % (j["value"], j["ms_per_step"], r["ms_per_launch"], r["frac"], j["n1_share_value"], j["weak_scaling_ratio"], j["c4_full_one_gpu"]))
PY
