set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab6.jsonl gpurun_out/ab6.err
for rep in 1 2; do for lib in r1 cur cw7 cw6; do timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$lib.so >> gpurun_out/ab6.jsonl 2>>gpurun_out/ab6.err; done; done
cat gpurun_out/ab6.jsonl; tail -5 gpurun_out/ab6.err
timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r2f_c4.json 2> gpurun_out/bench_r2f_c4.err; python - <<PY
import json
j=json.load(open("gpurun_out/bench_r2f_c4.json")); r=j["roofline"]
print("c4 value %.4g ms/step %.3f kernel ms %.3f n1_share ms %.3f profile %s" % (j["value"], j["ms_per_step"], r["ms_per_launch"], j["n1_share_ms_per_step"], j["ms_per_step_profile"]))
PY
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import numpy as np
torch.cuda.set_device(0)
t=time.perf_counter(); torch.cuda.synchronize(); print("sync1 ms", 1e3*(time.perf_counter()-t))
t=time.perf_counter(); torch.cuda.synchronize(); print("sync2 ms", 1e3*(time.perf_counter()-t))
import stochvolmodels_amd as sv
import bench
wl=bench.make_workload("c4", sv); P=sv.LOGSV_BTC_PARAMS
for i in range(5): bench.price(sv, wl, P, 1<<21, i)
t=time.perf_counter(); torch.cuda.synchronize(); print("sync after 5 c4 calls ms", 1e3*(time.perf_counter()-t))
for i in range(50): bench.price(sv, wl, P, 1<<21, i)
t=time.perf_counter(); torch.cuda.synchronize(); print("sync after 50 c4 calls ms", 1e3*(time.perf_counter()-t))
t=time.perf_counter(); torch.cuda.synchronize(); print("sync again ms", 1e3*(time.perf_counter()-t))
wl=bench.make_workload("c2", sv)
for i in range(50): bench.price(sv, wl, P, 1<<20, i)
t=time.perf_counter(); torch.cuda.synchronize(); print("sync after 50 c2 calls ms", 1e3*(time.perf_counter()-t))
PY
