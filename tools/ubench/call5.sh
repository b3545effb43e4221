set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/bench_r2e_c4.json 2> gpurun_out/bench_r2e_c4.err; python - <<PY
import json
j=json.load(open("gpurun_out/bench_r2e_c4.json")); r=j["roofline"]
print("c4 value %.4g ms/step %.3f kernel ms %.3f n1_share ms %.3f profile %s" % (j["value"], j["ms_per_step"], r["ms_per_launch"], j["n1_share_ms_per_step"], j["ms_per_step_profile"]))
PY
bash tools/collect_profiles.sh
ls -R gpurun_out/prof_c2 | head -30
du -sh gpurun_out/prof_c2 gpurun_out/prof_c4
