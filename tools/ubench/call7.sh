cd $GRAFT_REPO_ROOT
for cfg in c4 c2 c4; do timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extra-legs --no-streamed > gpurun_out/bench_r2g_$cfg.json 2>/dev/null; python - <<PY
import json
j=json.load(open("gpurun_out/bench_r2g_$cfg.json"))
print("$cfg", j["ms_per_step"], j["ms_per_step_profile"])
PY
done
