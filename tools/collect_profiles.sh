#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats and the PMC passes of the bench
# command for the C2 and C4 configurations; results land in gpurun_out/ and are summarised locally by
# tools/rocpd_summary.py (text, committed under profiles/) and tools/make_pmc_json.py (profiles/rNN_pmc.json, which
# names the sha256 of the library the counters were collected on).
# Counter passes run with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
for CFG in c2 c4; do
  D=$R/gpurun_out/prof_$CFG
  rm -rf $D; mkdir -p $D
  SHA=$(sha256sum $R/stochvolmodels_amd/libsvmc.so | cut -d' ' -f1)
  if [ $CFG = c2 ]; then echo '{"config": "c2", "paths": 1048576, "steps": 1024, "streamed_paths": 1048576, "streamed_steps": 1024, "lib_sha256": "'$SHA'"}' > $D/config.json
  else echo '{"config": "c4", "paths": 2097152, "steps": 1024, "lib_sha256": "'$SHA'"}' > $D/config.json; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $D/kt -o bench -- python $R/bench.py --config $CFG --no-cpu-baseline --no-extra-legs > $D/kt.log 2>&1; echo $CFG kt rc=$?
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $D/fetch -o bench -- python $R/bench.py --config $CFG $SHORT > $D/fetch.log 2>&1; echo $CFG fetch rc=$?
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $D/write -o bench -- python $R/bench.py --config $CFG $SHORT > $D/write.log 2>&1; echo $CFG write rc=$?
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $D/sq -o bench -- python $R/bench.py --config $CFG $SHORT --no-streamed > $D/sq.log 2>&1; echo $CFG sq rc=$?
  # keep the merge small: databases only
  find $D -type f ! -name '*.db' ! -name '*.log' ! -name 'config.json' -delete
done
cd $R
