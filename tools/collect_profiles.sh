#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats and the two PMC passes of the
# default bench command; results land in gpurun_out/ and are summarised locally by tools/rocpd_summary.py.
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py $ARGS > $R/gpurun_out/prof_kt.log 2>&1; echo kt rc=$?
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1; echo write rc=$?
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/prof_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed > $R/gpurun_out/prof_sq.log 2>&1; echo sq rc=$?
cd $R
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
tail -c 600 gpurun_out/bench_final.json
