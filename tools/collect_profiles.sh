#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats and the two PMC passes of the
# default bench command; results land in gpurun_out/ and are summarised locally by tools/rocpd_summary.py.
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
ARGS="--no-cpu-baseline"          # the default bench command (K = 50, W = 10) minus the host-side CPU baseline
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py $ARGS > $R/gpurun_out/prof_kt.log 2>&1; echo kt rc=$?
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1; echo write rc=$?
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/prof_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed > $R/gpurun_out/prof_sq.log 2>&1; echo sq rc=$?
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/prof_clk -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed > $R/gpurun_out/prof_clk.log 2>&1; echo clk rc=$?
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-trace -d $R/gpurun_out/prof_lds -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed > $R/gpurun_out/prof_lds.log 2>&1; echo lds rc=$?
cd $R
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
tail -c 600 gpurun_out/bench_final.json
