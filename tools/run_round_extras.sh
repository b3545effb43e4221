set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
SVMC_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo 2rank rc=$?
tail -c 1500 gpurun_out/bench_2rank_gloo.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rough -o rough -- python $R/tools/bench_rough.py > $R/gpurun_out/prof_rough.log 2>&1; echo rough rc=$?
cd $R
python tools/bench_calibration.py 100000 2>&1 | grep nb_path
