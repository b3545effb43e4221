set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab3.jsonl gpurun_out/ab3.err
for rep in 1 2; do for lib in base preqe cur w8; do timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$lib.so >> gpurun_out/ab3.jsonl 2>>gpurun_out/ab3.err; done; done
cat gpurun_out/ab3.jsonl; tail -5 gpurun_out/ab3.err
timeout 1800 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 > gpurun_out/pytest_gpu_r2b.log; tail -30 gpurun_out/pytest_gpu_r2b.log
timeout 600 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; tail -c 1800 gpurun_out/bench_r2b.json; tail -3 gpurun_out/bench_r2b.err
timeout 600 python bench.py --config c4 > gpurun_out/bench_r2b_c4.json 2> gpurun_out/bench_r2b_c4.err; tail -c 1500 gpurun_out/bench_r2b_c4.json; tail -3 gpurun_out/bench_r2b_c4.err
SVMC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/bench_r2b_2rank_gloo.json 2> gpurun_out/bench_r2b_2rank_gloo.err; tail -c 1500 gpurun_out/bench_r2b_2rank_gloo.json; tail -3 gpurun_out/bench_r2b_2rank_gloo.err
