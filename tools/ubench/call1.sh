set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab2.jsonl
for rep in 1 2; do for lib in base cur r10 s64 s80 s96; do timeout 300 python tools/ubench/ab_kernels.py tools/ubench/ab/libsvmc_$lib.so >> gpurun_out/ab2.jsonl 2>>gpurun_out/ab2.err; done; done
cat gpurun_out/ab2.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -30 > gpurun_out/pytest_gpu_r2a.log; tail -15 gpurun_out/pytest_gpu_r2a.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=10 2>&1 | tail -40 > gpurun_out/pytest_full_r2a.log; tail -40 gpurun_out/pytest_full_r2a.log
timeout 600 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; tail -c 2500 gpurun_out/bench_r2a.json
