O=gpurun_out/r05_f3i; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frozen or fixed or calib or payoff or sets or graph" 2>&1 | tail -3) > $O/tests.log 2>&1
timeout 600 python tools/r05/bench_frozen.py 100000 300 > $O/frozen.jsonl 2>> $O/err.txt
timeout 600 python tools/r05/frozen_breakdown.py 100000 300 > $O/breakdown.jsonl 2>> $O/err.txt
timeout 600 python tools/r05/frozen_scaling.py 100 > $O/scaling.jsonl 2>> $O/err.txt
(cd /tmp && export TMPDIR=/tmp && SVMC_BENCH_FROZEN_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/frozen_prof -o fr -- python $GRAFT_REPO_ROOT/tools/r05/bench_frozen.py 100000 100 > $GRAFT_REPO_ROOT/$O/frozen_prof.log 2>&1)
find $O/frozen_prof -type f ! -name '*.db' -delete 2>/dev/null
cat $O/tests.log $O/frozen.jsonl $O/breakdown.jsonl $O/scaling.jsonl; tail -2 $O/err.txt
