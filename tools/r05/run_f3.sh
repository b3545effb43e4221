O=gpurun_out/r05_f3d; mkdir -p $O
timeout 600 python tools/r05/frozen_breakdown.py 100000 300 > $O/breakdown.jsonl 2>> $O/frozen.err
cat $O/breakdown.jsonl; tail -3 $O/frozen.err
