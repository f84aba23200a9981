#!/bin/bash
# round 6, session 7: where the 0.4 ms between the synchronous predict() and bench.py's step go (ways of waiting for the result)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/experiments/gpu_predict_overhead.py > $O/predict_overhead.txt 2>&1
python $R/tools/experiments/gpu_predict_overhead.py > $O/predict_overhead2.txt 2>&1
for i in 1 2; do
  python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_$i.json 2> /dev/null
done
cat $O/predict_overhead.txt
