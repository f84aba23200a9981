#!/bin/bash
# round 6, session 2: fixed RCCL test, RCCL one-rank trace (sum: no kernel; avg: oneRankReduce), foveated level-0 variants
# (wave priority, 6 waves per workgroup, 4-pixel phases) and ablations (no tail / L2-resident rows) on the round-6 build
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s2
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fullsize_maps.py tests/test_gpu_sharding.py::test_rccl_executes_on_one_rank_through_the_step_path \
  tests/test_gpu_sharding.py::test_bench_one_rank_takes_the_same_step_path tests/test_gpu_sharding.py::test_bench_multi_rank_step_under_gloo \
  tests/test_gpu_sharding.py::test_bench_frame_sharded_under_gloo -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
python $R/tools/gpu_rccl_one_rank.py > $O/rccl_one_rank.txt 2> $O/rccl_one_rank.err
rocprofv3 --kernel-trace --stats -d /tmp/kt_r1 -o r1 -- python $R/tools/gpu_rccl_one_rank.py > $O/rccl_one_rank_profiled.txt 2> /tmp/kt_r1.err
python $R/tools/rocpd_summary.py $(find /tmp/kt_r1 -name "*.db" | head -1) --band-levels 7 > $O/kernel_trace_rccl_one_rank.md
python - <<PY > $O/rccl_kernels.txt
import sqlite3, glob
db = glob.glob("/tmp/kt_r1/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for r in con.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%ccl%' or name like '%oneRank%' or name like '%Reduce%' group by name").fetchall():
    print(r)
PY
python $R/bench.py --collective auto --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_auto.json 2> $O/bench_auto.err
for i in 1 2 3; do
  for V in base fov_prio fov_tailprio fov_wpb6 fov_phase4 fov_abl_tail fov_abl_mem; do
    if [ $V = base ]; then python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" > $O/fov_${V}_$i.txt
    else FVVDP_LIB=$R/build_variants/$V.so python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" > $O/fov_${V}_$i.txt; fi
  done
done
ls $O
