#!/bin/bash
# round 6, session 1: new tests (RCCL on one rank, stated LUT range, full-size D maps, 4K-geometry corner), bench with / without the
# forced collective, RCCL kernel trace, configs[3] bench + placement A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s1
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fullsize_maps.py tests/test_gpu_fused.py::test_stated_table_range_is_enforced_not_trusted \
  tests/test_gpu_sharding.py::test_rccl_executes_on_one_rank_through_the_step_path tests/test_gpu_sharding.py::test_bench_one_rank_takes_the_same_step_path \
  -x -q -s > $O/pytest_new.log 2>&1
echo "pytest rc $?" >> $O/pytest_new.log
cp gpurun_out/fullsize_maps_*.json $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_auto.json 2> $O/bench_auto.err
python $R/bench.py --collective off --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_off.json 2> $O/bench_off.err
python $R/bench.py --collective force --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_force.json 2> $O/bench_force.err
python $R/bench.py --collective off --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_off2.json 2>/dev/null
python $R/bench.py --collective force --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_force2.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/kt_rccl -o rccl -- python $R/bench.py --collective force --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_rccl_profiled.json 2> /tmp/kt_rccl.err
python $R/tools/rocpd_summary.py $(find /tmp/kt_rccl -name "*.db" | head -1) --band-levels 7 > $O/kernel_trace_rccl.md
python $R/bench.py --config 3 --no-cpu-baseline --no-h2d > $O/bench_config3.json 2> $O/bench_config3.err
for i in 1 2 3; do
  for P in 0 1 default; do
    if [ $P = default ]; then env -u FVVDP_PLACEMENT_PROBE python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" > $O/fov_ab_${P}_$i.txt
    else FVVDP_PLACEMENT_PROBE=$P python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" > $O/fov_ab_${P}_$i.txt; fi
  done
done
ls -la $O
