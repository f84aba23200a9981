#!/bin/bash
# session 26: final check of the ring-8 / map-first / frame-fastest band_item: tests, foveated config, bench, PMC of the foveated kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s26
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | tail -2 | tee -a $OUT/fov.txt; done
python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | cut -c1-400 | tee $OUT/bench.txt
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/s1 -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/s1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/s3 -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/s3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/s4 -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/s4.log 2>&1
python $R/tools/pmc_sq_summary.py band $(find /tmp/s1 /tmp/s3 /tmp/s4 -name "*.db") 2>/dev/null | grep -E "^###|^shares" | head -4 | cut -c1-500 | tee $OUT/pmc.txt
