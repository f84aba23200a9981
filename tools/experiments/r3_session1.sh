#!/bin/bash
# round-3 GPU session 1: correctness of everything + baseline numbers + PMC evidence (one gpurun call)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
timeout 600 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- python $R/bench.py > $OUT/bench_profiled.json 2> /tmp/kt.err
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --band-levels 7 --dispatches band2_kernel > $OUT/kernel_trace_bench.md
timeout 300 python $R/tools/gpu_fps.py 30:60:u8 60:120:u8 120:120:u8 144:120:u8 240:120:u8 30:60:u16 60:60:u16 120:60:u16 144:60:u16 30:60:f32rgb 120:60:f32rgb 30:60:f32gray 2>/dev/null | grep -v Warn > $OUT/fps_probe.txt
timeout 300 python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 > $OUT/yuv_probe.txt 2>/dev/null
timeout 300 python $R/tools/gpu_config4.py > $OUT/fov_probe.txt 2>/dev/null
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"
for T in bandonly fov_bandonly; do
  timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/p1_$T -o a -- python $R/tools/gpu_$T.py > /tmp/p1_$T.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d /tmp/p2_$T -o a -- python $R/tools/gpu_$T.py > /tmp/p2_$T.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p3_$T -o a -- python $R/tools/gpu_$T.py > /tmp/p3_$T.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p4_$T -o a -- python $R/tools/gpu_$T.py > /tmp/p4_$T.log 2>&1
  python $R/tools/pmc_sq_summary.py band $(find /tmp/p1_$T /tmp/p2_$T /tmp/p3_$T /tmp/p4_$T -name "*.db") > $OUT/pmc_sq_$T.md 2> $OUT/pmc_sq_$T.err
done
tail -3 /tmp/p2_fov_bandonly.log > $OUT/pmc_log_tail.txt
ls -la $OUT
