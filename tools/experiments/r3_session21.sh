#!/bin/bash
# session 21: SQ counters of the 16-slot YUV ingest kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s21
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"
SQ3="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_DEP_WAIT"
for C in 2160x3840x60:10:420:60 2160x3840x60:8:420; do
  T=$(echo $C | tr ':x' '__')
  rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/s1_$T -o a -- python $R/tools/gpu_yuv.py $C > /tmp/s1_$T.log 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace -d /tmp/s2_$T -o a -- python $R/tools/gpu_yuv.py $C > /tmp/s2_$T.log 2>&1
  rocprofv3 --pmc $SQ3 --kernel-trace -d /tmp/s5_$T -o a -- python $R/tools/gpu_yuv.py $C > /tmp/s5_$T.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/s3_$T -o a -- python $R/tools/gpu_yuv.py $C > /tmp/s3_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/s4_$T -o a -- python $R/tools/gpu_yuv.py $C > /tmp/s4_$T.log 2>&1
  python $R/tools/pmc_sq_summary.py temporal_yuv $(find /tmp/s1_$T /tmp/s2_$T /tmp/s3_$T /tmp/s4_$T /tmp/s5_$T -name "*.db") > $OUT/pmc_sq_$T.md 2> $OUT/err_$T.txt
  cat $OUT/pmc_sq_$T.md
  tail -3 /tmp/s5_$T.log
done
