#!/bin/bash
# round 4, session 30: SQ counters of the 10-bit 4:2:0 YUV temporal kernel at 60 fps
R=$(pwd); OUT=$R/gpurun_out/r4s30; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/y1 -o a -- python $R/tools/gpu_yuv.py 2160x3840x60:10:420:60 > /tmp/y1.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace -d /tmp/y2 -o a -- python $R/tools/gpu_yuv.py 2160x3840x60:10:420:60 > /tmp/y2.log 2>&1
python $R/tools/pmc_sq_summary.py temporal_yuv $(find /tmp/y1 /tmp/y2 -name "*.db") > $OUT/pmc_sq_yuv.md 2>$OUT/err.txt
cat $OUT/pmc_sq_yuv.md | cut -c1-400
tail -n 3 /tmp/y1.log
