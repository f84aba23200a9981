#!/bin/bash
# round 5, session 1: the temporal kernel's slow / fast destination buffers -- one process, 13 level-0 candidates, the real kernel and
# replays of its address stream on each (tools/microbench/k1_stream.hip); then the same binary under rocprofv3 --pmc, one pass per
# counter group (every pass = its own draw of allocations; durations under the counters say which buffers were slow in that pass).
R=$(pwd); OUT=$R/gpurun_out/r5s1; mkdir -p $OUT
B=$R/build_variants/k1_stream
N="5 5 3"
cd /tmp && export TMPDIR=/tmp
$B $N > $OUT/stream.txt 2>&1
cat $OUT/stream.txt
SETS=(
 "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_LEVEL"
 "TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ"
 "TCC_EA0_RDREQ TCC_EA0_WRREQ_64B TCC_EA0_RDREQ_32B TCC_BUBBLE"
 "TCC_REQ TCC_HIT TCC_MISS TCC_WRITEBACK"
 "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS"
 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ"
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_MULTI_MISS"
 "TCC_BUSY TCC_CYCLE TCC_IB_STALL TCC_SRC_FIFO_FULL"
 "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES"
)
ARGS=""
for i in "${!SETS[@]}"; do
  rocprofv3 --pmc ${SETS[$i]} --kernel-trace -d /tmp/r5p$i -o p -- $B $N pmc > $OUT/pmc$i.log 2> $OUT/pmc$i.err
  DB=$(find /tmp/r5p$i -name "*.db" | head -1)
  [ -n "$DB" ] && ARGS="$ARGS $OUT/pmc$i.log:$DB"
done
python $R/tools/pmc_k1_mode.py "temporal_vec_kernel" $ARGS > $OUT/pmc_k1.md 2> $OUT/pmc_k1.err
python $R/tools/pmc_k1_mode.py "replay" $ARGS > $OUT/pmc_replay.md 2>> $OUT/pmc_k1.err
head -60 $OUT/pmc_k1.md
# the library on this box: bench line, and six contexts held at once
cd $R
python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
FVVDP_PLACEMENT_PROBE=0 python tools/experiments/gpu_alloc_draws.py 6 > $OUT/draws.txt 2>/dev/null
cat $OUT/draws.txt
