#!/bin/bash
# round 5, session 2: (a) more variants of the temporal kernel's stream on slow and fast buffers of one process -- several waves per
# workgroup (real kernel and replay, with / without a barrier per frame), occupancy, the frame in slabs (one launch each), one round
# of resident waves, plain streaming write / read of the same buffers; (b) two counter passes with the per-instance values kept
# (rocprofv3 JSON output): is the write-request stall of a slow buffer spread over the L2 channels or concentrated?
R=$(pwd); OUT=$R/gpurun_out/r5s2; mkdir -p $OUT
B=$R/build_variants/k1_stream
N="5 5 3"
cd /tmp && export TMPDIR=/tmp
$B $N > $OUT/stream.txt 2>&1
cat $OUT/stream.txt
i=0
for SET in "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY TCC_REQ"; do
  rocprofv3 --pmc $SET --kernel-trace --output-format json -d /tmp/r5j$i -o p -- $B $N pmc > $OUT/pmcj$i.log 2> $OUT/pmcj$i.err
  J=$(find /tmp/r5j$i -name "*.json" | head -1)
  ls -la $J
  python $R/tools/pmc_instances.py $J $OUT/pmcj$i.log > $OUT/instances$i.md 2> $OUT/instances$i.err
  head -c 3000 $OUT/instances$i.md
  i=$((i+1))
done
