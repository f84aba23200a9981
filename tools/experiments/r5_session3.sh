#!/bin/bash
# round 5, session 3: what the box says about its memory (partition mode, driver state), and the granularity of the slow / fast
# property: streaming-write rate of every GiB of every candidate buffer
R=$(pwd); OUT=$R/gpurun_out/r5s3; mkdir -p $OUT
B=$R/build_variants/k1_stream
( echo "# memory / compute partition"; cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>&1
  echo "# available"; cat /sys/class/drm/card*/device/available_memory_partition 2>&1
  echo "# vram"; cat /sys/class/drm/card*/device/mem_info_vram_total /sys/class/drm/card*/device/mem_info_vram_used 2>&1
  echo "# debugfs"; ls /sys/kernel/debug 2>&1 | head; ls /sys/kernel/debug/dri 2>&1 | head
  echo "# kfd mem banks"; for f in /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties; do echo $f; cat $f; done 2>&1 | head -60
  echo "# rocm-smi"; rocm-smi --showmemuse --showclocks --showtemp 2>&1 | head -60
  echo "# uptime"; uptime; cat /proc/uptime ) > $OUT/sysinfo.txt 2>&1
cat $OUT/sysinfo.txt
$B 5 5 3 regions > $OUT/regions.txt 2>&1
cat $OUT/regions.txt
( rocm-smi --showtemp --showclocks 2>&1 | head -40 ) > $OUT/sysinfo_after.txt
