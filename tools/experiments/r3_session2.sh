#!/bin/bash
# round-3 GPU session 2: A/B of foveated-kernel variants, K1 placement with the VMM allocator, 64-slot ring occupancy, host CPU probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( echo "nproc $(nproc)"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; rocm-smi --showclocks 2>/dev/null | head -20 ) > $OUT/host.txt 2>&1
for v in default nopad pad33 wpb6 wpb12 phase4; do
  L=$R/build_variants/$v.so; [ $v = default ] && L=$R/fovvideovdp_amd/libfvvdp_hip.so
  echo "== $v" >> $OUT/fov_ab.txt
  FVVDP_LIB=$L timeout 200 python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per|^kernel" | tail -3 >> $OUT/fov_ab.txt
done
# second round in reverse order (box drift)
for v in phase4 wpb12 wpb6 pad33 nopad default; do
  L=$R/build_variants/$v.so; [ $v = default ] && L=$R/fovvideovdp_amd/libfvvdp_hip.so
  echo "== $v (2)" >> $OUT/fov_ab.txt
  FVVDP_LIB=$L timeout 200 python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel" | tail -1 >> $OUT/fov_ab.txt
done
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d /tmp/q1 -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/q1.log 2>&1
python $R/tools/pmc_sq_summary.py band $(find /tmp/q1 -name "*.db") > $OUT/pmc_fov_padded.md 2>&1
echo "== hipMalloc" > $OUT/k1_placement.txt
timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>/dev/null | grep ctx >> $OUT/k1_placement.txt
echo "== FVVDP_ALLOC=vmm" >> $OUT/k1_placement.txt
FVVDP_ALLOC=vmm timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror" >> $OUT/k1_placement.txt
echo "== FVVDP_ALLOC=vmm FVVDP_VMM_ALIGN_MB=1024" >> $OUT/k1_placement.txt
FVVDP_ALLOC=vmm FVVDP_VMM_ALIGN_MB=1024 timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror" >> $OUT/k1_placement.txt
for v in default w64_2; do
  L=$R/build_variants/$v.so; [ $v = default ] && L=$R/fovvideovdp_amd/libfvvdp_hip.so
  FVVDP_LIB=$L timeout 300 python $R/tools/gpu_fps.py 144:120:u8 240:120:u8 144:60:u16 144:60:f32gray 2>/dev/null | grep -v Warn >> $OUT/ring64_ab.txt
done
ls -la $OUT
