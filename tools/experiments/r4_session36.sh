#!/bin/bash
# round 4, session 36: randomised parity sweeps of the FINAL build against the oracle (new seeds): default; the two-level kernel forced
# everywhere with 2 / 4 waves per workgroup and with short chunks forced (FVVDP_BAND2_KR2); wide frames (the >= 2560-column rule)
R=$(pwd); OUT=$R/gpurun_out/r4s36; mkdir -p $OUT
cd $R
S=tools/experiments
( echo "# gpu_stress.py 500 4201 (default build)"; timeout 2400 python $S/gpu_stress.py 500 4201 2>/dev/null | tail -1
  echo "# FVVDP_BAND_FUSE=1 FVVDP_BAND2_WPB=4 FVVDP_BAND2_KR2=2 gpu_stress.py 400 4202 (two-level kernel everywhere, 4 waves per workgroup where 4 divides the strips, short chunks of 2 rows)"; FVVDP_BAND_FUSE=1 FVVDP_BAND2_WPB=4 FVVDP_BAND2_KR2=2 timeout 2400 python $S/gpu_stress.py 400 4202 2>/dev/null | tail -1
  echo "# FVVDP_BAND_FUSE=1 FVVDP_BAND2_WPB=2 FVVDP_BAND2_KR2=1 gpu_stress.py 300 4203"; FVVDP_BAND_FUSE=1 FVVDP_BAND2_WPB=2 FVVDP_BAND2_KR2=1 timeout 2400 python $S/gpu_stress.py 300 4203 2>/dev/null | tail -1
  echo "# FVVDP_K1_TICKET=0 gpu_stress.py 200 4204 (temporal kernel: one workgroup per block)"; FVVDP_K1_TICKET=0 timeout 2400 python $S/gpu_stress.py 200 4204 2>/dev/null | tail -1
  echo "# wide frames HLO=40 HHI=400 WLO=2300 WHI=4200 NMAX=4 gpu_stress.py 150 4205 (the >= 2560-column rule, two-level launches by default)"; HLO=40 HHI=400 WLO=2300 WHI=4200 NMAX=4 timeout 3000 python $S/gpu_stress.py 150 4205 2>/dev/null | tail -1
  echo "# mid-size frames HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 gpu_stress.py 200 4206"; HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 timeout 3000 python $S/gpu_stress.py 200 4206 2>/dev/null | tail -1
  echo "# gpu_stress_yuv.py 100 4207"; timeout 2400 python $S/gpu_stress_yuv.py 100 4207 2>/dev/null | tail -1
  echo "# gpu_stress_heat.py 60 4208"; timeout 2400 python $S/gpu_stress_heat.py 60 4208 2>/dev/null | tail -1
  echo "# gpu_stress_shapes.py"; timeout 2400 python $S/gpu_stress_shapes.py 2>/dev/null | tail -1
) > $OUT/stress.txt 2>&1
cat $OUT/stress.txt
