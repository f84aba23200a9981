#!/bin/bash
# round 4, session 11: randomised parity sweeps of the final build against the oracle (new seeds), incl. the two-level kernel
# forced everywhere (FVVDP_BAND_FUSE=1: the clamp-free variant on small frames) and the stage overlap
R=$(pwd); OUT=$R/gpurun_out/r4s11; mkdir -p $OUT
cd $R
S=tools/experiments
( echo "# gpu_stress.py 600 4101 (default build)"; timeout 2400 python $S/gpu_stress.py 600 4101 2>/dev/null | tail -1
  echo "# FVVDP_BAND_FUSE=1 gpu_stress.py 600 4102 (two-level kernel wherever its border logic is valid: clamp-free variant on SDR displays)"; FVVDP_BAND_FUSE=1 timeout 2400 python $S/gpu_stress.py 600 4102 2>/dev/null | tail -1
  echo "# FVVDP_BAND_FUSE=1 FVVDP_BAND_INRANGE=0 gpu_stress.py 200 4102 (the same cases, variant with clamps)"; FVVDP_BAND_FUSE=1 FVVDP_BAND_INRANGE=0 timeout 2400 python $S/gpu_stress.py 200 4102 2>/dev/null | tail -1
  echo "# FVVDP_PIPELINE=2 gpu_stress.py 300 4103 (stage overlap on two streams)"; FVVDP_PIPELINE=2 NMAX=40 timeout 2400 python $S/gpu_stress.py 300 4103 2>/dev/null | tail -1
  echo "# mid-size frames HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 gpu_stress.py 300 4104"; HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 timeout 3000 python $S/gpu_stress.py 300 4104 2>/dev/null | tail -1
  echo "# gpu_stress_yuv.py 200 4105"; timeout 2400 python $S/gpu_stress_yuv.py 200 4105 2>/dev/null | tail -1
  echo "# gpu_stress_heat.py 100 4106"; timeout 2400 python $S/gpu_stress_heat.py 100 4106 2>/dev/null | tail -1
  echo "# gpu_stress_shapes.py"; timeout 2400 python $S/gpu_stress_shapes.py 2>/dev/null | tail -1
) > $OUT/stress.txt 2>&1
cat $OUT/stress.txt
