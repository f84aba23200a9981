#!/bin/bash
# round 6, session 18: the whole GPU suite + smoke + the randomised sweeps against the oracle + the profile bundle on the final build of the round
# (as s12.sh, with larger sweeps, on the final build: + full-screen resize kernels)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s18
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?" >> $O/smoke.log
cd /tmp && export TMPDIR=/tmp
python $R/tools/experiments/gpu_stress.py 1500 6501 > $O/stress_default.txt 2>&1
FVVDP_BAND_FUSE=1 python $R/tools/experiments/gpu_stress.py 400 6502 > $O/stress_fuse.txt 2>&1
FVVDP_BAND_FUSE=1 FVVDP_BAND_INRANGE=0 python $R/tools/experiments/gpu_stress.py 200 6502 > $O/stress_fuse_clamps.txt 2>&1
HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 python $R/tools/experiments/gpu_stress.py 200 6503 > $O/stress_mid.txt 2>&1
HLO=40 HHI=400 WLO=2300 WHI=4200 NMAX=4 python $R/tools/experiments/gpu_stress.py 100 6504 > $O/stress_wide.txt 2>&1
python $R/tools/experiments/gpu_stress_yuv.py 1000 6505 > $O/stress_yuv.txt 2>&1
python $R/tools/experiments/gpu_stress_heat.py 60 6506 > $O/stress_heat.txt 2>&1
python $R/tools/experiments/gpu_stress_shapes.py > $O/stress_shapes.txt 2>&1
tail -n 1 $O/pytest_gpu.log $O/smoke.log $O/stress_*.txt
cd $R
bash tools/collect_profiles.sh > $R/gpurun_out/prof_bundle_log.txt 2>&1; tail -2 $R/gpurun_out/prof_bundle_log.txt
