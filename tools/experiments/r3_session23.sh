#!/bin/bash
# session 23: foveated kernel with frame-fastest work order (rho map tile L2-resident) vs the default order
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s23
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for V in default fovff; do
  if [ $V = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$V.so; fi
  python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | sed "s/^/$V: /" | tee -a $OUT/fov_order.txt
done
done
unset FVVDP_LIB
for V in default fovff; do
  if [ $V = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$V.so; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/f_$V -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/f_$V.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/w_$V -o a -- python $R/tools/gpu_fov_bandonly.py > /tmp/w_$V.log 2>&1
  python $R/tools/pmc_sq_summary.py band $(find /tmp/f_$V /tmp/w_$V -name "*.db") 2>/dev/null | grep -E "^###|HBM read" | sed "s/^/$V: /" | cut -c1-400 | tee -a $OUT/fov_order_pmc.txt
done
