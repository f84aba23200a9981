R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for i in 1 2 3; do
  for d in standard_4k standard_hdr_pq; do
  PROBE_DISPLAY=$d python $R/tools/gpu_fps.py 30:60:u16 60:60:u16 30:60:f32rgb 60:60:f32rgb 2>/dev/null | grep -v Warn | sed "s/^/base $d #$i /"
  FVVDP_LIB=$R/build_variants/k1_early_u16.so PROBE_DISPLAY=$d python $R/tools/gpu_fps.py 30:60:u16 60:60:u16 2>/dev/null | grep -v Warn | sed "s/^/early $d #$i /"
  FVVDP_LIB=$R/build_variants/k1_early_f32.so PROBE_DISPLAY=$d python $R/tools/gpu_fps.py 30:60:f32rgb 60:60:f32rgb 2>/dev/null | grep -v Warn | sed "s/^/early $d #$i /"
  done
done | sort -k2,2 -k7,9 -k1,1
