#!/bin/bash
# round 6, session 9: one drain of the vector-memory counter BEFORE the frame loop of the temporal kernels (the loop header's wait then is
# the back edge's own, counted, instead of vmcnt(0) every TD / FL frames): A/B against the build before it (build_variants/r6_pre_drain.so),
# same box, alternating processes -- YUV ingest, the RGB temporal kernel at several rates / sample types, and the bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420 1080x1920x60:8:420"
K1S="30:60:u8 60:60:u8 120:120:u8 144:120:u8 30:60:u16 60:60:u16 30:60:f32rgb 30:60:f32gray"
OLD=$R/build_variants/${OLDLIB:-r6_pre_drain}.so
for i in 1 2 3; do
  FVVDP_LIB=$OLD python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/old #$i /" >> $O/yuv.txt
  python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/new #$i /" >> $O/yuv.txt
  FVVDP_LIB=$OLD python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn | sed "s/^/old #$i /" >> $O/k1.txt
  python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn | sed "s/^/new #$i /" >> $O/k1.txt
  FVVDP_LIB=$OLD python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_old_$i.json 2>/dev/null
  python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_new_$i.json 2>/dev/null
done
cat $O/yuv.txt $O/k1.txt
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], j["ms_per_step"], j["roofline_k1"]["median_launch_ms"] if "median_launch_ms" in j.get("roofline_k1",{}) else j.get("roofline_k1",{}).get("achieved"), j["roofline"]["median_launch_ms"])
PY
done
