#!/bin/bash
# round 6, session 10: first filter tap as a packed multiply (no accumulator clears) on top of the entry drain: tests, then a three-way A/B
# (build before the drain / drain only / drain + first tap), same box, alternating processes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s10
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py tests/test_gpu_sizes.py tests/test_gpu_fused.py tests/test_gpu_cabi_plain.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420"
K1S="30:60:u8 60:60:u8 120:120:u8 144:120:u8 30:60:u16 30:60:f32rgb"
for i in 1 2 3; do
  for v in ${LIBS:-r6_pre_drain r6_drain new}; do
    if [ $v = new ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$v.so; fi
    python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/$v #$i /" >> $O/yuv.txt
    python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn | sed "s/^/$v #$i /" >> $O/k1.txt
    python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $O/bench_${v}_$i.json 2>/dev/null
  done
done
unset FVVDP_LIB
python - $O <<'PY'
import json,sys,glob,re,collections,statistics as st
O=sys.argv[1]
for fn,pat in (("yuv.txt",r"^(\S+) #\d \S+ (\S+) .*temporal ([\d.]+)"),("k1.txt",r"^(\S+) #\d \S+ (.*?): .*K1 ([\d.]+)")):
    d=collections.defaultdict(list)
    for l in open(O+"/"+fn):
        m=re.match(pat,l)
        if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
    for k in sorted(d): print(fn,k[0],k[1],d[k],"median",st.median(d[k]))
d=collections.defaultdict(list)
for f in sorted(glob.glob(O+"/bench_*.json")):
    j=json.load(open(f)); v=re.match(r"bench_(.*)_\d",f.split("/")[-1]).group(1)
    d[v].append((j["ms_per_step"],j["roofline_k1"].get("median_launch_ms"),j["roofline"]["median_launch_ms"]))
for v in d: print("bench",v,d[v])
PY
