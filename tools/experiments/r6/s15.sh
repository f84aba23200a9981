#!/bin/bash
# round 6, session 15: the packed (test, reference) display model of the 16-bit / float temporal kernels after the fix (no inline assembly on the
# result of a transcendental): whole GPU suite, level 0 against the build before it (largest difference relative to the luminance), A/B timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s15
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
OLD=$R/build_variants/r6_pre_pairs.so
FVVDP_LIB=$OLD python $R/tools/experiments/gpu_k1_bits.py save /tmp/k1bits 2>/dev/null | grep -c level0
python $R/tools/experiments/gpu_k1_bits.py cmp /tmp/k1bits 2>/dev/null | grep level0 > $O/bits_cmp.txt
python - $O/bits_cmp.txt <<'PY'
import re,sys
w=0;n=0;same=0
for l in open(sys.argv[1]):
    m=re.search(r"differ (\d+) of \d+, max \|d\| / luminance ([\d.e+-]+)",l)
    if m:
        n+=1; same+= (m.group(1)=="0"); v=float(m.group(2))
        if v>w: w=v; wl=l.strip()
print("cases",n,"bit-identical",same,"largest difference relative to the luminance %.3g"%w); print(wl[:230])
PY
K1S="30:60:u16 60:60:u16 120:120:u16 240:120:u16 30:60:f32rgb 120:120:f32rgb"
for i in 1 2 3; do
  for d in standard_4k standard_hdr_pq; do
    FVVDP_LIB=$OLD PROBE_DISPLAY=$d python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn | sed "s/^/old $d #$i /" >> $O/k1.txt
    PROBE_DISPLAY=$d python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn | sed "s/^/new $d #$i /" >> $O/k1.txt
  done
done
python - $O <<'PY'
import sys,re,collections,statistics as st
d=collections.defaultdict(list)
for l in open(sys.argv[1]+"/k1.txt"):
    m=re.match(r"^(\S+) (\S+) #\d \S+ (.*?): .*K1 ([\d.]+) .*JOD ([\d.]+)",l)
    if m: d[(m.group(2),m.group(3),m.group(1))].append((float(m.group(4)),m.group(5)))
for k in sorted(d): print(k[0],k[1],k[2],[x[0] for x in d[k]],"median",st.median(x[0] for x in d[k]),"JOD",d[k][0][1])
PY
