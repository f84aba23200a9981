#!/bin/bash
# round 6, session 14: the closed-form display model of the 16-bit / float temporal kernels on (test, reference) pairs (packed instructions, the same
# operations in the same order): bits of level 0 against the build before it, the temporal / parity tests, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OLD=$R/build_variants/r6_pre_pairs.so
FVVDP_LIB=$OLD python $R/tools/experiments/gpu_k1_bits.py 2>/dev/null | grep -v Warn > $O/bits_old.txt
python $R/tools/experiments/gpu_k1_bits.py 2>/dev/null | grep -v Warn > $O/bits_new.txt
echo "cases $(wc -l < $O/bits_new.txt), lines that differ: $(diff $O/bits_old.txt $O/bits_new.txt | grep -c '^>')" | tee $O/bits_summary.txt
diff $O/bits_old.txt $O/bits_new.txt | head -20
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py tests/test_gpu_sizes.py tests/test_gpu_max_sizes.py tests/test_gpu_fused.py tests/test_gpu_cabi_plain.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
cd /tmp
K1S="30:60:u16 60:60:u16 120:120:u16 30:60:f32rgb 30:60:f32gray"
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
    m=re.match(r"^(\S+) (\S+) #\d \S+ (.*?): .*K1 ([\d.]+)",l)
    if m: d[(m.group(2),m.group(3),m.group(1))].append(float(m.group(4)))
for k in sorted(d): print(k[0],k[1],k[2],d[k],"median",st.median(d[k]))
PY
