#!/bin/bash
# round 5, session 8: the GPU test suite on the round-5 build (creation-time level-0 choice, 4 waves per workgroup in the 8-slot
# temporal kernel, foveated variant without axis clamps, one bench step path), then bench lines: 4K, 1080p, frames-sharded, configs[3]
R=$(pwd); OUT=$R/gpurun_out/r5s8; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python bench.py --no-cpu-baseline --no-h2d > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
python bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_fhd.json 2>> $OUT/bench.err
FVVDP_DEBUG_VARIANT=1 python tools/gpu_config4.py > $OUT/config4.txt 2>&1
grep -E "^config4|^Q_per_ch|^kernel us|CSF query|candidates" $OUT/config4.txt | sort | uniq -c | head -20
