P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
echo before; python bench.py $B 2>/dev/null | python -c "$P"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
echo after; for i in 1 2 3; do python bench.py $B 2>/dev/null | python -c "$P"; done
echo malloc; FVVDP_ALLOC=malloc python bench.py $B 2>/dev/null | python -c "$P"
echo chunk2; FVVDP_VMM_CHUNK_MB=2 python bench.py $B 2>/dev/null | python -c "$P"
