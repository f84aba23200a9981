#!/bin/bash
# Produce the profile bundle kept under profiles/ (run on the GPU box through gpurun; writes gpurun_out/prof_bundle/).
#   1. bench.py unprofiled                       2. rocprofv3 --kernel-trace of bench.py (summary table)
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the band kernels -> level-0 HBM traffic per launch
#   4. kernel traces of the YUV ingest, PU21-PSNR and heat-map probes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_bundle
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-measure-traffic > $OUT/bench_profiled.json 2> /tmp/kt.err
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) --band-levels 7 --dispatches band2_kernel > $OUT/kernel_trace_bench.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- python $R/tools/gpu_bandonly.py > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- python $R/tools/gpu_bandonly.py > /tmp/pw.log 2>&1
python $R/tools/pmc_level0.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $OUT/pmc_level0.json > /dev/null
# the same two passes for the one-level kernels (FVVDP_BAND_FUSE=0), and their kernel trace: the A/B of the two-level kernel on this box
FVVDP_BAND_FUSE=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf0 -o f -- python $R/tools/gpu_bandonly.py > /tmp/pf0.log 2>&1
FVVDP_BAND_FUSE=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw0 -o w -- python $R/tools/gpu_bandonly.py > /tmp/pw0.log 2>&1
python $R/tools/pmc_level0.py $(find /tmp/pf0 -name "*.db" | head -1) $(find /tmp/pw0 -name "*.db" | head -1) $OUT/pmc_level0_onelevel.json > /dev/null
FVVDP_BAND_FUSE=0 rocprofv3 --kernel-trace --stats -d /tmp/kt0 -o bench0 -- python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_profiled_onelevel.json 2> /tmp/kt0.err
python $R/tools/rocpd_summary.py $(find /tmp/kt0 -name "*.db" | head -1) --band-levels 7 > $OUT/kernel_trace_bench_onelevel.md
python $R/tools/gpu_bandonly_speed.py 12 > $OUT/bandonly_fused.txt 2>/dev/null
FVVDP_BAND_FUSE=0 python $R/tools/gpu_bandonly_speed.py 12 > $OUT/bandonly_onelevel.txt 2>/dev/null
$R/build_variants/mix > $OUT/mix.txt 2>&1
python $R/tools/gpu_parity_report.py > $OUT/parity.md 2>/dev/null
( python $R/tools/gpu_g9_report.py; python $R/tools/gpu_g10_report.py; python $R/tools/gpu_g11_report.py; python $R/tools/gpu_g12_report.py ) 2>/dev/null | grep -v Warn > $OUT/parity_goldens.txt
python $R/tools/gpu_fps.py 30:60:u8 60:120:u8 120:120:u8 144:120:u8 240:120:u8 30:60:u16 60:60:u16 120:60:u16 144:60:u16 30:60:f32rgb 120:60:f32rgb 30:60:f32gray 2>/dev/null | grep -v Warn > $OUT/fps_probe.txt
python $R/tools/gpu_k1_ab.py 30:60:u8:5 60:60:u8:5 120:60:u8:5 2>/dev/null | grep -v Warn >> $OUT/fps_probe.txt
# round 6: the same probe behind the PQ display model (16-bit / float frames: closed-form display model on (test, reference) pairs)
echo "PROBE_DISPLAY=standard_hdr_pq:" >> $OUT/fps_probe.txt
PROBE_DISPLAY=standard_hdr_pq python $R/tools/gpu_fps.py 30:60:u8 30:60:u16 60:60:u16 120:120:u16 240:120:u16 30:60:f32rgb 120:120:f32rgb 2>/dev/null | grep -v Warn >> $OUT/fps_probe.txt
BATCHES=None,60 python $R/tools/gpu_feeder.py 2>/dev/null | cut -c1-110 > $OUT/feeder_probe.txt
HH=2160 WW=3840 BATCHES=None,60 python $R/tools/gpu_feeder.py 2>/dev/null | cut -c1-110 >> $OUT/feeder_probe.txt
python $R/bench.py --pairs-per-gpu 8 --no-cpu-baseline --no-h2d > $OUT/bench_pairs8.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/kr -o rates -- python $R/tools/gpu_fps.py 60:120:u8 120:120:u8 144:120:u8 240:120:u8 30:60:u16 > /dev/null 2> /tmp/kr.err
python $R/tools/rocpd_summary.py $(find /tmp/kr -name "*.db" | head -1) --only temporal_vec > $OUT/kernel_trace_rates.md
# K1 (temporal kernel): HBM traffic per 60-frame launch from the same two counters (algorithmic: 67 frames read x 49.8 MB + 60 x 132.7 MB written)
STAGE=all REPS=2 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf1 -o f -- python $R/tools/gpu_bandonly.py > /tmp/pf1.log 2>&1
STAGE=all REPS=2 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw1 -o w -- python $R/tools/gpu_bandonly.py > /tmp/pw1.log 2>&1
( python $R/tools/pmc_query.py $(find /tmp/pf1 -name "*.db" | head -1) temporal_vec_kernel; python $R/tools/pmc_query.py $(find /tmp/pw1 -name "*.db" | head -1) temporal_vec_kernel ) > $OUT/pmc_k1.txt
rocprofv3 --kernel-trace --stats -d /tmp/ky -o yuv -- python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 > $OUT/yuv_probe.txt 2> /tmp/ky.err
python $R/tools/rocpd_summary.py $(find /tmp/ky -name "*.db" | head -1) --only temporal_yuv > $OUT/kernel_trace_yuv.md
echo "PROBE_DISPLAY=standard_hdr_pq:" >> $OUT/yuv_probe.txt
PROBE_DISPLAY=standard_hdr_pq python $R/tools/gpu_yuv.py 2160x3840x60:10:420 2160x3840x60:10:420:60 2>/dev/null | grep -v Warn >> $OUT/yuv_probe.txt
rocprofv3 --kernel-trace --stats -d /tmp/kp -o psnr -- python $R/tools/gpu_psnr.py > $OUT/psnr_probe.txt 2> /tmp/kp.err
python $R/tools/rocpd_summary.py $(find /tmp/kp -name "*.db" | head -1) --only pu21 > $OUT/kernel_trace_psnr.md
rocprofv3 --kernel-trace --stats -d /tmp/kh -o heat -- python $R/tools/gpu_heatprof.py 12 threshold > /tmp/kh.out 2> /tmp/kh.err
grep "^total" /tmp/kh.out > $OUT/heat_probe.txt
python $R/tools/rocpd_summary.py $(find /tmp/kh -name "*.db" | head -1) --band-levels 7 > /tmp/kh.md
grep -E "kernel \||---|colour_|heat_level|band_kernel<4, true, 0>" /tmp/kh.md > $OUT/kernel_trace_heat.md
rocprofv3 --kernel-trace --stats -d /tmp/kf -o fov -- python $R/tools/gpu_config4.py > /tmp/kf.out 2> /tmp/kf.err
grep -E "^config4|^Q_per_ch|^kernel us" /tmp/kf.out > $OUT/fov_probe.txt
python $R/tools/rocpd_summary.py $(find /tmp/kf -name "*.db" | head -1) --band-levels 7 > /tmp/kf.md
grep -E "kernel \||---|band_kernel<4, false, 1>|temporal_vec" /tmp/kf.md > $OUT/kernel_trace_fov.md
# SQ / TCC counters of the dominant pyramid kernel and of the foveated kernel: what bounds them
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"
for T in bandonly fov_bandonly; do
  rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/s1_$T -o a -- python $R/tools/gpu_$T.py > /tmp/s1_$T.log 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace -d /tmp/s2_$T -o a -- python $R/tools/gpu_$T.py > /tmp/s2_$T.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/s3_$T -o a -- python $R/tools/gpu_$T.py > /tmp/s3_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/s4_$T -o a -- python $R/tools/gpu_$T.py > /tmp/s4_$T.log 2>&1
  python $R/tools/pmc_sq_summary.py band $(find /tmp/s1_$T /tmp/s2_$T /tmp/s3_$T /tmp/s4_$T -name "*.db") > $OUT/pmc_sq_$T.md 2>/dev/null
done
python $R/tools/codeobj_report.py > $OUT/codeobj.md 2>/dev/null
# BASELINE configs[1]: 1920x1080 x60 (standard_fhd, 6 bands): kernel table + the bench line
rocprofv3 --kernel-trace --stats -d /tmp/kfhd -o fhd -- python $R/bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_fhd.json 2> /tmp/kfhd.err
python $R/tools/rocpd_summary.py $(find /tmp/kfhd -name "*.db" | head -1) --band-levels 6 > $OUT/kernel_trace_fhd.md
python $R/bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_fhd_plain.json 2>/dev/null
# the same allocation A/B as profiles/r04_level0_chunks.md on this box
FVVDP_ALLOC=malloc FVVDP_PLACEMENT_PROBE=0 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_malloc.json 2>/dev/null
FVVDP_PLACEMENT_PROBE=0 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_chunks.json 2>/dev/null
FVVDP_BAND_INRANGE=0 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_clamps.json 2>/dev/null
[ -f $R/build_variants/k1wpb1.so ] && FVVDP_LIB=$R/build_variants/k1wpb1.so python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_k1wpb1.json 2>/dev/null
FVVDP_BAND2_TICKET=0 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_noticket.json 2>/dev/null
python $R/bench.py --shard frames --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_frames.json 2>/dev/null
# round 6: the one-rank collective off / forced (the default is auto = forced where RCCL initialises), configs[3] with its roofline_fov block
python $R/bench.py --collective off --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_collective_off.json 2>/dev/null
python $R/bench.py --collective force --no-cpu-baseline --no-h2d --no-measure-traffic > $OUT/bench_collective_force.json 2>/dev/null
python $R/bench.py --config 3 --no-cpu-baseline --no-h2d > $OUT/bench_config3.json 2>/dev/null
python $R/tools/gpu_rccl_one_rank.py > $OUT/rccl_one_rank.txt 2>/dev/null
ls -la $OUT
