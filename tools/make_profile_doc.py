#!/usr/bin/env python3
"""Turn gpurun_out/prof_bundle/ (written by tools/collect_profiles.sh on the GPU box) into the tracked files
profiles/<round>_final_kernel_trace.md and profiles/<round>_pmc_level0.json.   usage: make_profile_doc.py r01"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "gpurun_out", "prof_bundle")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rd = lambda n: open(os.path.join(B, n)).read().strip() if os.path.exists(os.path.join(B, n)) else "(not collected)"
kt = [l for l in rd("kernel_trace_bench.md").split("\n")
      if ("temporal_" in l or "band_kernel" in l or "finalize" in l or l.startswith("| kernel") or l.startswith("|---"))]
doc = f"""# Round {tag[1:]} — rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline` (final build of the round)

Produced by `tools/collect_profiles.sh` in ONE gpurun call (one box) and formatted by `tools/make_profile_doc.py`.
Workload: 3840x2160 x60 uint8 RGB pair, standard_4k, 30 fps (8 taps), non-foveated; 2 warm-up + 5 timed steps + the
in-library HIP-event timing pass.  band_kernel rows are labelled by pyramid level (dispatch order).  torch kernels of
the synthetic input generator (setup, untimed) are omitted.

{chr(10).join(kt)}

bench.py JSON of the SAME profiled process (HIP events on the kernels' stream):

```
{rd('bench_profiled.json')}
```

bench.py JSON of the unprofiled run on the same box, right before:

```
{rd('bench_plain.json')}
```

## HBM traffic of the dominant kernel (separate `--pmc` passes, `tools/gpu_bandonly.py`)

`profiles/{tag}_pmc_level0.json` (same call): FETCH_SIZE x2 + WRITE_SIZE per level-0 launch vs the algorithmic bytes.

```
{rd('pmc_level0.json')}
```

K1 (`temporal_vec_kernel<8,4,0>`), same counters over whole `predict` calls (KiB per dispatch).  Algorithmic per 60-frame
launch: 60 x 132.7 MB = 7.96 GB written -- WRITE_SIZE agrees to 4 digits; 67 source frames x 49.8 MB = 3.33 GB read -- the
raw FETCH_SIZE is 1.49 GB, i.e. the counter's scale for these 4 B/lane loads is 2.23, not the 2.0 that MI355X_MICROARCH.md
calibrates for 16 B/lane streaming reads ("other access widths are uncalibrated"); every source byte is read exactly once
by construction (one lane owns its pixels for the whole launch):

```
{rd('pmc_k1.txt')}
```

## "Next"-row kernels (SURVEY section 8(f)) and the foveated configuration, same call

YUV ingest (`tools/gpu_yuv.py`: 4K x60, 4:2:0 8 bit 30 fps / 4:2:0 10 bit 60 fps / 4:4:4 8 bit 30 fps; one launch = 60 output frames):

{rd('kernel_trace_yuv.md')}

```
{rd('yuv_probe.txt')}
```

PU21-PSNR (`tools/gpu_psnr.py`: 4K x60 uint8, 4K x20 fp32):

{rd('kernel_trace_psnr.md')}

```
{rd('psnr_probe.txt')}
```

Coloured heat maps (`tools/gpu_heatprof.py 12 threshold`: 4K, 12 frames in batches of 5; `band_kernel<4, true, 0>` is
the map-writing variant of the pyramid kernel):

{rd('kernel_trace_heat.md')}

```
{rd('heat_probe.txt')}
```

BASELINE configs[3] (`tools/gpu_config4.py`: 4K x120, foveated, moving gaze, PQ display; `band_kernel<4, false, 1>` is the
foveated variant with the band's LUT slice in LDS):

{rd('kernel_trace_fov.md')}

```
{rd('fov_probe.txt')}
```
"""
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_final_kernel_trace.md"), "w").write(doc)
if os.path.exists(os.path.join(B, "pmc_level0.json")):
    shutil.copy(os.path.join(B, "pmc_level0.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_level0.json"))
print("wrote profiles/%s_final_kernel_trace.md (%d bytes)" % (tag, len(doc)))
