#!/bin/bash
# round 5, session 42: the tile kernel for small levels -- parity with the streaming kernels, then the suite with it forced everywhere
cd $(pwd)
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "tile_kernel" 2>&1 | tail -15
FVVDP_BAND_TILE=100000000 timeout 1800 python -m pytest tests -m gpu -q -k "not tile_kernel" > gpurun_out/r5s42_suite.log 2>&1; tail -12 gpurun_out/r5s42_suite.log
