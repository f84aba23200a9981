#!/bin/bash
# A/B builds of ONE temporal translation unit: recompile temporal_launch.hip for K1_PART=<part> with extra flags and link it with the
# other objects of the current build -> build_variants/<name>.so   (use with FVVDP_LIB=build_variants/<name>.so)
#   tools/build_variant_part.sh <name> <part 0..3> "<extra hipcc flags>"      (part 3 = the planar-YUV kernels)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/fovvideovdp_amd/csrc
mkdir -p $R/build_variants/obj
NAME=$1; PART=$2; FLAGS=$3
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -I$R/include -I$C"
/opt/rocm/bin/hipcc $HF $FLAGS -DK1_PART=$PART -c $C/temporal_launch.hip -o $R/build_variants/obj/$NAME.part$PART.o
PARTS=""
for k in 0 1 2 3; do
  if [ $k = $PART ]; then PARTS="$PARTS $R/build_variants/obj/$NAME.part$k.o"; else PARTS="$PARTS $C/_build/temporal_part$k.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $C/_build/fvvdp_hip.o $PARTS -o $R/build_variants/$NAME.so
echo built $R/build_variants/$NAME.so
