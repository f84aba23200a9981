#!/bin/bash
# A/B builds: recompile the band / pooling translation unit (fvvdp_hip.hip, ~10 s) with extra flags and link it with the
# temporal-kernel objects of the current build -> build_variants/<name>.so   (use with FVVDP_LIB=build_variants/<name>.so)
#   tools/build_variant.sh <name> "<extra hipcc flags>" [temporal]     (3rd argument: also recompile the temporal parts)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/fovvideovdp_amd/csrc
mkdir -p $R/build_variants/obj
NAME=$1; FLAGS=$2
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -I$R/include -I$C"
/opt/rocm/bin/hipcc $HF $FLAGS -c $C/fvvdp_hip.hip -o $R/build_variants/obj/$NAME.main.o
PARTS=""
for k in 0 1 2 3; do
  if [ "${3:-}" = "temporal" ]; then
    /opt/rocm/bin/hipcc $HF $FLAGS -DK1_PART=$k -c $C/temporal_launch.hip -o $R/build_variants/obj/$NAME.part$k.o &
    PARTS="$PARTS $R/build_variants/obj/$NAME.part$k.o"
  else
    PARTS="$PARTS $C/_build/temporal_part$k.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $R/build_variants/obj/$NAME.main.o $PARTS -o $R/build_variants/$NAME.so
echo built $R/build_variants/$NAME.so
