#!/bin/bash
# Where a chunk of the Winograd kernel loses its time: variant libraries with parts of the loop removed (WINO_ABLATE bits, conv_wino.hip), timed on a few shapes.
#   bash tools/wino_ablate.sh build        (here: hipcc cross-compiles)      bash tools/wino_ablate.sh run   (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  for a in 1 2 4 6 8 16 7; do python3 -c "from semiuhpe_amd.build import build_variant; print(build_variant('abl$a', 'conv_wino.hip', ['WINO_ABLATE=$a']))"; done
  exit 0
fi
for a in 0 1 4 8; do
  lib=$R/semiuhpe_amd/csrc/libsuhpe_abl$a.so; [ $a = 0 ] && lib=$R/semiuhpe_amd/csrc/libsuhpe.so
  echo "== WINO_ABLATE=$a"
  SUHPE_LIB=$lib python3 $R/tools/wino_time.py
done
