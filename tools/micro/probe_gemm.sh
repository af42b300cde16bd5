#!/bin/bash
# Builds instrumented copies of libsuhpe.so (SUHPE_PROBE=1..4, see conv_igemm.hip) next to this script: tools/micro/libsuhpe_probeN.so
# and prints nothing else; tools/micro/run_probe_gemm.py times the persistent pointwise GEMM with each of them.
set -e
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../semiuhpe_amd/csrc
for n in ${PROBES:-1 2 3 4}; do
  objs=""
  for f in common head nn_elem layout_aug dwconv attention se_fused; do objs="$objs $csrc/build/$f.o"; done
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DSUHPE_PROBE=$( [ $n = 5 ] && echo 0 || echo $n ) $( [ $n = 5 ] && echo -DSUHPE_STAGES=3 ) -I$csrc/../../include -c $csrc/conv_igemm.hip -o $here/conv_probe$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $here/libsuhpe_probe$n.so $objs $here/conv_probe$n.o
  rm -f $here/conv_probe$n.o
done
