SH="48,192,1,56,160,32;64,256,1,28,160,32;512,512,1,14,320,64;256,256,1,28,320,64;128,128,2,56,320,64"
for a in 3 4 2 1; do
  lib=semiuhpe_amd/csrc/libsuhpe_ws$a.so; [ $a = 3 ] && lib=semiuhpe_amd/csrc/libsuhpe.so
  echo "== WW_STAGE_AT=$a"
  SUHPE_LIB=$PWD/$lib WINO_SHAPES="$SH" python tools/wino_bench.py 2>&1 | grep -o "n[0-9]*:.*\| wgrad [0-9]* -> [0-9]* us" | grep -o "wgrad [0-9]* -> [0-9]* us" | tr '\n' ' '; echo
done
