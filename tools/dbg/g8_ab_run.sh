#!/bin/bash
# GPU box: phase stamps and per-shape tables, old library against the working tree, on ONE box (tools/dbg/build_ab.sh first)
OUT=${1:-gpurun_out/ab}
mkdir -p $OUT
S=./tools/lab/g8_stamps
for lib in libmvoc_g8dbg_prev.so libmvoc_g8dbg.so; do
  echo "== $lib"
  export LD_PRELOAD=$PWD/tools/lab/$lib
  $S 81920 640 640 81 1 8
  $S 81920 640 5760 81 1 8
  $S 327680 320 2880 82 1 8
  $S 327680 320 960 82 1 8
  $S 81920 1920 640 81 0 2
  $S 81920 5120 640 81 0 3
  $S 16384 640 1920 81 1 8
  unset LD_PRELOAD
done > $OUT/stamps.txt 2>&1
for b in 1 5; do
  MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_old.so python tools/gemm_bench.py $b 0 > $OUT/gemm_old_B$b.txt 2>&1; echo "OLD B=$b $(tail -1 $OUT/gemm_old_B$b.txt)"
  python tools/gemm_bench.py $b 0 > $OUT/gemm_new_B$b.txt 2>&1; echo "NEW B=$b $(tail -1 $OUT/gemm_new_B$b.txt)"
done
