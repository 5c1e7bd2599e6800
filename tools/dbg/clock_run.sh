#!/bin/bash
# GPU box: held clock + MFMA slot fill of the gemm8 K loop on production shapes (stamped library: tools/lab/README.md), after a
# warm-up long enough for the chip to settle its clock
OUT=${1:-gpurun_out/clock}; mkdir -p $OUT
S=./tools/lab/g8_stamps
export LD_PRELOAD=$PWD/tools/lab/libmvoc_g8dbg.so
{ for i in 1 2 3; do $S 81920 1280 11520 81 0 8 > /dev/null; done
  $S 81920 1280 11520 81 0 8; $S 81920 640 5760 81 1 8; $S 327680 320 2880 82 1 8; $S 81920 640 640 81 1 8; $S 327680 320 960 82 1 8
  $S 16384 640 1920 81 1 8; $S 65536 320 2880 82 1 8; } > $OUT/g8_clock.txt 2>&1
unset LD_PRELOAD
cat $OUT/g8_clock.txt
