S=./tools/lab/g8_stamps
export LD_PRELOAD=$PWD/tools/lab/libmvoc_g8dbg.so
$S 327680 320 960 82 0 8
$S 327680 320 960 82 1 8
$S 327680 320 2880 82 1 8
$S 327680 320 1280 82 1 0
unset LD_PRELOAD
