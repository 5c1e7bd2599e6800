S=./tools/lab/g8_stamps
for lib in libmvoc_g8dbg.so libmvoc_g8dbg_prev.so; do
echo "== $lib"
export LD_PRELOAD=$PWD/tools/lab/$lib
$S 81920 640 5760 81 1 8
$S 81920 1280 2048 81 1 8
$S 327680 320 2880 82 1 8
$S 81920 1920 640 81 0 2
$S 81920 5120 640 81 0 3
done
