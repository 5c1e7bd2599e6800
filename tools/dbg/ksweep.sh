for k in 512 1024 2048 4096 8192; do ./tools/lab/g8_lab 327680 320 $k 2>&1 | grep -E "^==|median" ; done
for k in 512 1024 2048 4096 8192; do ./tools/lab/g8_lab 81920 1280 $k 2>&1 | grep -E "^==|median" ; done
