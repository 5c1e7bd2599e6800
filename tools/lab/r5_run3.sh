#!/bin/bash
python -m pytest tests/test_ops_gpu.py -q -x -k "subpixel or conv3x3_upsample or upsample" 2>&1 | tail -4
python -m pytest tests/test_fullwidth_gpu.py tests/test_unet_gpu.py -q -x 2>&1 | tail -3
bash tools/lab/ab_bench.sh 2>&1 | grep -v "no-tail\|no-share\|no-fold"
