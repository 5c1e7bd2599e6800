#!/bin/bash
O=gpurun_out/r6c6; mkdir -p $O
timeout 900 python -m pytest -q -x tests/test_ops_gpu.py -k "ws_linear" > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/lab/stream_rate.py 2>&1 | grep -v amdgpu | grep "wslin\|gemm8\|xslin\|torch add" | tee $O/stream_rate.txt
