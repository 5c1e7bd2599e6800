#!/bin/bash
# round 6, GPU call 7: the weight-stationary kernel WITHOUT a per-tile barrier (LDS counters) -- tests under a timeout (a lost wake-up would hang), rate, step A/B
O=gpurun_out/r6c7; mkdir -p $O
timeout 300 python -m pytest -q -x tests/test_ops_gpu.py -k "ws_linear" > $O/tests.log 2>&1; echo "rc $?"; tail -3 $O/tests.log
timeout 300 python tools/lab/stream_rate.py 2>&1 | grep -v amdgpu | grep "wslin\|gemm8\|xslin\|torch add" | tee $O/stream_rate.txt
run() { local label=$1; shift
  for mix in comp inv; do
    timeout 600 env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done; }
{ for rep in 1 2; do run ws1 MVOC_WS=1; run ws0 MVOC_WS=0; done; } 2>&1 | tee $O/ws_ab.txt
