#!/bin/bash
# round 6, GPU call 5: the weight-stationary 320 -> 320 projection kernel -- exactness tests, micro-benchmark, step A/B
O=gpurun_out/r6c5; mkdir -p $O
timeout 900 python -m pytest -q -x tests/test_ops_gpu.py -k "ws_linear or xs_linear or groupnorm_folded" > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/lab/stream_rate.py 2>&1 | grep -v amdgpu | tee $O/stream_rate.txt
run() { local label=$1; shift
  for mix in comp inv; do
    env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done; }
{ for rep in 1 2 3; do run ws1 MVOC_WS=1; run ws0 MVOC_WS=0; done; } 2>&1 | tee $O/ws_ab.txt
timeout 900 python -m pytest -q -x tests/test_fullwidth_gpu.py tests/test_unet_gpu.py > $O/tests_net.log 2>&1; tail -3 $O/tests_net.log
