#!/bin/bash
# round 6, last GPU call: the final tree -- GPU suite, smoke, PMC traffic of the GEMM family with THIS library digest, the metric line
O=gpurun_out/r6final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
bash tools/pmc_bench.sh r6final 4 > $O/pmc.log 2>&1; cp gpurun_out/pmc_bench_r6final/traffic.json $O/pmc_gemm_traffic.json
mkdir -p profiles/r6; cp $O/pmc_gemm_traffic.json profiles/r6/pmc_gemm_traffic.json
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json
b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=b['roofline']
print(b['value'], r['frac'], r['traffic'], r['traffic_source'])"
