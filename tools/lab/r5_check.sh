#!/bin/bash
# round-5 loop: the forward / pipeline GPU tests that cover the day's change, then the metric line
python -m pytest tests/test_unet_gpu.py tests/test_fullwidth_gpu.py tests/test_pipeline_gpu.py -q -x -s 2>&1 | grep -v amdgpu.ids | grep "prune_source_tail\|passed\|failed\|Error\|hinted\|error\|dedupe\|fold" | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/${1:-r5_bench}.json 2> gpurun_out/${1:-r5_bench}.err
python - <<PYEOF
import json
d=json.loads(open("gpurun_out/${1:-r5_bench}.json").read().strip().splitlines()[-1])
c=d["config"]
print("steps/s", d["value"], "seq", (c.get("sequential_inversions") or {}).get("value"), "inv", c["inversion_step_ms"], "inv3", c.get("inversion_step_ms_three_clips_concurrent"), "comp", c["composition_step_ms"])
f=d["roofline"].get("by_family_ms"); print(f, "sum", round(sum(f.values()),1) if f else None, "gemm TF", d["roofline"]["achieved"])
PYEOF
