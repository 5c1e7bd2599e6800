#!/bin/bash
# GPU box: end-to-end bench, round-3 tree (tools/lab/r3_tree, `git worktree add` + build on the build machine) against the working
# tree, interleaved on ONE box.  usage: bash tools/dbg/e2e_ab.sh <outdir> [rounds]
OUT=${1:-gpurun_out/e2e}; N=${2:-3}
mkdir -p $OUT
for i in $(seq 1 $N); do
  (cd tools/lab/r3_tree && python bench.py --steps 8 --warmup 4 2>/dev/null) > $OUT/old_$i.json
  python bench.py --steps 8 --warmup 4 2>/dev/null > $OUT/new_$i.json
done
python - $OUT $N <<'PY'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for tag in ("old", "new"):
    for i in range(1, n + 1):
        d = json.load(open(f"{out}/{tag}_{i}.json")); c = d["config"]; r = d["roofline"]
        print(tag, i, d["value"], "inv", c.get("inversion_step_ms"), "comp", c.get("composition_step_ms"), "gemm TF", r.get("achieved"), "fam ms", r.get("by_family_ms"))
PY
