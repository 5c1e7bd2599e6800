#!/bin/bash
# HBM traffic of the dominant kernel during bench.py steps, collected as MI355X_MICROARCH.md prescribes: separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), FETCH_SIZE doubled on gfx950 for wide coalesced
# reads (it tallies 128-B requests at 64 B).  Run via gpurun; writes gpurun_out/pmc_bench_<tag>/traffic.json
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench_$1
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 bench.py --no-cpu-baseline --no-roofline --no-graphs --steps 4 --warmup 0 > $OUT/$C.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); n += 1
    res[c] = {"sum_kb": tot, "launches": n}
fetch = res["FETCH_SIZE"]["sum_kb"] * 1024 * 2   # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
write = res["WRITE_SIZE"]["sum_kb"] * 1024
n = res["FETCH_SIZE"]["launches"]
j = {"kernel": "gemm_glds_kernel/gemm_kernel (all instantiations)", "launches": n,
     "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n, "hbm_bytes_per_launch": (fetch + write) / n,
     "note": "whole bench.py process (priming + 4 eager steps); FETCH_SIZE x2 per the gfx950 correction; KB->bytes x1024"}
json.dump(j, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(j))
PY
