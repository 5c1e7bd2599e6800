#!/bin/bash
# HBM traffic of the dominant kernel family over exactly bench.py's step mix, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside them), FETCH_SIZE doubled on
# gfx950 for wide coalesced reads (it tallies 128-B requests at 64 B), KB -> bytes x1024.  `bench.py --pmc-pass` launches
# the K timed steps eagerly between two marker kernels (delay_kernel); only the GEMM-family rows between the markers are
# summed.  Run via gpurun: bash tools/pmc_bench.sh <tag> [steps]; writes gpurun_out/pmc_bench_<tag>/traffic.json
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench_$1
STEPS=${2:-8}
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 bench.py --pmc-pass --steps $STEPS > $OUT/$C.log 2>&1
done
python3 - $OUT $STEPS <<'PY'
import csv, glob, json, sys
out, steps = sys.argv[1], int(sys.argv[2])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "delay_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 2, f"{c}: marker kernels not found ({len(marks)})"
    tot, n = 0.0, 0
    per = {}
    for r in rows[marks[-2] + 1:marks[-1]]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "splitk_reduce" in name:      # second kernel of a split-K call: its bytes belong to that call
            tot += float(r["Counter_Value"])
        elif "gemm" in name or "xslin_kernel" in name:  # one row per mvoc_gemm_f16 / mvoc_xs_linear_f16 call = one "launch" of bench.py's roofline leg
            tot += float(r["Counter_Value"]); n += 1
        else:
            continue
        e = per.setdefault(name.split("(")[0], [0, 0.0])
        e[0] += 1; e[1] += float(r["Counter_Value"])
    res[c] = {"sum_kb": tot, "launches": n, "per": per}
fetch = res["FETCH_SIZE"]["sum_kb"] * 1024 * 2   # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
write = res["WRITE_SIZE"]["sum_kb"] * 1024
n = res["FETCH_SIZE"]["launches"]
assert n == res["WRITE_SIZE"]["launches"], res
try:
    digest = open("mvoc_amd/libmvoc_hip.so.stamp").read().strip()
except OSError:
    digest = None
j = {"kernel": "implicit-GEMM family (gemm8_kernel / gemm_glds_kernel / gemm_kernel / xslin_kernel instantiations + splitk_reduce_kernel)",
     "steps": steps, "mix": "3 inversion : 1 composition", "launches": n, "launches_per_step": n / steps,
     "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n, "hbm_bytes_per_launch": (fetch + write) / n,
     "lib_digest": digest,
     # which instantiations carry the traffic: launches and corrected bytes per launch of every kernel of the family
     "by_kernel": {k: {"launches": v[0], "fetch_bytes_per_launch": round(v[1] * 2048 / v[0]),
                       "write_bytes_per_launch": round(res["WRITE_SIZE"]["per"].get(k, [1, 0.0])[1] * 1024 / max(res["WRITE_SIZE"]["per"].get(k, [1, 0.0])[0], 1))}
                   for k, v in sorted(res["FETCH_SIZE"]["per"].items(), key=lambda kv: -kv[1][1])},
     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --pmc-pass` (the timed step mix, eager "
             "launches, rows between the two marker kernels only); FETCH_SIZE x2 per the gfx950 correction; KB -> bytes x1024"}
json.dump(j, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(j))
PY
