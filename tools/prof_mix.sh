#!/bin/bash
# rocprofv3 kernel-trace statistics over EXACTLY the timed step mix (run via gpurun): `bench.py --pmc-pass --steps K` launches
# the K steps of the 3 : 1 mix eagerly once (no priming passes, no per-kind timing), so the family averages of this profile are
# over the same launches as bench.py's roofline leg.  usage: bash tools/prof_mix.sh <tag> [steps] [mix: job | inv | comp]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profmix_$1
STEPS=${2:-8}
MIX=${3:-job}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --pmc-pass --steps $STEPS --mix $MIX > $OUT/bench.log 2>&1
f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace*.csv" -delete
python3 - $OUT/kernel_stats.csv $STEPS $MIX <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps, mix = int(sys.argv[2]), sys.argv[3]
rows = [r for r in rows if "at::native" not in r["Name"] and "delay_kernel" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
g = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in ("gemm", "xslin_kernel", "splitk")))
n = sum(int(r["Calls"]) for r in rows if ("gemm" in r["Name"] or "xslin_kernel" in r["Name"]) and "splitk" not in r["Name"])
print(f"steps {steps} (mix {mix}: job = 3 inversion : 1 composition), library kernels total {tot/1e6:.1f} ms = {tot/1e6/steps:.2f} ms per step")
print(f"implicit-GEMM family: {g/1e6:.1f} ms = {100*g/tot:.1f} % of GPU time, {n} launches ({n/steps:.1f} per step), average launch {g/n/1e3:.2f} us")
print()
print(f"{'kernel':76s} {'calls':>6s} {'total ms':>9s} {'%':>6s} {'avg us':>8s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print(f"{name[:76]:76s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f} {float(r['AverageNs'])/1e3:8.1f}")
PY
