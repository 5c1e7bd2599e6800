#!/bin/bash
# MFMA utilisation / wait / LDS counters of ONE kernel family inside any script (run via gpurun):
#   bash tools/pmc_kernel.sh <tag> <kernel-name substring> <script.py> [args..]
# Separate --pmc passes (counters that do not fit one pass; no trace domains beside them), program directly behind `--`.
# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
# (rocprofv3 reports the sum over the 8 XCDs; MI355X_MICROARCH.md "DVFS give-back" / PMC units).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; KERN=$2; shift 2
OUT=gpurun_out/pmck_$TAG
mkdir -p $OUT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/log$i.txt 2>&1
done
python3 - $OUT "$KERN" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in kern.split("|")):  # <kernel-name substring>: alternatives separated by |
            agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
grand = collections.defaultdict(float)
lib = [n for n in agg if not any(t in n for t in ("at::native", "rocclr", "rocblas", "delay_kernel"))]  # libmvoc_hip's own kernels
for name in lib:
    for k, v in agg[name].items():
        grand[k] += sum(v)
if len(lib) > 1 and grand.get("GRBM_GUI_ACTIVE"):
    cyc = grand["GRBM_GUI_ACTIVE"] / 8
    print(f"== ALL {len(lib)} matching library kernels together (weighted by their cycles; torch's own fill / copy / init kernels left out): MFMA utilisation "
          f"{100 * grand['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * cyc):.1f} % | waves waiting "
          f"{100 * grand.get('SQ_WAIT_ANY', 0) / max(grand.get('SQ_WAVE_CYCLES', 0), 1):.1f} %")
for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    print(f"== {name}  ({n} dispatches)")
    tot = {k: sum(v) for k, v in c.items()}
    for k in sorted(tot):
        print(f"   {k:28s} sum {tot[k]:.4g}")
    if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
        cyc = tot["GRBM_GUI_ACTIVE"] / 8
        print(f"   MFMA utilisation = {100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * cyc):.1f} % of (4 SIMD x 256 CU x kernel cycles)")
        if "SQ_WAVE_CYCLES" in tot and tot["SQ_WAVE_CYCLES"] > 0:
            print(f"   waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) = {100 * tot.get('SQ_WAIT_ANY', 0) / tot['SQ_WAVE_CYCLES']:.1f} %")
PY
rm -rf $OUT/p1 $OUT/p2   # the per-dispatch counter CSVs are large: keep the summary and the logs
