#!/bin/bash
# PMC passes for one GEMM shape (run via gpurun). usage: pmc_gemm.sh tag mode M N K cin
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/one_gemm.py "$@" > $OUT/log$i.txt 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{k:28s} n={len(v)} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}")
PY
done
