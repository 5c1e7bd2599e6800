#!/bin/bash
# Per-shape counter table of the implicit-GEMM family (run via gpurun): bash tools/pmc_shapes.sh <tag> [B] [top] [reps]
# One un-profiled timing pass + separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not share a pass; no trace domains
# beside them; the program directly behind `--`) over tools/gemm_shapes_pmc.py; the dispatches between two marker kernels are one
# shape.  FETCH_SIZE x2 (gfx950 tallies a 128-B request of a wide coalesced read at 64 B), KB -> bytes x1024; MFMA utilisation =
# SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE / 8); held clock = (GRBM_GUI_ACTIVE / 8) / dispatch duration of the
# same pass.  Writes gpurun_out/pmcs_<tag>/table.txt (+ table.json).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; B=${2:-5}; TOP=${3:-20}; REPS=${4:-3}
OUT=gpurun_out/pmcs_$TAG
mkdir -p $OUT
MVOC_PMC_PASS=time python3 tools/gemm_shapes_pmc.py $OUT $B $TOP $REPS > $OUT/log_time.txt 2>&1
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  export MVOC_PMC_PASS=p$i
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/gemm_shapes_pmc.py $OUT $B $TOP $REPS > $OUT/log$i.txt 2>&1
done
python3 - $OUT <<'PY' | tee $OUT/table.txt
import csv, glob, json, sys, collections
out = sys.argv[1]
man = json.load(open(f"{out}/manifest_time.json"))
per = [collections.defaultdict(float) for _ in man]
dur = [0.0 for _ in man]
names = [collections.Counter() for _ in man]
for p in sorted(glob.glob(f"{out}/p*/")):
    f = glob.glob(f"{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file in", p); continue
    rows = list(csv.DictReader(open(f[0])))
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    ids = sorted(disp)
    marks = [i for i in ids if "delay_kernel" in disp[i][0]["Kernel_Name"]]
    marks = marks[-2 * len(man):]  # (opening, closing) per shape; what lies between a closing and the next opening is warm-up
    assert len(marks) == 2 * len(man), (p, len(marks), len(man))
    for s in range(len(man)):
        for i in ids:
            if marks[2 * s] < i < marks[2 * s + 1]:
                r0 = disp[i][0]
                if "at::native" in r0["Kernel_Name"]:
                    continue
                for r in disp[i]:
                    per[s][r["Counter_Name"]] += float(r["Counter_Value"])
                if "GRBM_GUI_ACTIVE" in [r["Counter_Name"] for r in disp[i]]:
                    dur[s] += (int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) * 1e-9
                    names[s][r0["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] += 1
tab = []
print(f"{'mode':4}{'ko':>2} {'M':>7} {'N':>6} {'K':>6} {'cin':>5} act res cnt | {'us':>7} {'TF/s':>5} | {'alg MB':>7} {'fetch':>7} {'write':>7} {'ratio':>5} | {'MFMA%':>5} {'wait%':>5} {'GHz':>5} {'L2hit%':>6} | kernel")
for s, m in enumerate(man):
    c, reps = per[s], m["reps"]
    fetch = c.get("FETCH_SIZE", 0) * 2048 / reps
    write = c.get("WRITE_SIZE", 0) * 1024 / reps
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * cyc) if cyc else 0
    wait = 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
    ghz = cyc / dur[s] / 1e9 if dur[s] else 0
    hit = 100 * c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    kn = "+".join(k for k, _ in names[s].most_common(2))
    row = dict(m, fetch_bytes=fetch, write_bytes=write, traffic_ratio=(fetch + write) / m["alg_bytes"], mfma_busy_pct=busy, wait_pct=wait, held_ghz=ghz, l2_hit_pct=hit, kernels=kn,
               tflops=m["flop"] / m["us"] / 1e6)
    tab.append(row)
    print(f"{m['mode']:4d}{m.get('korder', 0):2d} {m['M']:7d} {m['N']:6d} {m['K']:6d} {m['cin']:5d} {m['act']:3d} {int(m['resid']):3d} {m['count']:3d} | {m['us']:7.1f} {row['tflops']:5.0f} | "
          f"{m['alg_bytes'] / 1e6:7.1f} {fetch / 1e6:7.1f} {write / 1e6:7.1f} {row['traffic_ratio']:5.2f} | {busy:5.1f} {wait:5.1f} {ghz:5.2f} {hit:6.1f} | {kn}")
tw = sum(r["us"] * r["count"] for r in tab)
print(f"time-weighted over these {len(tab)} shapes ({tw / 1e3:.1f} ms per forward): traffic ratio "
      f"{sum((r['fetch_bytes'] + r['write_bytes']) * r['count'] for r in tab) / sum(r['alg_bytes'] * r['count'] for r in tab):.2f}, "
      f"MFMA busy {sum(r['mfma_busy_pct'] * r['us'] * r['count'] for r in tab) / tw:.1f} %")
json.dump(tab, open(f"{out}/table.json", "w"), indent=0)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
