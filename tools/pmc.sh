#!/bin/bash
# Collects PMC counters for the network kernel in separate passes (one rocprofv3 run per counter set).
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <outdir> [mlp_bench args]
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 tools/mlp_bench.py --reps 2 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mlp_" not in k and "mlp16_" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in agg:
        fo.write(k + "\n")
        for c in sorted(agg[k]):
            fo.write("  %-32s %18.0f per launch (%d launches)\n" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
print(open(out + "/summary.txt").read())
PY
