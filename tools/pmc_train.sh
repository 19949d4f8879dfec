#!/bin/bash
# PMC counters of the training kernels (forward with tape, input gradients, weight gradients), one rocprofv3 run per
# counter set.  usage (on the GPU box, from the repo root): bash tools/pmc_train.sh <outdir> [bench.py args]
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "nerf::" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0)):
        fo.write(k + "\n")
        for c in sorted(agg[k]):
            fo.write("  %-32s %18.0f per launch (%d launches)\n" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
print(open(out + "/summary.txt").read())
PY
