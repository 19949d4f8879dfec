#!/bin/bash
# Profiles `python3 bench.py` on the GPU box: one kernel-trace pass (per-kernel durations) and two PMC passes
# (FETCH_SIZE, WRITE_SIZE: HBM-side bytes of every kernel), each its own rocprofv3 run.
# usage (from the repo root): bash tools/profile_bench.sh <outdir> <precision>
set -u
out=$1; prec=$2
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 3 --warmup 1 --precision "$prec" --no-cpu-baseline --no-extras > "$out/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$out/trace.log"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 bench.py --steps 1 --warmup 1 --precision "$prec" --no-cpu-baseline --no-extras > "$out/pmc_$c.log" 2>&1 || { echo "$c pass failed"; tail -5 "$out/pmc_$c.log"; exit 1; }
done
python3 - "$out" "$prec" <<'PY'
import collections, csv, glob, json, sys
out, prec = sys.argv[1], sys.argv[2]
# ---- kernel trace: per-kernel count / total / average, and the two launch sizes of the network kernel apart
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
tot = sum(sum(v) for v in per.values())
with open(out + "/kernel_stats.txt", "w") as fo:
    fo.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --precision %s --no-cpu-baseline --no-extras\n" % prec)
    fo.write("%-64s %6s %12s %10s %10s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "share"))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        fo.write("%-64s %6d %12.3f %10.4f %10.4f %10.4f %6.2f%%\n" % (k[:64], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    for k, v in per.items():
        if ("mlp_" in k or "mlp16_" in k) and "kernel" in k and "pack" not in k:
            big = [x for x in v if x > 0.5 * max(v)]
            small = [x for x in v if x <= 0.5 * max(v)]
            fo.write("\n%s\n  fine-pass launches (192 samples/ray): %d, avg %.4f ms\n  coarse-pass launches (64 samples/ray): %d, avg %.4f ms\n"
                     % (k[:80], len(big), sum(big) / max(1, len(big)), len(small), sum(small) / max(1, len(small))))
# ---- HBM-side bytes per launch.  FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B? -> the raw
# counter is in bytes/1024 on rocprofv3 for gfx9; keep the raw number and the corrected one side by side.
traffic = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            traffic[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
with open(out + "/traffic.txt", "w") as fo:
    fo.write("per-launch counter values (raw units as rocprofv3 prints them; KB = 1024 B)\n")
    for k, d in traffic.items():
        fo.write(k[:100] + "\n")
        for c, v in d.items():
            fo.write("  %-12s launches %4d  max %14.1f  min %14.1f  mean %14.1f\n" % (c, len(v), max(v), min(v), sum(v) / len(v)))
        if ("mlp_" in k or "mlp16_" in k) and "pack" not in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            # largest launches = fine pass; FETCH_SIZE x2 (gfx950 tallies 128-B read requests as 64 B), WRITE_SIZE as is
            fs = [x for x in d["FETCH_SIZE"] if x > 0.5 * max(d["FETCH_SIZE"])]
            ws = [x for x in d["WRITE_SIZE"] if x > 0.5 * max(d["WRITE_SIZE"])]
            res = {"kernel": k[:100], "precision": prec, "fetch_size_kb_raw": sum(fs) / len(fs), "write_size_kb_raw": sum(ws) / len(ws),
                   "bytes_per_fine_launch": (2.0 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024.0,
                   "correction": "FETCH_SIZE x 2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE x 1, KB = 1024 B"}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(open(out + "/kernel_stats.txt").read())
print(open(out + "/traffic.txt").read())
print(json.dumps(res))
PY
